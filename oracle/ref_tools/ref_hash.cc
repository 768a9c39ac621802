/* oracle/_ref tool (test infrastructure): the REFERENCE decoder's hashes, computed by the reference itself (compiled in
 * place from /root/reference/src; boost::hash_combine comes from oracle/ref_shims, i.e. the pre-1.81 boost formula over
 * std::hash, which is the identity for the integer types hashed here).
 *
 *   ref_hash in.ivf        one JSON line per frame: DecoderState::hash, the three reference rasters' hashes,
 *                          DecoderHash::hash, Decoder::minihash (decoder.cc:143-153,266-281,482-529, raster.cc:52-61)
 * Used by tests/golden/make_hash_golden.py to pin aa_stream_decoder_hash / aa_stream_minihash. */
#include <cstdio>
#include <iostream>
#include "ivf.hh"
#include "decoder.hh"

using namespace std;

int main( int argc, char * argv[] )
{
  try {
    if ( argc != 2 ) { cerr << "usage: ref_hash in.ivf\n"; return 2; }
    IVF ivf( argv[ 1 ] );
    Decoder decoder( ivf.width(), ivf.height() );
    for ( unsigned i = 0; i < ivf.frame_count(); i++ ) {
      decoder.get_frame_output( ivf.frame( i ) );
      const References refs = decoder.get_references();
      printf( "{\"frame\": %u, \"state\": %zu, \"last\": %zu, \"golden\": %zu, \"alternative\": %zu, \"hash\": %zu, \"minihash\": %u}\n",
              i, decoder.get_state().hash(), refs.last.hash(), refs.golden.hash(), refs.alternative.hash(),
              decoder.get_hash().hash(), decoder.minihash() );
    }
  } catch ( const exception & e ) {
    cerr << "ref_hash: " << e.what() << "\n";
    return 1;
  }
  return 0;
}
