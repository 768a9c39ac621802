// Boundary test (tests/test_reference_callers.py): the Decoder / DecoderState members the reference's callers use beyond
// decode -- get_state(), Decoder( DecoderState, References ), operator==, get_hash(), minihash(), serialize round trip --
// and that RasterHandles give their rasters back: a long FilePlayer loop must hold HBM flat.
//   decoder_state_check FILE.ivf [loops]
#include <cstdio>
#include <iostream>
#define ALFALFA_AMD_GLOBAL_NAMES
#include "alfalfa_amd/alfalfa.hh"

using namespace std;

int main( int argc, char * argv[] )
{
  try {
    if ( argc < 2 ) { cerr << "Usage: " << argv[0] << " FILE.ivf [loops]" << endl; return EXIT_FAILURE; }
    const IVF ivf( argv[1] );
    const unsigned half = ivf.frame_count() / 2;

    // a: straight through.  b: stops half way; c continues from b's ( DecoderState, References ) -- decoder.cc:43-46
    Decoder a( ivf.width(), ivf.height() ), b( ivf.width(), ivf.height() );
    for ( unsigned i = 0; i < half; i++ ) { a.get_frame_output( ivf.frame( i ) ); b.get_frame_output( ivf.frame( i ) ); }
    if ( !( a == b ) || a.minihash() != b.minihash() || a.get_hash().str() != b.get_hash().str() ) { cerr << "equal decoders differ" << endl; return 1; }
    const DecoderState st = b.get_state();
    if ( st.width != ivf.width() || st.height != ivf.height() || st.hash() != b.get_state().hash() ) { cerr << "state" << endl; return 1; }
    Decoder c( st, b.get_references() );
    if ( !( c == a ) ) { cerr << "Decoder( state, references ) != source: " << c.get_hash().str() << " vs " << a.get_hash().str() << endl; return 1; }
    for ( unsigned i = half; i < ivf.frame_count(); i++ ) {
      const pair<bool, RasterHandle> x = a.get_frame_output( ivf.frame( i ) ), y = c.get_frame_output( ivf.frame( i ) );
      if ( x.first != y.first || x.second != y.second || !( x.second.get() == y.second.get() ) ) { cerr << "continuation differs at frame " << i << endl; return 1; }
    }
    if ( !( a == c ) || a == b ) { cerr << "operator== after continuation" << endl; return 1; }
    if ( !a.minihash_match( 0 ) || !a.minihash_match( c.minihash() ) || a.minihash_match( c.minihash() ^ 1 ) ) { cerr << "minihash_match" << endl; return 1; }

    // DecoderState / Decoder wire format round trips (decoder.cc:54-81,283-330)
    EncoderStateSerializer s1; a.get_state().serialize( s1 );
    EncoderStateDeserializer d1( s1.data() );
    if ( !( DecoderState::deserialize( d1 ) == a.get_state() ) ) { cerr << "DecoderState round trip" << endl; return 1; }
    EncoderStateSerializer s2; a.serialize( s2 );
    EncoderStateDeserializer d2( s2.data() );
    Decoder e = Decoder::deserialize( d2 );
    if ( e.get_state() != a.get_state() || e.get_references().last != a.get_references().last ) { cerr << "Decoder round trip" << endl; return 1; }
    printf( "minihash %08x state %zx\n", a.minihash(), a.get_state().hash() );

    // RasterHandle lifetime: every handle dies at the end of an iteration; HBM must stay flat
    const int loops = argc > 2 ? atoi( argv[2] ) : 0;
    size_t free_first = 0, free_min = ~size_t( 0 ), total = 0;
    unsigned long frames = 0;
    for ( int l = 0; l < loops; l++ ) {
      FilePlayer p( argv[1] );
      unsigned sink = 0;
      while ( !p.eof() ) { RasterHandle r = p.advance(); frames++; if ( frames % 97 == 0 ) sink += r.get().Y().at( 0, 0 ); }
      if ( sink == 0xFFFFFFFFu ) return 9;
      size_t f = 0;
      alfalfa_amd::check( aa_ctx_memory( alfalfa_amd::GpuContext::process_default()->get(), &f, &total ) );
      if ( l == 2 ) free_first = f;
      if ( l >= 2 && f < free_min ) free_min = f;
    }
    if ( loops > 3 ) {
      printf( "frames %lu free after warm-up %zu min %zu\n", frames, free_first, free_min );
      if ( free_first - free_min > ( size_t( 96 ) << 20 ) ) { cerr << "HBM use grows" << endl; return 1; }
    }
  } catch ( const exception & e ) {
    print_exception( argv[0], e );
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
