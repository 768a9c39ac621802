/* oracle/_ref tool (test-stream tooling): re-serialises an IVF through the REFERENCE's own
 * parser and serialiser (DecoderState::parse_and_apply decoder_state.hh:72-167 ->
 * Frame::serialize encoder/serializer.cc:801-829) with header fields overridden, so that
 * encoder-generated streams exercise the loop filter (the reference encoder picks level 0
 * with the stub quality metric).  Same mechanism as tests/roundtrip.cc:90-112.
 *   ref_rewrite in.ivf out.ivf <loop_filter_level|-1> <sharpness|-1> [only_frame=-1]
 * only_frame >= 0: that frame alone gets the overrides, the others are re-serialised as they are (loop-filter candidates
 * of ONE frame for tests/test_lf_search_pin.py).
 * Pixel drift relative to the encoder's own reconstruction is irrelevant: both decoders
 * under test consume the rewritten stream. */
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <vector>
#include "ivf.hh"
#include "decoder.hh"
#include "frame.hh"
#include "decoder_state.hh"
#include "uncompressed_chunk.hh"

using namespace std;

static void put32( FILE * f, uint32_t v ) { uint8_t b[4] = { uint8_t( v ), uint8_t( v >> 8 ), uint8_t( v >> 16 ), uint8_t( v >> 24 ) }; fwrite( b, 4, 1, f ); }
static void put16( FILE * f, uint16_t v ) { uint8_t b[2] = { uint8_t( v ), uint8_t( v >> 8 ) }; fwrite( b, 2, 1, f ); }

int main( int argc, char * argv[] )
{
  try {
    if ( argc < 5 ) { cerr << "usage: ref_rewrite in.ivf out.ivf level sharpness\n"; return 2; }
    IVF ivf( argv[ 1 ] );
    const int level_arg = atoi( argv[ 3 ] ), sharp_arg = atoi( argv[ 4 ] );
    const int only_frame = argc > 5 ? atoi( argv[ 5 ] ) : -1;
    FILE * out = fopen( argv[ 2 ], "wb" );
    if ( not out ) { perror( "fopen" ); return 2; }
    fwrite( "DKIF", 4, 1, out ); put16( out, 0 ); put16( out, 32 ); fwrite( "VP80", 4, 1, out );
    put16( out, ivf.width() ); put16( out, ivf.height() ); put32( out, 30 ); put32( out, 1 );
    put32( out, ivf.frame_count() ); put32( out, 0 );
    DecoderState state( ivf.width(), ivf.height() );
    for ( unsigned i = 0; i < ivf.frame_count(); i++ ) {
      UncompressedChunk uc( ivf.frame( i ), ivf.width(), ivf.height(), false );
      vector<uint8_t> bytes;
      const bool here = only_frame < 0 or only_frame == static_cast<int>( i );
      const int level = here ? level_arg : -1, sharp = here ? sharp_arg : -1;
      if ( uc.key_frame() ) {
        KeyFrame f = state.parse_and_apply<KeyFrame>( uc );
        if ( level >= 0 ) f.mutable_header().loop_filter_level = Unsigned<6>( uint8_t( level ) );
        if ( sharp >= 0 ) f.mutable_header().sharpness_level = Unsigned<3>( uint8_t( sharp ) );
        bytes = f.serialize( state.probability_tables );
      } else {
        InterFrame f = state.parse_and_apply<InterFrame>( uc );
        if ( level >= 0 ) f.mutable_header().loop_filter_level = Unsigned<6>( uint8_t( level ) );
        if ( sharp >= 0 ) f.mutable_header().sharpness_level = Unsigned<3>( uint8_t( sharp ) );
        bytes = f.serialize( state.probability_tables );
      }
      put32( out, bytes.size() ); put32( out, i ); put32( out, 0 );
      fwrite( bytes.data(), bytes.size(), 1, out );
    }
    fclose( out );
  } catch ( const exception & e ) {
    cerr << "ref_rewrite: " << e.what() << "\n";
    return 1;
  }
  return 0;
}
