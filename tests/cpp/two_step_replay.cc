// Test program (not product code): the two-step decode of the reference's callers, written the way frontend/xc-enc.cc:286-300
// (the -r replay loop) and frontend/xc-terminate-chunk.cc use it -- UncompressedChunk, Decoder::parse_frame<KeyFrame|InterFrame>,
// Decoder::decode_frame -- against the compat headers (the reference's header names), plus References( MutableRasterHandle && )
// (decoder.cc:165-169) with Decoder( DecoderState, References ).  Prints one line per frame: shown flag and the raster's hash by
// both routes; exit status 0 iff the two-step decoder and a get_frame_output decoder agree on every frame and every hash.
//   two_step_replay file.ivf
#include <cstdio>
#include <iostream>
#include <utility>
#include <vector>

#include "decoder.hh"
#include "frame.hh"
#include "ivf.hh"
#include "uncompressed_chunk.hh"

using namespace std;

int main( int argc, char * argv[] )
{
  if ( argc != 2 ) { cerr << "usage: two_step_replay file.ivf\n"; return 2; }
  try {
    IVF ivf { argv[ 1 ] };
    Decoder pred_decoder( ivf.width(), ivf.height() ), plain( ivf.width(), ivf.height() );
    vector<pair<Optional<KeyFrame>, Optional<InterFrame> > > prediction_frames;
    bool ok = true;
    for ( unsigned int i = 0; i < ivf.frame_count(); i++ ) {
      UncompressedChunk unch { ivf.frame( i ), ivf.width(), ivf.height(), false };
      pair<bool, RasterHandle> out { false, RasterHandle() };
      if ( unch.key_frame() ) {
        KeyFrame frame = pred_decoder.parse_frame<KeyFrame>( unch );
        out = pred_decoder.decode_frame( frame );
        prediction_frames.emplace_back( move( frame ), Optional<InterFrame>() );
      } else {
        InterFrame frame = pred_decoder.parse_frame<InterFrame>( unch );
        out = pred_decoder.decode_frame( frame );
        prediction_frames.emplace_back( Optional<KeyFrame>(), move( frame ) );
      }
      const pair<bool, RasterHandle> want = plain.get_frame_output( ivf.frame( i ) );
      const bool same = out.first == want.first and out.second.hash() == want.second.hash() and out.first == unch.show_frame();
      printf( "frame %u shown=%d two_step=%016zx one_step=%016zx %s\n", i, int( out.first ), out.second.hash(), want.second.hash(), same ? "ok" : "MISMATCH" );
      ok = ok and same;
    }
    ok = ok and pred_decoder.get_hash().hash() == plain.get_hash().hash() and pred_decoder == plain;

    // a decoder continued from a state and a caller-made raster: References( MutableRasterHandle && ) aliases all three
    MutableRasterHandle blank { ivf.width(), ivf.height() };
    for ( unsigned int r = 0; r < blank.get().Y().height(); r++ ) for ( unsigned int c = 0; c < blank.get().Y().width(); c++ ) blank.get().Y().at( c, r ) = 0;
    for ( unsigned int r = 0; r < blank.get().U().height(); r++ ) for ( unsigned int c = 0; c < blank.get().U().width(); c++ ) { blank.get().U().at( c, r ) = 0; blank.get().V().at( c, r ) = 0; }
    References refs( move( blank ) );
    ok = ok and refs.last.hash() == refs.golden.hash() and refs.golden.hash() == refs.alternative.hash();
    Decoder fresh( ivf.width(), ivf.height() );
    Decoder continued( fresh.get_state(), refs );
    ok = ok and continued == fresh;                                    // (a new decoder's references are one all-zero raster)
    const pair<bool, RasterHandle> a = continued.get_frame_output( ivf.frame( 0 ) ), b = fresh.get_frame_output( ivf.frame( 0 ) );
    ok = ok and a.second.hash() == b.second.hash();
    printf( "%s\n", ok ? "ALL OK" : "FAILED" );
    return ok ? 0 : 1;
  } catch ( const exception & e ) {
    cerr << "two_step_replay: " << e.what() << "\n";
    return 1;
  }
}
