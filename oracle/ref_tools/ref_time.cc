/* oracle/_ref tool (test infrastructure, CPU baseline leg of bench.py): times the REFERENCE
 * decoder, single thread, on an IVF file held in memory (mmap, warm).  Phases follow the
 * reference's own split (decoder.cc:101-118): parse = Decoder::parse_frame<F>
 * (DecoderState::parse_and_apply, decoder_state.hh:72-167), reconstruct+loopfilter =
 * Decoder::decode_frame<F> (Frame::decode + Frame::loopfilter, frame.cc:139-250).
 *   ref_time in.ivf [repeats]
 * stdout: JSON {"frames":..,"macroblocks":..,"seconds":..,"parse_s":..,"decode_s":..,"mb_per_s":..}
 * NOTE: built WITHOUT the libvpx x86 asm kernels (no yasm/nasm in the image). */
#include <chrono>
#include <cstdio>
#include <iostream>
#include "ivf.hh"
#include "decoder.hh"
#include "frame.hh"
#include "uncompressed_chunk.hh"

using namespace std;
using clk = chrono::steady_clock;

int main( int argc, char * argv[] )
{
  try {
    if ( argc < 2 ) { cerr << "usage: ref_time in.ivf [repeats]\n"; return 2; }
    const int repeats = argc > 2 ? atoi( argv[ 2 ] ) : 1;
    IVF ivf( argv[ 1 ] );
    const unsigned mbw = ( ivf.width() + 15 ) / 16, mbh = ( ivf.height() + 15 ) / 16;
    double parse_s = 0, decode_s = 0;
    unsigned long frames = 0;
    size_t sink = 0;
    for ( int rep = 0; rep < repeats; rep++ ) {
      Decoder decoder( ivf.width(), ivf.height() );
      for ( unsigned i = 0; i < ivf.frame_count(); i++ ) {
        const Chunk chunk = ivf.frame( i );
        const auto t0 = clk::now();
        UncompressedChunk uc = decoder.decompress_frame( chunk );
        if ( uc.key_frame() ) {
          KeyFrame f = decoder.parse_frame<KeyFrame>( uc );
          const auto t1 = clk::now();
          auto out = decoder.decode_frame( f );
          const auto t2 = clk::now();
          sink += out.second.get().Y().at( 0, 0 );
          parse_s += chrono::duration<double>( t1 - t0 ).count();
          decode_s += chrono::duration<double>( t2 - t1 ).count();
        } else {
          InterFrame f = decoder.parse_frame<InterFrame>( uc );
          const auto t1 = clk::now();
          auto out = decoder.decode_frame( f );
          const auto t2 = clk::now();
          sink += out.second.get().Y().at( 0, 0 );
          parse_s += chrono::duration<double>( t1 - t0 ).count();
          decode_s += chrono::duration<double>( t2 - t1 ).count();
        }
        frames++;
      }
    }
    const double total = parse_s + decode_s;
    const double mbs = double( frames ) * mbw * mbh;
    printf( "{\"frames\": %lu, \"macroblocks\": %.0f, \"seconds\": %.6f, \"parse_s\": %.6f, "
            "\"decode_s\": %.6f, \"mb_per_s\": %.1f, \"sink\": %zu}\n",
            frames, mbs, total, parse_s, decode_s, mbs / total, sink );
  } catch ( const exception & e ) {
    cerr << "ref_time: " << e.what() << "\n";
    return 1;
  }
  return 0;
}
