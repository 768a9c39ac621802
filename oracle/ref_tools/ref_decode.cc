/* oracle/_ref tool (test infrastructure): drives the REFERENCE decoder (compiled in place
 * from /root/reference/src) over an IVF file and dumps what it produced, so that the C
 * restatement (oracle/vp8_oracle.c) and the HIP path can be compared byte for byte.
 *
 *   ref_decode in.ivf out.raw            every decoded frame (hidden ones included), the
 *                                        three PADDED planes Y,U,V (16-aligned dims)
 *   ref_decode --display in.ivf out.raw  only shown frames, display rectangle as planar
 *                                        I420 == tests/decode-to-stdout.cc:43-49
 *   ref_decode --conceal in.ivf out.raw  Decoder::set_error_concealment( true ) (decoder.hh:298) first
 * stdout: one line per frame "frame <n> key=<0|1> shown=<0|1> bytes=<n>".
 * Entry points used: Decoder::get_frame_output (decoder.cc:125-135), IVF (util/ivf.cc). */
#include <cstdio>
#include <cstring>
#include <iostream>
#include <string>
#include "ivf.hh"
#include "decoder.hh"
#include "frame.hh"
#include "uncompressed_chunk.hh"

using namespace std;

static void write_plane( FILE * f, const TwoD<uint8_t> & p )
{
  for ( unsigned r = 0; r < p.height(); r++ ) {
    if ( fwrite( &p.at( 0, r ), p.width(), 1, f ) != 1 ) throw runtime_error( "short write" );
  }
}

int main( int argc, char * argv[] )
{
  try {
    bool display = false, conceal = false;
    int a = 1;
    while ( a < argc and argv[ a ][ 0 ] == '-' ) {
      if ( string( argv[ a ] ) == "--display" ) display = true;
      else if ( string( argv[ a ] ) == "--conceal" ) conceal = true;
      else break;
      a++;
    }
    if ( argc - a != 2 ) { cerr << "usage: ref_decode [--display] [--conceal] in.ivf out.raw\n"; return 2; }
    IVF ivf( argv[ a ] );
    FILE * out = fopen( argv[ a + 1 ], "wb" );
    if ( not out ) { perror( "fopen" ); return 2; }
    Decoder decoder( ivf.width(), ivf.height() );
    decoder.set_error_concealment( conceal );
    for ( unsigned i = 0; i < ivf.frame_count(); i++ ) {
      const Chunk chunk = ivf.frame( i );
      const bool key = chunk.size() and not ( chunk.octet() & 1 );
      pair<bool, RasterHandle> res = decoder.get_frame_output( chunk );
      const VP8Raster & r = res.second.get();
      if ( display ) {
        if ( res.first ) r.dump( out );
      } else {
        write_plane( out, r.Y() ); write_plane( out, r.U() ); write_plane( out, r.V() );
      }
      printf( "frame %u key=%d shown=%d bytes=%lu\n", i, key, res.first, (unsigned long) chunk.size() );
    }
    fclose( out );
  } catch ( const exception & e ) {
    cerr << "ref_decode: " << e.what() << "\n";
    return 1;
  }
  return 0;
}
