/* oracle/_ref tool (test infrastructure): the REFERENCE decoder's state wire format, produced and consumed by the
 * reference itself (compiled in place from /root/reference/src).
 *
 *   ref_state save in.ivf N out.state     decode the first N frames, write Decoder::serialize (decoder.cc:54-69)
 *   ref_state resume in.ivf N in.state out.raw
 *                                         EncoderStateDeserializer::build<Decoder> (decoder.cc:48-52,71-81), decode
 *                                         frames N.. and dump their three PADDED planes (like ref_decode)
 * Used by tests/golden/make_golden.py to pin the product's aa_stream_serialize / aa_stream_deserialize. */
#include <cstdio>
#include <cstdlib>
#include <iostream>
#include <string>
#include "ivf.hh"
#include "decoder.hh"
#include "enc_state_serializer.hh"

using namespace std;

static void write_plane( FILE * f, const TwoD<uint8_t> & p )
{
  for ( unsigned r = 0; r < p.height(); r++ ) if ( fwrite( &p.at( 0, r ), p.width(), 1, f ) != 1 ) throw runtime_error( "short write" );
}

int main( int argc, char * argv[] )
{
  try {
    if ( argc < 5 ) { cerr << "usage: ref_state save in.ivf N out.state | ref_state resume in.ivf N in.state out.raw\n"; return 2; }
    const string mode = argv[ 1 ];
    IVF ivf( argv[ 2 ] );
    const unsigned n = atoi( argv[ 3 ] );
    if ( mode == "save" ) {
      Decoder decoder( ivf.width(), ivf.height() );
      for ( unsigned i = 0; i < n and i < ivf.frame_count(); i++ ) decoder.get_frame_output( ivf.frame( i ) );
      EncoderStateSerializer odata;
      decoder.serialize( odata );
      odata.write( argv[ 4 ] );
    } else if ( mode == "resume" and argc == 6 ) {
      Decoder decoder = EncoderStateDeserializer::build<Decoder>( argv[ 4 ] );
      FILE * out = fopen( argv[ 5 ], "wb" );
      if ( not out ) { perror( "fopen" ); return 2; }
      for ( unsigned i = n; i < ivf.frame_count(); i++ ) {
        pair<bool, RasterHandle> res = decoder.get_frame_output( ivf.frame( i ) );
        const VP8Raster & r = res.second.get();
        write_plane( out, r.Y() ); write_plane( out, r.U() ); write_plane( out, r.V() );
      }
      fclose( out );
    } else { cerr << "bad arguments\n"; return 2; }
  } catch ( const exception & e ) {
    cerr << "ref_state: " << e.what() << "\n";
    return 1;
  }
  return 0;
}
