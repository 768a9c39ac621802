// xc-decode-bundle on the MI355X decode path: frontend/decode-bundle.cc (:46-108) against the mirror headers -- decodes a
// sequence of IVF files whose names are read from standard input, ONE decoder carried from file to file (optionally
// starting from a reference-format state file), to a YUV4MPEG2 video on standard output.  Not carried over: the minihash
// check of each IVF header against the decoder (boost::hash_combine based) and the state hashes printed on stderr.
//   g++ -std=c++14 -O2 -Iinclude examples/xc_decode_bundle.cc -Lalfalfa_amd/lib -lalfalfa_amd -Wl,-rpath,$PWD/alfalfa_amd/lib
#define ALFALFA_AMD_GLOBAL_NAMES
#include "alfalfa_amd/alfalfa.hh"

#include <iostream>

using namespace std;

int main( int argc, char * argv[] )
{
  try {
    if ( argc > 2 ) {
      cerr << "Usage: " << argv[0] << " [starting_state]" << endl;
      return EXIT_FAILURE;
    }
    FileDescriptor out( 1 );      // stdout
    unique_ptr<FramePlayer> player;
    while ( true ) {
      string filename;
      getline( cin, filename );
      if ( not cin.good() ) break;
      cerr << "Opening " << filename << "... ";
      IVF ivf { filename };
      cerr << "done (" << ivf.frame_count() << " frames).\n";
      if ( not player ) {
        cerr << "Initializing with size " << ivf.width() << "x" << ivf.height() << "\n";
        if ( argc > 1 ) {
          player.reset( new FramePlayer( EncoderStateDeserializer::build<FramePlayer>( argv[1] ) ) );
          if ( ivf.width() != player->width() or ivf.height() != player->height() ) throw Unsupported( "state vs. file dimension mismatch" );
        } else {
          player.reset( new FramePlayer( ivf.width(), ivf.height() ) );
        }
        out.write( YUV4MPEGHeader( player->example_raster() ).to_string() );
      }
      for ( unsigned int frame_no = 0; frame_no < ivf.frame_count(); frame_no++ ) {
        Optional<RasterHandle> raster = player->decode( ivf.frame( frame_no ) );
        if ( raster.initialized() ) YUV4MPEGFrameWriter::write( raster.get(), out );
      }
    }
  } catch ( const exception & e ) {
    print_exception( argv[0], e );
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
