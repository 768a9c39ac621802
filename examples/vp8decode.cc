// vp8decode on the MI355X decode path: the reference's frontend/vp8decode.cc (:43-101) written against the mirror
// headers -- same option letters, same output (YUV4MPEG2: header of the first shown raster, then one FRAME per shown
// frame).  `-s decoder_state` continues from a reference-format state file (EncoderStateDeserializer::build<Player>); the
// reference's minihash check of state against IVF header is the one thing missing.
//   g++ -std=c++14 -O2 -Iinclude examples/vp8decode.cc -Lalfalfa_amd/lib -lalfalfa_amd -Wl,-rpath,$PWD/alfalfa_amd/lib
#define ALFALFA_AMD_GLOBAL_NAMES
#include "alfalfa_amd/alfalfa.hh"

#include <getopt.h>
#include <iostream>

using namespace std;

static int usage( char * argv0 )
{
  cerr << "Usage: " << argv0 << " [-s decoder_state] [-o y4m_output] input_file" << endl;
  return EXIT_FAILURE;
}

int main( int argc, char * argv[] )
{
  try {
    if ( argc < 2 ) return usage( argv[0] );
    FileDescriptor y4m_fd;
    char * decoder_state = nullptr;
    while ( true ) {
      const int opt = getopt( argc, argv, "s:o:" );
      if ( opt == -1 ) break;
      switch ( static_cast<char>( opt ) ) {
      case 's': decoder_state = optarg; break;
      case 'o': y4m_fd = FileDescriptor( fopen( optarg, "wb" ) ); break;
      default: return usage( argv[0] );
      }
    }
    if ( optind >= argc ) return usage( argv[0] );

    Player player = decoder_state == nullptr ? Player( argv[optind] )
                                             : EncoderStateDeserializer::build<Player>( decoder_state, string( argv[optind] ) );
    while ( not player.eof() ) {
      RasterHandle raster = player.advance();
      if ( y4m_fd.valid() ) {
        if ( y4m_fd.tell() == 0 ) y4m_fd.write( YUV4MPEGHeader( raster ).to_string() );      // position 0: no header yet
        YUV4MPEGFrameWriter::write( raster, y4m_fd );
      }
    }
  } catch ( const exception & e ) {
    print_exception( argv[0], e );
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
