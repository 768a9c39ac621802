// Golden-test driver on the C++ shim (include/alfalfa_amd/alfalfa.hh with the reference's class names exported): for each
// IVF named on the command line, every SHOWN frame's display rectangle is appended to stdout as planar I420.  For one
// file that byte stream is what the reference's own golden test hashes (src/tests/decoding.test: sha1sum of
// `decode-to-stdout FILE`), which is how tests/test_cpp_mirror.py uses it.
#define ALFALFA_AMD_GLOBAL_NAMES
#include "alfalfa_amd/alfalfa.hh"

#include <cstdio>
#include <string>
#include <vector>

namespace {

// Player::advance skips hidden frames and returns the next shown raster (player.cc:134-144)
size_t dump_file( const std::string & path, FILE * out )
{
  Player player( path );
  size_t shown = 0;
  for ( ; !player.eof(); ++shown ) {
    const RasterHandle frame = player.advance();
    frame.get().dump( out );
  }
  return shown;
}

} // namespace

int main( int argc, char * argv[] )
{
  const std::vector<std::string> inputs( argv + ( argc > 0 ? 1 : 0 ), argv + argc );
  if ( inputs.empty() ) {
    std::fprintf( stderr, "usage: %s FILE.ivf [FILE.ivf ...]\n", argc > 0 ? argv[0] : "dump_shown_frames" );
    return 2;
  }
  int status = 0;
  for ( const std::string & path : inputs ) {
    try {
      dump_file( path, stdout );
    } catch ( const std::exception & e ) {
      print_exception( path.c_str(), e );
      status = 1;
      break;
    }
  }
  std::fflush( stdout );
  return status;
}
