// Our counterpart of the reference's golden-test driver (src/tests/decode-to-stdout.cc + decoding.test): plays an IVF
// through Player::advance() and dumps every SHOWN frame's display rectangle as planar I420, so that
// `sha1sum` of the output can be compared with the reference's.  Written against the mirror headers with the
// reference's class names exported (ALFALFA_AMD_GLOBAL_NAMES): the body is what a caller of the reference writes.
#define ALFALFA_AMD_GLOBAL_NAMES
#include "alfalfa_amd/alfalfa.hh"

#include <cstdlib>
#include <iostream>

using namespace std;

int main( int argc, char * argv[] )
{
  try {
    if ( argc != 2 ) {
      cerr << "Usage: " << argv[ 0 ] << " FILENAME" << endl;
      return EXIT_FAILURE;
    }

    Player player( argv[ 1 ] );

    while ( not player.eof() ) {
      RasterHandle raster = player.advance();
      raster.get().dump( stdout );
    }
  } catch ( const exception & e ) {
    print_exception( argv[ 0 ], e );
    return EXIT_FAILURE;
  }

  return EXIT_SUCCESS;
}
