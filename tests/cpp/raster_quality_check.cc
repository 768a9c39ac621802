/* Host-only members of the shim's VP8Raster that the encoder feedback path uses: quality() = SSIM of the luma
 * planes (raster.cc:63-66), copy_from (raster.cc:78-83).  Prints the quality of a raster against three distortions of it;
 * tests/test_cpp_mirror.py compares the numbers with the oracle's restatement.  No GPU involved. */
#define ALFALFA_AMD_GLOBAL_NAMES
#include "alfalfa_amd/alfalfa.hh"

#include <cstdint>
#include <cstdio>
#include <cstdlib>

static uint32_t lcg( uint32_t & s ) { s = s * 1664525u + 1013904223u; return s >> 8; }

int main( int argc, char * argv[] )
{
  if ( argc != 4 ) { fprintf( stderr, "usage: %s WIDTH HEIGHT SEED\n", argv[0] ); return 2; }
  const unsigned width = atoi( argv[1] ), height = atoi( argv[2] );
  uint32_t seed = atoi( argv[3] );
  VP8Raster a( width, height ), b( width, height );
  for ( unsigned row = 0; row < a.height(); row++ )
    for ( unsigned col = 0; col < a.width(); col++ ) a.Y().at( col, row ) = static_cast<uint8_t>( lcg( seed ) );
  b.copy_from( a );
  if ( !( a == b ) ) { fprintf( stderr, "copy_from: rasters differ\n" ); return 1; }
  printf( "%.17g\n", a.quality( b ) );
  for ( const int amplitude : { 2, 9, 60 } ) {
    b.copy_from( a );
    for ( unsigned row = 0; row < a.height(); row++ )
      for ( unsigned col = 0; col < a.width(); col++ ) {
        const int v = a.Y().at( col, row ) + static_cast<int>( lcg( seed ) % ( 2 * amplitude + 1 ) ) - amplitude;
        b.Y().at( col, row ) = static_cast<uint8_t>( v < 0 ? 0 : ( v > 255 ? 255 : v ) );
      }
    printf( "%.17g\n", a.quality( b ) );
    printf( "%.17g\n", ssim( b.Y(), a.Y() ) );
  }
  return 0;
}
