/* TEST INFRASTRUCTURE -- oracle for the quality measure of the encoder's loop-filter search.
 *
 * The reference computes BaseRaster::quality() = ssim( Y(), other.Y() ) (util/raster.cc:63-66) by calling INTO libx264:
 * x264_8_pixel_ssim_wxh over the whole padded luma plane, stride = width, result / count (util/ssim.cc:57-71).  libx264 is
 * a system package of the reference's build (no version pinned by the reference; the bit-depth-prefixed symbol name means
 * x264 >= core 153) and is absent from /root/reference and from this image.  PARITY UNPINNED: what follows restates the
 * published algorithm of x264's common/pixel.c (ssim_4x4x2_core, ssim_end1, ssim_end4, pixel_ssim_wxh -- the plain-C
 * versions; x264's SIMD versions add the four values of ssim_end4 in a different order, so the real library may differ
 * from this in the last bits of a float).  Anchors: the reference's call site (plane, stride, division by the count).
 *
 * Structure: a window is 8x8 pixels, windows step by 4 in both directions; per window the sums s1, s2 (pixels), ss (squares
 * of both), s12 (products) come from four 4x4 blocks; ssim_end1 turns them into one float; floats are accumulated IN FLOAT,
 * four windows at a time from 0.0f, row by row. */
#include <stdint.h>
#include <stdlib.h>

static void sums_4x4( const uint8_t * a, int sa, const uint8_t * b, int sb, int s[4] )
{
  uint32_t s1 = 0, s2 = 0, ss = 0, s12 = 0;
  for ( int y = 0; y < 4; y++ )
    for ( int x = 0; x < 4; x++ ) {
      const int p = a[x + y * sa], q = b[x + y * sb];
      s1 += p; s2 += q; ss += p * p; ss += q * q; s12 += p * q;
    }
  s[0] = (int) s1; s[1] = (int) s2; s[2] = (int) ss; s[3] = (int) s12;
}

float oracle_ssim_window( int s1, int s2, int ss, int s12 )
{
  const int c1 = (int) ( .01 * .01 * 255 * 255 * 64 + .5 );
  const int c2 = (int) ( .03 * .03 * 255 * 255 * 64 * 63 + .5 );
  const int vars = ss * 64 - s1 * s1 - s2 * s2;
  const int covar = s12 * 64 - s1 * s2;
  return (float) ( 2 * s1 * s2 + c1 ) * (float) ( 2 * covar + c2 ) / ( (float) ( s1 * s1 + s2 * s2 + c1 ) * (float) ( vars + c2 ) );
}

/* windows_out (optional): (height/4 - 1) x (width/4 - 1) floats, row-major -- what the device kernel must reproduce */
double oracle_ssim_plane( const uint8_t * a, const uint8_t * b, int width, int height, float * windows_out )
{
  const int w4 = width >> 2, h4 = height >> 2;
  if ( w4 < 2 || h4 < 2 ) return 0.0;
  int ( *rows )[4] = malloc( sizeof( int[4] ) * (size_t) w4 * 2 );
  int ( *cur )[4] = rows, ( *prev )[4] = rows + w4;
  float total = 0.0f;
  for ( int x = 0; x < w4; x++ ) sums_4x4( a + 4 * x, width, b + 4 * x, width, prev[x] );
  for ( int y = 1; y < h4; y++ ) {
    for ( int x = 0; x < w4; x++ ) sums_4x4( a + 4 * x + 4 * y * width, width, b + 4 * x + 4 * y * width, width, cur[x] );
    for ( int x = 0; x < w4 - 1; x += 4 ) {
      const int n = w4 - x - 1 < 4 ? w4 - x - 1 : 4;
      float part = 0.0f;
      for ( int i = 0; i < n; i++ ) {
        const float v = oracle_ssim_window( cur[x + i][0] + cur[x + i + 1][0] + prev[x + i][0] + prev[x + i + 1][0],
                                            cur[x + i][1] + cur[x + i + 1][1] + prev[x + i][1] + prev[x + i + 1][1],
                                            cur[x + i][2] + cur[x + i + 1][2] + prev[x + i][2] + prev[x + i + 1][2],
                                            cur[x + i][3] + cur[x + i + 1][3] + prev[x + i][3] + prev[x + i + 1][3] );
        if ( windows_out ) windows_out[(size_t) ( y - 1 ) * ( w4 - 1 ) + x + i] = v;
        part += v;
      }
      total += part;
    }
    int ( *t )[4] = cur; cur = prev; prev = t;
  }
  free( rows );
  return (double) total / ( (double) ( h4 - 1 ) * ( w4 - 1 ) );
}
