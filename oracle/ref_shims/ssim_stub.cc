/* Build glue for oracle/_ref only.  util/ssim.cc needs x264 (absent).  The only caller on
 * our path is the reference ENCODER's loop-filter level search (encoder.cc:489-508), which
 * ranks candidate levels by this value, so any monotone quality measure will do: we return
 * a PSNR-like score so that generated test streams carry non-zero loop_filter_level. */
#include <cmath>
#include "2d.hh"
#include "ssim.hh"
double ssim( const TwoD<uint8_t> & a, const TwoD<uint8_t> & b )
{
  double sse = 0;
  for ( unsigned r = 0; r < a.height(); r++ )
    for ( unsigned c = 0; c < a.width(); c++ ) {
      const double d = double( a.at( c, r ) ) - double( b.at( c, r ) );
      sse += d * d;
    }
  const double mse = sse / ( double( a.width() ) * a.height() );
  return 1.0 - mse / ( 255.0 * 255.0 );
}
