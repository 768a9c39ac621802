/* Build glue for oracle/_ref/xc-enc-ssim only: the reference's ssim() (util/ssim.hh:31; util/ssim.cc:57-71 calls into libx264,
 * absent here) answered by OUR restatement of x264's algorithm (oracle/ssim_x264.c).  With it the reference ENCODER's loop-filter
 * level search (encoder.cc:459-516) runs its own code -- candidate filtering, range, early stop -- on the quality measure
 * aa_stream_lf_search uses, so that the levels it writes into frame headers can be compared with ours (tests/test_lf_search_pin.py). */
#include "2d.hh"
#include "ssim.hh"
extern "C" double oracle_ssim_plane( const uint8_t * a, const uint8_t * b, int width, int height, float * windows_out );
double ssim( const TwoD<uint8_t> & a, const TwoD<uint8_t> & b )
{
  return oracle_ssim_plane( &a.at( 0, 0 ), &b.at( 0, 0 ), a.width(), a.height(), nullptr );
}
