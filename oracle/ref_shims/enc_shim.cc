/* Build glue for oracle/_ref only: encoder/variance.cc defines Encoder::sad/sse/variance
 * templates but instantiates them only under HAVE_SSE2 (variance_sse2.cc). */
#include "encoder.hh"
#include "variance.cc"
template uint32_t Encoder::sad<16>( const VP8Raster::Block<16> &, const TwoDSubRange<uint8_t,16,16> & );
template uint32_t Encoder::sse<4>( const VP8Raster::Block<4> &, const TwoDSubRange<uint8_t,4,4> & );
template uint32_t Encoder::sse<8>( const VP8Raster::Block<8> &, const TwoDSubRange<uint8_t,8,8> & );
template uint32_t Encoder::sse<16>( const VP8Raster::Block<16> &, const TwoDSubRange<uint8_t,16,16> & );
template uint32_t Encoder::variance<4>( const VP8Raster::Block<4> &, const TwoDSubRange<uint8_t,4,4> & );
template uint32_t Encoder::variance<8>( const VP8Raster::Block<8> &, const TwoDSubRange<uint8_t,8,8> & );
template uint32_t Encoder::variance<16>( const VP8Raster::Block<16> &, const TwoDSubRange<uint8_t,16,16> & );
