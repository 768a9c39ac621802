// Per-element integer arithmetic of the VP8 reconstruction path, shared by every HIP kernel.
//
// Everything here is a pure function of its arguments (no memory, no lane ids) and is marked
// AA_HD so that tests/csrc/math_check.cpp can compile the SAME source for the host and pin it
// against the oracle on CPU; the product only ever calls these from device code.
// Reference: transform.cc:47-137 (iWHT, IDCT), quantization.cc:118-121 (dequant), prediction.cc
// (intra predictors :197-618, six-tap :645-653,861-915), loopfilter_filters.hh:50-183.
#pragma once
#include <stdint.h>

#if defined( __HIPCC__ )
#define AA_HD __host__ __device__ __forceinline__
#else
#define AA_HD inline
#endif

namespace aa {

AA_HD int clamp255( int v ) { return v < 0 ? 0 : ( v > 255 ? 255 : v ); }
AA_HD int iabs( int v ) { return v < 0 ? -v : v; }

// ---- dequantisation: int16 wrap-around product (Q4) ----
AA_HD int dequant( int coeff, int factor ) { return static_cast<int16_t>( coeff * factor ); }

// ---- 4x4 inverse DCT, one 1-D pass over (a0,a1,a2,a3) = inputs at stride positions 0,4,8,12 / 0,1,2,3 ----
AA_HD int mul20091( int a ) { return ( ( a * 20091 ) >> 16 ) + a; }
AA_HD int mul35468( int a ) { return ( a * 35468 ) >> 16; }

struct Quad { int v0, v1, v2, v3; };

// First (vertical) pass of DCTCoefficients::idct_add for one column: results are truncated to int16 (Q5).
AA_HD Quad idct_pass1( int c0, int c4, int c8, int c12 )
{
  const int t0 = c0 + c8, t1 = c0 - c8;
  const int t2 = mul35468( c4 ) - mul20091( c12 );
  const int t3 = mul20091( c4 ) + mul35468( c12 );
  Quad q;
  q.v0 = static_cast<int16_t>( t0 + t3 ); q.v1 = static_cast<int16_t>( t1 + t2 );
  q.v2 = static_cast<int16_t>( t1 - t2 ); q.v3 = static_cast<int16_t>( t0 - t3 );
  return q;
}
// Second (horizontal) pass for one row: the four residual values ((x+4)>>3) to add to the prediction.
AA_HD Quad idct_pass2( int i0, int i4, int i8, int i12 )
{
  const int t0 = i0 + i8, t1 = i0 - i8;
  const int t2 = mul35468( i4 ) - mul20091( i12 );
  const int t3 = mul20091( i4 ) + mul35468( i12 );
  Quad q;
  q.v0 = ( t0 + t3 + 4 ) >> 3; q.v1 = ( t1 + t2 + 4 ) >> 3;
  q.v2 = ( t1 - t2 + 4 ) >> 3; q.v3 = ( t0 - t3 + 4 ) >> 3;
  return q;
}

// ---- inverse Walsh-Hadamard (DCTCoefficients::iwht) ----
AA_HD Quad iwht_pass1( int c0, int c4, int c8, int c12 )   // column i: inputs i, i+4, i+8, i+12 -> rows 0..3 of column i
{
  const int a1 = c0 + c12, b1 = c4 + c8, c1 = c4 - c8, d1 = c0 - c12;
  Quad q;
  q.v0 = static_cast<int16_t>( a1 + b1 ); q.v1 = static_cast<int16_t>( c1 + d1 );
  q.v2 = static_cast<int16_t>( a1 - b1 ); q.v3 = static_cast<int16_t>( d1 - c1 );
  return q;
}
AA_HD Quad iwht_pass2( int i0, int i1, int i2, int i3 )     // row i -> DC of Y blocks (row i, col 0..3)
{
  const int a1 = i0 + i3, b1 = i1 + i2, c1 = i1 - i2, d1 = i0 - i3;
  Quad q;
  q.v0 = ( a1 + b1 + 3 ) >> 3; q.v1 = ( c1 + d1 + 3 ) >> 3;
  q.v2 = ( a1 - b1 + 3 ) >> 3; q.v3 = ( d1 - c1 + 3 ) >> 3;
  return q;
}

// ---- six-tap sub-pixel filter: one output of one pass, clamped to u8 (Q6) ----
AA_HD int sixtap_coeff( int frac, int tap )
{
  // sixtap_filters, prediction.cc:645-653
  const int16_t t[8][6] = { { 0, 0, 128, 0, 0, 0 },   { 0, -6, 123, 12, -1, 0 }, { 2, -11, 108, 36, -8, 1 }, { 0, -9, 93, 50, -6, 0 },
                            { 3, -16, 77, 77, -16, 3 }, { 0, -6, 50, 93, -9, 0 },  { 1, -8, 36, 108, -11, 2 }, { 0, -1, 12, 123, -6, 0 } };
  return t[frac][tap];
}
AA_HD int sixtap( int p0, int p1, int p2, int p3, int p4, int p5, int f0, int f1, int f2, int f3, int f4, int f5 )
{
  return clamp255( ( p0 * f0 + p1 * f1 + p2 * f2 + p3 * f3 + p4 * f4 + p5 * f5 + 64 ) >> 7 );
}

// ---- chroma MV from the sum of four luma MVs (MotionVector::luma_to_chroma, macroblock.cc:289-299) ----
AA_HD int chroma_mv( int sum_of_four )
{
  const int s = static_cast<int16_t>( sum_of_four );
  return s >= 0 ? ( s + 4 ) >> 3 : -( ( -s + 4 ) >> 3 );
}

// ---- 4x4 intra predictors ("B_PRED" sub-block modes), one output pixel ----
// E[0..12]: E[0..3] = left[3..0], E[4] = above-left, E[5..12] = above[0..7]  (Predictors::east, vp8_raster.hh:79)
AA_HD int avg3( int x, int y, int z ) { return ( x + 2 * y + z + 2 ) >> 2; }
AA_HD int avg2( int x, int y ) { return ( x + y + 1 ) >> 1; }

AA_HD int bpred_pixel( int mode, const uint8_t * E, int c, int r )
{
  const uint8_t * A = E + 5;          // above[0..7], A[-1] = above-left
  switch ( mode ) {
  case 0: {                           // B_DC_PRED: dc_predict_simple, both edges always
    int s = 4;
    for ( int i = 0; i < 4; i++ ) s += A[i] + E[i];
    return s >> 3; }
  case 1: return clamp255( E[3 - r] + A[c] - E[4] );                       // B_TM_PRED
  case 2: return avg3( A[c - 1], A[c], A[c + 1] );                         // B_VE_PRED
  case 3: return r < 3 ? avg3( E[4 - r], E[3 - r], E[2 - r] ) : avg3( E[1], E[0], E[0] );   // B_HE_PRED
  case 4: { const int i = c + r; return i < 6 ? avg3( A[i], A[i + 1], A[i + 2] ) : avg3( A[6], A[7], A[7] ); }   // B_LD_PRED
  case 5: { const int d = c - r + 3; return avg3( E[d], E[d + 1], E[d + 2] ); }                                    // B_RD_PRED
  case 6: {                           // B_VR_PRED
    const int k = 2 * c - r;          // -3..6
    if ( k == -3 ) return avg3( E[1], E[2], E[3] );
    if ( k == -2 ) return avg3( E[2], E[3], E[4] );
    if ( k == -1 ) return avg3( E[3], E[4], E[5] );
    if ( k & 1 ) return avg3( E[4 + ( k >> 1 ) ], E[5 + ( k >> 1 )], E[6 + ( k >> 1 )] );   // k=1,3,5 -> E4..6, E5..7, E6..8
    return avg2( E[4 + ( k >> 1 )], E[5 + ( k >> 1 )] );                                    // k=0,2,4,6 -> (E4,E5),(E5,E6),(E6,E7),(E7,E8)
  }
  case 7: {                           // B_VL_PRED
    if ( c == 3 && r == 2 ) return avg3( A[4], A[5], A[6] );
    if ( c == 3 && r == 3 ) return avg3( A[5], A[6], A[7] );
    const int i = c + ( r >> 1 );
    return ( r & 1 ) ? avg3( A[i], A[i + 1], A[i + 2] ) : avg2( A[i], A[i + 1] );
  }
  case 8: {                           // B_HD_PRED
    const int k = 2 * ( 3 - r ) + c;  // 0..9
    if ( k >= 8 ) return avg3( E[k - 4], E[k - 3], E[k - 2] );       // (2,0): E4,E5,E6 ; (3,0): E5,E6,E7
    return ( k & 1 ) ? avg3( E[k >> 1], E[( k >> 1 ) + 1], E[( k >> 1 ) + 2] ) : avg2( E[k >> 1], E[( k >> 1 ) + 1] );
  }
  default: {                          // 9: B_HU_PRED, L[i] = E[3-i]
    const int k = 2 * r + c;          // 0..9
    if ( k >= 6 ) return E[0];
    if ( k == 5 ) return avg3( E[1], E[0], E[0] );
    const int i = k >> 1;
    return ( k & 1 ) ? avg3( E[3 - i], E[2 - i], E[1 - i] ) : avg2( E[3 - i], E[2 - i] );
  }
  }
}

// ---- 16x16 / 8x8 intra predictors, one output pixel.  A[-1..n-1], L[0..n-1]; dc = precomputed DC value ----
AA_HD int bigpred_pixel( int mode, int above, int left, int corner, int dc )
{
  switch ( mode ) {
  case 0: return dc;
  case 1: return above;
  case 2: return left;
  default: return clamp255( left + above - corner );
  }
}
// DC value with the edge variants of VP8Raster::Block<size>::dc_predict (prediction.cc:397-431)
AA_HD int bigpred_dc( int sum_above, int sum_left, bool have_above, bool have_left, int log2n )
{
  if ( have_above && have_left ) return ( sum_above + sum_left + ( 1 << log2n ) ) >> ( log2n + 1 );
  if ( have_above ) return ( sum_above + ( 1 << ( log2n - 1 ) ) ) >> log2n;
  if ( have_left ) return ( sum_left + ( 1 << ( log2n - 1 ) ) ) >> log2n;
  return 128;
}

// ---- loop filter (normal): loopfilter_filters.hh:50-183 ----
AA_HD int sclamp( int t ) { return t < -128 ? -128 : ( t > 127 ? 127 : t ); }
AA_HD int s8( int v ) { return static_cast<int8_t>( v ); }

struct LfParams { int interior_limit, mb_limit, sb_limit, hev_threshold; };

// SimpleLoopFilter / NormalLoopFilter ctors, loopfilter.cc:81-125 (level already clamped to 1..63)
AA_HD LfParams lf_params( int level, int sharpness, bool key_frame )
{
  LfParams p;
  int il = level;
  if ( sharpness ) {
    il >>= sharpness > 4 ? 2 : 1;
    if ( il > 9 - sharpness ) il = 9 - sharpness;
  }
  if ( il < 1 ) il = 1;
  p.interior_limit = il;
  p.mb_limit = ( ( level + 2 ) * 2 ) + il;
  p.sb_limit = ( level * 2 ) + il;
  int h = level >= 15 ? 1 : 0;
  if ( level >= 40 ) h++;
  if ( level >= 20 && !key_frame ) h++;
  p.hev_threshold = h;
  return p;
}

// |a - b| of two pixel values (0..255): one v_sad_u8 on the device
AA_HD int absdiff_u8( int a, int b )
{
#if defined( __HIP_DEVICE_COMPILE__ )
  return static_cast<int>( __builtin_amdgcn_sad_u8( static_cast<unsigned>( a ), static_cast<unsigned>( b ), 0u ) );
#else
  return iabs( a - b );
#endif
}
AA_HD int imax( int a, int b ) { return a > b ? a : b; }

// vp8_filter_mask / vp8_hevmask.  Written without short-circuit operators on purpose: `||` chains compile to a cascade
// of exec-mask branches per edge on gfx950; max-of-differences is straight-line VALU (v_sad_u8 + v_max3).
AA_HD bool lf_mask( int limit, int blimit, int p3, int p2, int p1, int p0, int q0, int q1, int q2, int q3 )
{
  const int m = imax( imax( imax( absdiff_u8( p3, p2 ), absdiff_u8( p2, p1 ) ), imax( absdiff_u8( p1, p0 ), absdiff_u8( q1, q0 ) ) ),
                      imax( absdiff_u8( q2, q1 ), absdiff_u8( q3, q2 ) ) );
  const int e = absdiff_u8( p0, q0 ) * 2 + ( absdiff_u8( p1, q1 ) >> 1 );
  return ( static_cast<int>( m <= limit ) & static_cast<int>( e <= blimit ) ) != 0;
}
AA_HD bool lf_hev( int thresh, int p1, int p0, int q0, int q1 ) { return imax( absdiff_u8( p1, p0 ), absdiff_u8( q1, q0 ) ) > thresh; }

// vp8_filter (sub-block edges): p[0..3] = p1,p0,q0,q1 in place
AA_HD void lf_subblock( bool mask, bool hev, int & p1, int & p0, int & q0, int & q1 )
{
  const int ps1 = s8( p1 ^ 0x80 ), ps0 = s8( p0 ^ 0x80 ), qs0 = s8( q0 ^ 0x80 ), qs1 = s8( q1 ^ 0x80 );
  int f = hev ? sclamp( ps1 - qs1 ) : 0;
  f = sclamp( f + 3 * ( qs0 - ps0 ) );
  if ( !mask ) f = 0;
  const int f1 = sclamp( f + 4 ) >> 3, f2 = sclamp( f + 3 ) >> 3;
  q0 = ( sclamp( qs0 - f1 ) ^ 0x80 ) & 0xFF;
  p0 = ( sclamp( ps0 + f2 ) ^ 0x80 ) & 0xFF;
  int g = ( f1 + 1 ) >> 1;
  if ( hev ) g = 0;
  q1 = ( sclamp( qs1 - g ) ^ 0x80 ) & 0xFF;
  p1 = ( sclamp( ps1 + g ) ^ 0x80 ) & 0xFF;
}
// vp8_mbfilter (macroblock edges)
AA_HD void lf_macroblock( bool mask, bool hev, int & p2, int & p1, int & p0, int & q0, int & q1, int & q2 )
{
  const int ps2 = s8( p2 ^ 0x80 ), ps1 = s8( p1 ^ 0x80 ); int ps0 = s8( p0 ^ 0x80 );
  int qs0 = s8( q0 ^ 0x80 ); const int qs1 = s8( q1 ^ 0x80 ), qs2 = s8( q2 ^ 0x80 );
  int f = sclamp( sclamp( ps1 - qs1 ) + 3 * ( qs0 - ps0 ) );
  if ( !mask ) f = 0;
  const int fh = hev ? f : 0;
  const int f1 = sclamp( fh + 4 ) >> 3, f2 = sclamp( fh + 3 ) >> 3;
  qs0 = sclamp( qs0 - f1 ); ps0 = sclamp( ps0 + f2 );
  if ( hev ) f = 0;
  int u = sclamp( ( 63 + f * 27 ) >> 7 );
  q0 = ( sclamp( qs0 - u ) ^ 0x80 ) & 0xFF; p0 = ( sclamp( ps0 + u ) ^ 0x80 ) & 0xFF;
  u = sclamp( ( 63 + f * 18 ) >> 7 );
  q1 = ( sclamp( qs1 - u ) ^ 0x80 ) & 0xFF; p1 = ( sclamp( ps1 + u ) ^ 0x80 ) & 0xFF;
  u = sclamp( ( 63 + f * 9 ) >> 7 );
  q2 = ( sclamp( qs2 - u ) ^ 0x80 ) & 0xFF; p2 = ( sclamp( ps2 + u ) ^ 0x80 ) & 0xFF;
}

} // namespace aa
