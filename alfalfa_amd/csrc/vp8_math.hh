// Per-element integer arithmetic of the VP8 reconstruction path, shared by every HIP kernel.
//
// Everything here is a pure function of its arguments (no memory, no lane ids) and is marked
// AA_MHD so that tests/cpp/math_check.cc can compile the SAME source for the host and pin it
// against the oracle on CPU; the product only ever calls these from device code.
// Reference: transform.cc:47-137 (iWHT, IDCT), quantization.cc:118-121 (dequant), prediction.cc
// (intra predictors :197-618, six-tap :645-653,861-915), loopfilter_filters.hh:50-183.
#pragma once
#include <stdint.h>

#if defined( __HIPCC__ )
#define AA_MHD __host__ __device__ __forceinline__
#else
#define AA_MHD inline
#endif

namespace aa {

AA_MHD int clamp255( int v ) { return v < 0 ? 0 : ( v > 255 ? 255 : v ); }
AA_MHD int iabs( int v ) { return v < 0 ? -v : v; }

// ---- dequantisation: int16 wrap-around product (Q4) ----
AA_MHD int dequant( int coeff, int factor ) { return static_cast<int16_t>( coeff * factor ); }

// ---- 4x4 inverse DCT, one 1-D pass over (a0,a1,a2,a3) = inputs at stride positions 0,4,8,12 / 0,1,2,3 ----
AA_MHD int mul20091( int a ) { return ( ( a * 20091 ) >> 16 ) + a; }
AA_MHD int mul35468( int a ) { return ( a * 35468 ) >> 16; }

struct Quad { int v0, v1, v2, v3; };

// First (vertical) pass of DCTCoefficients::idct_add for one column: results are truncated to int16 (Q5).
AA_MHD Quad idct_pass1( int c0, int c4, int c8, int c12 )
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
AA_MHD Quad idct_pass2( int i0, int i4, int i8, int i12 )
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
AA_MHD Quad iwht_pass1( int c0, int c4, int c8, int c12 )   // column i: inputs i, i+4, i+8, i+12 -> rows 0..3 of column i
{
  const int a1 = c0 + c12, b1 = c4 + c8, c1 = c4 - c8, d1 = c0 - c12;
  Quad q;
  q.v0 = static_cast<int16_t>( a1 + b1 ); q.v1 = static_cast<int16_t>( c1 + d1 );
  q.v2 = static_cast<int16_t>( a1 - b1 ); q.v3 = static_cast<int16_t>( d1 - c1 );
  return q;
}
AA_MHD Quad iwht_pass2( int i0, int i1, int i2, int i3 )     // row i -> DC of Y blocks (row i, col 0..3)
{
  const int a1 = i0 + i3, b1 = i1 + i2, c1 = i1 - i2, d1 = i0 - i3;
  Quad q;
  q.v0 = ( a1 + b1 + 3 ) >> 3; q.v1 = ( c1 + d1 + 3 ) >> 3;
  q.v2 = ( a1 - b1 + 3 ) >> 3; q.v3 = ( d1 - c1 + 3 ) >> 3;
  return q;
}

// ---- six-tap sub-pixel filter: one output of one pass, clamped to u8 (Q6) ----
AA_MHD int sixtap_coeff( int frac, int tap )
{
  // sixtap_filters, prediction.cc:645-653
  const int16_t t[8][6] = { { 0, 0, 128, 0, 0, 0 },   { 0, -6, 123, 12, -1, 0 }, { 2, -11, 108, 36, -8, 1 }, { 0, -9, 93, 50, -6, 0 },
                            { 3, -16, 77, 77, -16, 3 }, { 0, -6, 50, 93, -9, 0 },  { 1, -8, 36, 108, -11, 2 }, { 0, -1, 12, 123, -6, 0 } };
  return t[frac][tap];
}
AA_MHD int sixtap( int p0, int p1, int p2, int p3, int p4, int p5, int f0, int f1, int f2, int f3, int f4, int f5 )
{
  return clamp255( ( p0 * f0 + p1 * f1 + p2 * f2 + p3 * f3 + p4 * f4 + p5 * f5 + 64 ) >> 7 );
}

// ---- chroma MV from the sum of four luma MVs (MotionVector::luma_to_chroma, macroblock.cc:289-299) ----
AA_MHD int chroma_mv( int sum_of_four )
{
  const int s = static_cast<int16_t>( sum_of_four );
  return s >= 0 ? ( s + 4 ) >> 3 : -( ( -s + 4 ) >> 3 );
}

// ---- 4x4 intra predictors ("B_PRED" sub-block modes), one output pixel ----
// E[0..12]: E[0..3] = left[3..0], E[4] = above-left, E[5..12] = above[0..7]  (Predictors::east, vp8_raster.hh:79)
AA_MHD int avg3( int x, int y, int z ) { return ( x + 2 * y + z + 2 ) >> 2; }
AA_MHD int avg2( int x, int y ) { return ( x + y + 1 ) >> 1; }

AA_MHD int bpred_pixel( int mode, const uint8_t * E, int c, int r )
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

// The same ten 4x4 predictors as data: every mode except DC and TM is "average of up to three entries of E", so a lane
// can evaluate ANY mode with one table look-up and three byte reads instead of a ten-way branch (four macroblocks with
// four different modes share one wave in k_recon_intra4).  Entry = i0 | i1 << 8 | i2 << 16 | kind << 24;
// kind 0: avg3(E[i0],E[i1],E[i2])  1: avg2(E[i0],E[i1])  2: E[i0]  3: TM clamp255(E[i0] + E[i1] - E[i2])  4: DC
enum : int { BP_AVG3 = 0, BP_AVG2 = 1, BP_COPY = 2, BP_TM = 3, BP_DC = 4 };
AA_MHD uint32_t bpred_entry( int mode, int c, int r )
{
  int i0 = 0, i1 = 0, i2 = 0, kind = BP_AVG3;
  const int A = 5;                    // index of above[0] in E
  switch ( mode ) {
  case 0: kind = BP_DC; break;
  case 1: kind = BP_TM; i0 = 3 - r; i1 = A + c; i2 = 4; break;
  case 2: i0 = A + c - 1; i1 = A + c; i2 = A + c + 1; break;
  case 3: if ( r < 3 ) { i0 = 4 - r; i1 = 3 - r; i2 = 2 - r; } else { i0 = 1; i1 = 0; i2 = 0; } break;
  case 4: { const int i = c + r; if ( i < 6 ) { i0 = A + i; i1 = A + i + 1; i2 = A + i + 2; } else { i0 = A + 6; i1 = A + 7; i2 = A + 7; } break; }
  case 5: { const int d = c - r + 3; i0 = d; i1 = d + 1; i2 = d + 2; break; }
  case 6: {
    const int k = 2 * c - r;
    if ( k == -3 ) { i0 = 1; i1 = 2; i2 = 3; }
    else if ( k == -2 ) { i0 = 2; i1 = 3; i2 = 4; }
    else if ( k == -1 ) { i0 = 3; i1 = 4; i2 = 5; }
    else if ( k & 1 ) { i0 = 4 + ( k >> 1 ); i1 = i0 + 1; i2 = i0 + 2; }
    else { kind = BP_AVG2; i0 = 4 + ( k >> 1 ); i1 = i0 + 1; }
    break; }
  case 7: {
    if ( c == 3 && r == 2 ) { i0 = A + 4; i1 = A + 5; i2 = A + 6; }
    else if ( c == 3 && r == 3 ) { i0 = A + 5; i1 = A + 6; i2 = A + 7; }
    else { const int i = c + ( r >> 1 ); i0 = A + i; i1 = A + i + 1; i2 = A + i + 2; if ( !( r & 1 ) ) kind = BP_AVG2; }
    break; }
  case 8: {
    const int k = 2 * ( 3 - r ) + c;
    if ( k >= 8 ) { i0 = k - 4; i1 = k - 3; i2 = k - 2; }
    else { i0 = k >> 1; i1 = i0 + 1; i2 = i0 + 2; if ( !( k & 1 ) ) kind = BP_AVG2; }
    break; }
  default: {
    const int k = 2 * r + c;
    if ( k >= 6 ) { kind = BP_COPY; i0 = 0; }
    else if ( k == 5 ) { i0 = 1; i1 = 0; i2 = 0; }
    else { const int i = k >> 1; i0 = 3 - i; i1 = 2 - i; i2 = 1 - i; if ( !( k & 1 ) ) kind = BP_AVG2; }
    break; }
  }
  if ( kind == BP_AVG2 || kind == BP_COPY || kind == BP_DC ) i2 = 0;        // unused taps must still be valid indices
  if ( kind == BP_COPY || kind == BP_DC ) i1 = 0;
  return ( static_cast<uint32_t>( i0 ) & 0xFFu ) | ( ( static_cast<uint32_t>( i1 ) & 0xFFu ) << 8 ) | ( ( static_cast<uint32_t>( i2 ) & 0xFFu ) << 16 )
         | ( static_cast<uint32_t>( kind ) << 24 );
}
// evaluate an entry: e0,e1,e2 = E[i0],E[i1],E[i2]; dc = (sum of above[0..3] and left[0..3] + 4) >> 3
AA_MHD int bpred_eval( int kind, int e0, int e1, int e2, int dc )
{
  const int a3 = ( e0 + 2 * e1 + e2 + 2 ) >> 2, a2 = ( e0 + e1 + 1 ) >> 1, tm = clamp255( e0 + e1 - e2 );
  int v = a3;
  v = kind == BP_AVG2 ? a2 : v;
  v = kind == BP_COPY ? e0 : v;
  v = kind == BP_TM ? tm : v;
  v = kind == BP_DC ? dc : v;
  return v;
}

// ---- 16x16 / 8x8 intra predictors, one output pixel.  A[-1..n-1], L[0..n-1]; dc = precomputed DC value ----
AA_MHD int bigpred_pixel( int mode, int above, int left, int corner, int dc )
{
  switch ( mode ) {
  case 0: return dc;
  case 1: return above;
  case 2: return left;
  default: return clamp255( left + above - corner );
  }
}
// DC value with the edge variants of VP8Raster::Block<size>::dc_predict (prediction.cc:397-431)
AA_MHD int bigpred_dc( int sum_above, int sum_left, bool have_above, bool have_left, int log2n )
{
  if ( have_above && have_left ) return ( sum_above + sum_left + ( 1 << log2n ) ) >> ( log2n + 1 );
  if ( have_above ) return ( sum_above + ( 1 << ( log2n - 1 ) ) ) >> log2n;
  if ( have_left ) return ( sum_left + ( 1 << ( log2n - 1 ) ) ) >> log2n;
  return 128;
}

// ---- loop filter (normal): loopfilter_filters.hh:50-183 ----
AA_MHD int sclamp( int t ) { return t < -128 ? -128 : ( t > 127 ? 127 : t ); }
AA_MHD int s8( int v ) { return static_cast<int8_t>( v ); }

struct LfParams { int interior_limit, mb_limit, sb_limit, hev_threshold; };

// SimpleLoopFilter / NormalLoopFilter ctors, loopfilter.cc:81-125 (level already clamped to 1..63)
AA_MHD LfParams lf_params( int level, int sharpness, bool key_frame )
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
AA_MHD int absdiff_u8( int a, int b )
{
#if defined( __HIP_DEVICE_COMPILE__ )
  return static_cast<int>( __builtin_amdgcn_sad_u8( static_cast<unsigned>( a ), static_cast<unsigned>( b ), 0u ) );
#else
  return iabs( a - b );
#endif
}
AA_MHD int imax( int a, int b ) { return a > b ? a : b; }
// sum of the four bytes of a dword
AA_MHD int absdiff_sum4( uint32_t v )
{
#if defined( __HIP_DEVICE_COMPILE__ )
  return static_cast<int>( __builtin_amdgcn_sad_u8( v, 0u, 0u ) );
#else
  return static_cast<int>( ( v & 0xFF ) + ( ( v >> 8 ) & 0xFF ) + ( ( v >> 16 ) & 0xFF ) + ( v >> 24 ) );
#endif
}

// vp8_filter_mask / vp8_hevmask.  Written without short-circuit operators on purpose: `||` chains compile to a cascade
// of exec-mask branches per edge on gfx950; max-of-differences is straight-line VALU (v_sad_u8 + v_max3).
AA_MHD bool lf_mask( int limit, int blimit, int p3, int p2, int p1, int p0, int q0, int q1, int q2, int q3 )
{
  const int m = imax( imax( imax( absdiff_u8( p3, p2 ), absdiff_u8( p2, p1 ) ), imax( absdiff_u8( p1, p0 ), absdiff_u8( q1, q0 ) ) ),
                      imax( absdiff_u8( q2, q1 ), absdiff_u8( q3, q2 ) ) );
  const int e = absdiff_u8( p0, q0 ) * 2 + ( absdiff_u8( p1, q1 ) >> 1 );
  return ( static_cast<int>( m <= limit ) & static_cast<int>( e <= blimit ) ) != 0;
}
AA_MHD bool lf_hev( int thresh, int p1, int p0, int q0, int q1 ) { return imax( absdiff_u8( p1, p0 ), absdiff_u8( q1, q0 ) ) > thresh; }

// vp8_filter (sub-block edges): p[0..3] = p1,p0,q0,q1 in place
AA_MHD void lf_subblock( bool mask, bool hev, int & p1, int & p0, int & q0, int & q1 )
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
AA_MHD void lf_macroblock( bool mask, bool hev, int & p2, int & p1, int & p0, int & q0, int & q1, int & q2 )
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

// ---- packed arithmetic: two int16 lanes per 32-bit register (v_pk_*_i16 on gfx950) ------------------------------------
// The loop filter works on pixel values 0..255 and intermediates within +-3600, so two filter positions fit one VGPR and
// every VALU instruction does the work of two.  On the host the same functions are emulated half by half (math_check).
typedef uint32_t pk2;
#if defined( __HIP_DEVICE_COMPILE__ )
typedef short pk_v2s __attribute__( ( ext_vector_type( 2 ) ) );
typedef unsigned short pk_v2u __attribute__( ( ext_vector_type( 2 ) ) );
AA_MHD pk_v2s pk_s( pk2 a ) { return __builtin_bit_cast( pk_v2s, a ); }
AA_MHD pk2 pk_r( pk_v2s a ) { return __builtin_bit_cast( pk2, a ); }
AA_MHD pk2 pk_add( pk2 a, pk2 b ) { return pk_r( pk_s( a ) + pk_s( b ) ); }
AA_MHD pk2 pk_sub( pk2 a, pk2 b ) { return pk_r( pk_s( a ) - pk_s( b ) ); }
AA_MHD pk2 pk_min( pk2 a, pk2 b ) { return pk_r( __builtin_elementwise_min( pk_s( a ), pk_s( b ) ) ); }
AA_MHD pk2 pk_max( pk2 a, pk2 b ) { return pk_r( __builtin_elementwise_max( pk_s( a ), pk_s( b ) ) ); }
AA_MHD pk2 pk_mul( pk2 a, pk2 b ) { return pk_r( pk_s( a ) * pk_s( b ) ); }
template <int N> AA_MHD pk2 pk_ashr( pk2 a ) { return pk_r( pk_s( a ) >> static_cast<short>( N ) ); }
template <int N> AA_MHD pk2 pk_shl( pk2 a ) { return pk_r( pk_s( a ) << static_cast<short>( N ) ); }
template <int N> AA_MHD pk2 pk_lshr( pk2 a ) { return __builtin_bit_cast( pk2, __builtin_bit_cast( pk_v2u, a ) >> static_cast<unsigned short>( N ) ); }
#else
AA_MHD int pk_lo( pk2 a ) { return static_cast<int16_t>( a & 0xFFFFu ); }
AA_MHD int pk_hi( pk2 a ) { return static_cast<int16_t>( a >> 16 ); }
AA_MHD pk2 pk_make( int lo, int hi ) { return ( static_cast<uint32_t>( lo ) & 0xFFFFu ) | ( static_cast<uint32_t>( hi ) << 16 ); }
AA_MHD pk2 pk_add( pk2 a, pk2 b ) { return pk_make( pk_lo( a ) + pk_lo( b ), pk_hi( a ) + pk_hi( b ) ); }
AA_MHD pk2 pk_sub( pk2 a, pk2 b ) { return pk_make( pk_lo( a ) - pk_lo( b ), pk_hi( a ) - pk_hi( b ) ); }
AA_MHD pk2 pk_min( pk2 a, pk2 b ) { return pk_make( pk_lo( a ) < pk_lo( b ) ? pk_lo( a ) : pk_lo( b ), pk_hi( a ) < pk_hi( b ) ? pk_hi( a ) : pk_hi( b ) ); }
AA_MHD pk2 pk_max( pk2 a, pk2 b ) { return pk_make( pk_lo( a ) > pk_lo( b ) ? pk_lo( a ) : pk_lo( b ), pk_hi( a ) > pk_hi( b ) ? pk_hi( a ) : pk_hi( b ) ); }
AA_MHD pk2 pk_mul( pk2 a, pk2 b ) { return pk_make( pk_lo( a ) * pk_lo( b ), pk_hi( a ) * pk_hi( b ) ); }
template <int N> AA_MHD pk2 pk_ashr( pk2 a ) { return pk_make( pk_lo( a ) >> N, pk_hi( a ) >> N ); }
template <int N> AA_MHD pk2 pk_shl( pk2 a ) { return pk_make( pk_lo( a ) << N, pk_hi( a ) << N ); }
template <int N> AA_MHD pk2 pk_lshr( pk2 a ) { return pk_make( ( a & 0xFFFFu ) >> N, ( a >> 16 ) >> N ); }
#endif
AA_MHD pk2 pk_splat( int v ) { return ( static_cast<uint32_t>( v ) & 0xFFFFu ) * 0x10001u; }
AA_MHD pk2 pk_absdiff( pk2 a, pk2 b ) { return pk_max( pk_sub( a, b ), pk_sub( b, a ) ); }
AA_MHD pk2 pk_sclamp( pk2 a ) { return pk_min( pk_max( a, pk_splat( -128 ) ), pk_splat( 127 ) ); }
AA_MHD pk2 pk_clamp255( pk2 a ) { return pk_min( pk_max( a, 0u ), pk_splat( 255 ) ); }
// per half: 0xFFFF where x <= 0 / x > 0 (x signed), else 0 -- no compare instructions (there is no packed compare)
// (|x| is far below 2^15 here, so x - 1 and -x cannot wrap: the sign bit smeared over the half is the answer)
AA_MHD pk2 pk_mask_le0( pk2 x ) { return pk_ashr<15>( pk_sub( x, pk_splat( 1 ) ) ); }
AA_MHD pk2 pk_mask_gt0( pk2 x ) { return pk_ashr<15>( pk_sub( 0u, x ) ); }

// v_perm_b32: byte k of the result is byte sel[k] of the 8-byte value {hi:lo} (0..3 = lo, 4..7 = hi), 0x0c = constant 0
AA_MHD uint32_t perm_b32( uint32_t hi, uint32_t lo, uint32_t sel )
{
#if defined( __HIP_DEVICE_COMPILE__ )
  return __builtin_amdgcn_perm( hi, lo, sel );
#else
  const uint64_t both = ( static_cast<uint64_t>( hi ) << 32 ) | lo;
  uint32_t r = 0;
  for ( int k = 0; k < 4; k++ ) {
    const uint32_t s = ( sel >> ( 8 * k ) ) & 0xFFu;
    const uint32_t byte = s <= 7 ? static_cast<uint32_t>( ( both >> ( 8 * s ) ) & 0xFFu ) : 0u;   // only 0..7 and 0x0c are used
    r |= byte << ( 8 * k );
  }
  return r;
#endif
}
// byte K of dword a (-> low half) and of dword b (-> high half), zero-extended
template <int K> AA_MHD pk2 pk_from_bytes( uint32_t a, uint32_t b ) { return perm_b32( b, a, 0x0c000c00u | ( ( 4u + K ) << 16 ) | K ); }
// the two bytes of a 16-bit LDS read -> {byte 0, byte 1}, and back
AA_MHD pk2 pk_from_u16( uint32_t w ) { return perm_b32( 0u, w, 0x0c010c00u ); }
AA_MHD uint32_t pk_to_u16( pk2 v ) { return perm_b32( 0u, v, 0x0c0c0200u ); }
// four packed pixels (x0..x3 of both positions) -> the dword of position A (low halves) and of position B (high halves)
AA_MHD void pk_to_dwords( pk2 x0, pk2 x1, pk2 x2, pk2 x3, uint32_t & a, uint32_t & b )
{
  const uint32_t t01 = perm_b32( x1, x0, 0x06020400u ), t23 = perm_b32( x3, x2, 0x06020400u );    // {A0,A1,B0,B1}, {A2,A3,B2,B3}
  a = perm_b32( t23, t01, 0x05040100u ); b = perm_b32( t23, t01, 0x07060302u );
}

struct LfParamsPk { pk2 interior_limit, mb_limit, sb_limit, hev_threshold; };
AA_MHD LfParamsPk lf_params_pk( const LfParams & p )
{
  LfParamsPk q;
  q.interior_limit = pk_splat( p.interior_limit ); q.mb_limit = pk_splat( p.mb_limit );
  q.sb_limit = pk_splat( p.sb_limit ); q.hev_threshold = pk_splat( p.hev_threshold );
  return q;
}

// lf_mask / lf_hev of two filter positions at once; `gate` (0 or ~0 per half) switches positions off.
AA_MHD void lf_masks_pk( pk2 limit, pk2 blimit, pk2 thresh, pk2 gate, pk2 p3, pk2 p2, pk2 p1, pk2 p0, pk2 q0, pk2 q1, pk2 q2, pk2 q3,
                        pk2 & mask, pk2 & hev )
{
  const pk2 dp = pk_absdiff( p1, p0 ), dq = pk_absdiff( q1, q0 );
  const pk2 inner = pk_max( dp, dq );
  const pk2 m = pk_max( pk_max( pk_max( pk_absdiff( p3, p2 ), pk_absdiff( p2, p1 ) ), inner ),
                        pk_max( pk_absdiff( q2, q1 ), pk_absdiff( q3, q2 ) ) );
  const pk2 e = pk_add( pk_shl<1>( pk_absdiff( p0, q0 ) ), pk_lshr<1>( pk_absdiff( p1, q1 ) ) );
  mask = pk_mask_le0( pk_max( pk_sub( m, limit ), pk_sub( e, blimit ) ) ) & gate;
  hev = pk_mask_gt0( pk_sub( inner, thresh ) );
}
// lf_subblock on two positions (pixel values stay in the unsigned domain: sclamp(a - 128 + d) + 128 == clamp255(a + d))
AA_MHD void lf_subblock_pk( pk2 mask, pk2 hev, pk2 & p1, pk2 & p0, pk2 & q0, pk2 & q1 )
{
  const pk2 a = pk_sclamp( pk_sub( p1, q1 ) ) & hev;
  const pk2 f = pk_sclamp( pk_add( pk_mul( pk_sub( q0, p0 ), pk_splat( 3 ) ), a ) ) & mask;
  const pk2 f1 = pk_ashr<3>( pk_min( pk_add( f, pk_splat( 4 ) ), pk_splat( 127 ) ) );
  const pk2 f2 = pk_ashr<3>( pk_min( pk_add( f, pk_splat( 3 ) ), pk_splat( 127 ) ) );
  q0 = pk_clamp255( pk_sub( q0, f1 ) ); p0 = pk_clamp255( pk_add( p0, f2 ) );
  const pk2 g = pk_ashr<1>( pk_add( f1, pk_splat( 1 ) ) ) & ~hev;
  q1 = pk_clamp255( pk_sub( q1, g ) ); p1 = pk_clamp255( pk_add( p1, g ) );
}
// lf_macroblock on two positions
AA_MHD void lf_macroblock_pk( pk2 mask, pk2 hev, pk2 & p2, pk2 & p1, pk2 & p0, pk2 & q0, pk2 & q1, pk2 & q2 )
{
  const pk2 w = pk_sclamp( pk_add( pk_mul( pk_sub( q0, p0 ), pk_splat( 3 ) ), pk_sclamp( pk_sub( p1, q1 ) ) ) ) & mask;
  const pk2 fh = w & hev;
  const pk2 f1 = pk_ashr<3>( pk_min( pk_add( fh, pk_splat( 4 ) ), pk_splat( 127 ) ) );
  const pk2 f2 = pk_ashr<3>( pk_min( pk_add( fh, pk_splat( 3 ) ), pk_splat( 127 ) ) );
  const pk2 Q0 = pk_clamp255( pk_sub( q0, f1 ) ), P0 = pk_clamp255( pk_add( p0, f2 ) );
  const pk2 f = w & ~hev;                        // |63 + f * 27| <= 3519: the reference's clamp of u never acts
  pk2 u = pk_ashr<7>( pk_add( pk_mul( f, pk_splat( 27 ) ), pk_splat( 63 ) ) );
  q0 = pk_clamp255( pk_sub( Q0, u ) ); p0 = pk_clamp255( pk_add( P0, u ) );
  u = pk_ashr<7>( pk_add( pk_mul( f, pk_splat( 18 ) ), pk_splat( 63 ) ) );
  q1 = pk_clamp255( pk_sub( q1, u ) ); p1 = pk_clamp255( pk_add( p1, u ) );
  u = pk_ashr<7>( pk_add( pk_mul( f, pk_splat( 9 ) ), pk_splat( 63 ) ) );
  q2 = pk_clamp255( pk_sub( q2, u ) ); p2 = pk_clamp255( pk_add( p2, u ) );
}
// One edge on two positions: eight packed pixels p3..q3 across the edge, in place.
AA_MHD void lf_edge_pk( const LfParamsPk & P, bool mb_edge, pk2 gate, pk2 & p3, pk2 & p2, pk2 & p1, pk2 & p0, pk2 & q0, pk2 & q1, pk2 & q2, pk2 & q3 )
{
  pk2 mask, hev;
  lf_masks_pk( P.interior_limit, mb_edge ? P.mb_limit : P.sb_limit, P.hev_threshold, gate, p3, p2, p1, p0, q0, q1, q2, q3, mask, hev );
  if ( mb_edge ) lf_macroblock_pk( mask, hev, p2, p1, p0, q0, q1, q2 );
  else lf_subblock_pk( mask, hev, p1, p0, q0, q1 );
}

} // namespace aa
