// Device-only building blocks of the reconstruction kernels (kernels.hip) that are more than per-element arithmetic (vp8_math.hh):
// whole 4x4 inverse DCTs in a lane's registers, the packed six-tap pass, packed residual addition, packed-coefficient block reads.
// In a header of their own so that tests/cpp/stage_kernels.hip can run them ON THE GPU, stage by stage, against the oracle's
// functions (tests/test_gpu_stages.py) -- a raster mismatch names a macroblock, a stage mismatch names the instruction sequence.
#pragma once
#include <hip/hip_runtime.h>

#include "vp8_math.hh"
#include "coeff_pack.hh"

namespace aa {
namespace recon {

__device__ __forceinline__ void pack_taps( const int frac, uint32_t & t0123, uint32_t & t45 )
{
  t0123 = ( sixtap_coeff( frac, 0 ) & 0xFF ) | ( ( sixtap_coeff( frac, 1 ) & 0xFF ) << 8 ) | ( ( sixtap_coeff( frac, 2 ) & 0xFF ) << 16 )
          | ( static_cast<uint32_t>( sixtap_coeff( frac, 3 ) & 0xFF ) << 24 );
  t45 = ( sixtap_coeff( frac, 4 ) & 0xFF ) | ( ( sixtap_coeff( frac, 5 ) & 0xFF ) << 8 );
}

// dequantise + inverse DCT of one 4x4 block in one lane's registers: d[8] = 16 packed int16 coefficients -> 16 residuals
__device__ __forceinline__ void idct_block_regs( const uint32_t ( &d )[8], const int fdc, const int fac, const bool replace_dc, const int dc, int ( &r )[16] )
{
  // dequantise two coefficients per instruction: the int16 wrap-around of the reference (quantization.cc:110-121) is
  // exactly the low half of the product
  const pk2 fpair = pk_splat( fac ), f0 = ( static_cast<uint32_t>( fdc ) & 0xFFFFu ) | ( static_cast<uint32_t>( fac ) << 16 );
  int c[16];
#pragma unroll
  for ( int i = 0; i < 8; i++ ) {
    const pk2 m = pk_mul( d[i], i == 0 ? f0 : fpair );
    c[2 * i] = static_cast<int16_t>( m & 0xFFFFu ); c[2 * i + 1] = static_cast<int>( m ) >> 16;
  }
  if ( replace_dc ) c[0] = dc;
  int im[16];
#pragma unroll
  for ( int i = 0; i < 4; i++ ) { const Quad v = idct_pass1( c[i], c[i + 4], c[i + 8], c[i + 12] ); im[i * 4] = v.v0; im[i * 4 + 1] = v.v1; im[i * 4 + 2] = v.v2; im[i * 4 + 3] = v.v3; }
#pragma unroll
  for ( int i = 0; i < 4; i++ ) { const Quad v = idct_pass2( im[i], im[i + 4], im[i + 8], im[i + 12] ); r[i * 4] = v.v0; r[i * 4 + 1] = v.v1; r[i * 4 + 2] = v.v2; r[i * 4 + 3] = v.v3; }
}


// inclusive prefix sum over the 16 lanes of a slot (a DPP row = 16 lanes: row_shr never crosses slots; lanes shifted in read 0)
__device__ __forceinline__ int slot_scan16( int v )
{
  v += __builtin_amdgcn_update_dpp( 0, v, 0x111, 0xF, 0xF, true );
  v += __builtin_amdgcn_update_dpp( 0, v, 0x112, 0xF, 0xF, true );
  v += __builtin_amdgcn_update_dpp( 0, v, 0x114, 0xF, 0xF, true );
  v += __builtin_amdgcn_update_dpp( 0, v, 0x118, 0xF, 0xF, true );
  return v;
}
// A stored block of PACKED coefficient storage (coeff_pack.hh) -> the 16 coefficients in raster order, two per dword (what a dense
// block holds): mask = the block's mask word, v = its first value.  One 2-byte load per coefficient that is there -- an inter
// frame's stored block holds 2-3 --, none for positions no macroblock of the wave has a coefficient at.  Called by whole waves.
__device__ __forceinline__ void load_packed_block( const uint32_t mask, const int16_t * const v, uint32_t ( &d )[8] )
{
#pragma unroll
  for ( int j = 0; j < 16; j++ ) {
    const uint32_t zz = static_cast<uint32_t>( pack::kInvZigzagNib >> ( 4 * j ) ) & 15u;         // (compile-time)
    const bool there = ( mask >> zz ) & 1u;
    if ( !__any( there ) ) continue;
    uint32_t c = 0;
    if ( there ) c = static_cast<uint16_t>( v[__popc( mask & ( ( 1u << zz ) - 1u ) )] );
    d[j >> 1] |= c << ( 16 * ( j & 1 ) );
  }
}


// sixtap_x4 with per-lane offset / fraction / taps (see sixtap_x4): d0 d1 d2 = 12 source bytes, outputs k take bytes o+k..o+k+5
__device__ __forceinline__ uint32_t sixtap_x4_lane( const uint32_t d0, const uint32_t d1, const uint32_t d2, const int o, const int frac, const uint32_t t0123, const uint32_t t45 )
{
  const uint32_t w0 = __builtin_amdgcn_alignbyte( d1, d0, o ), w1 = __builtin_amdgcn_alignbyte( d2, d1, o ), w2 = __builtin_amdgcn_alignbyte( 0u, d2, o );
  const uint32_t ident = __builtin_amdgcn_alignbyte( w1, w0, 2 );          // fraction 0: the centre tap 128 does not fit int8
  const uint32_t b0 = w0 ^ 0x80808080u, b1 = w1 ^ 0x80808080u, b2 = w2 ^ 0x80808080u;
  int v[4];
#pragma unroll
  for ( int k = 0; k < 4; k++ ) {
    const int a = static_cast<int>( __builtin_amdgcn_alignbyte( b1, b0, k ) ), b = static_cast<int>( __builtin_amdgcn_alignbyte( b2, b1, k ) );
    v[k] = __builtin_amdgcn_sdot4( a, static_cast<int>( t0123 ), __builtin_amdgcn_sdot4( b, static_cast<int>( t45 ), 16384 + 64, false ), false );
  }
  // v_ashr_pk_u8_i32 = {sat_u8(s0 >> 7), sat_u8(s1 >> 7)} in the LOW 16 bits, upper bits left alone (tools/hw_probe_pk.hip):
  // the permute picks bytes 0,1 of each pair, so nothing depends on those upper bits
  const uint32_t lo = __builtin_amdgcn_ashr_pk_u8_i32( v[0], v[1], 7 ), hi = __builtin_amdgcn_ashr_pk_u8_i32( v[2], v[3], 7 );
  const uint32_t out = __builtin_amdgcn_perm( hi, lo, 0x05040100u );
  return frac == 0 ? ident : out;
}

// four prediction pixels + four int16 residuals -> four clamped pixels, in packed int16 (v_sat_pk_u8_i16 = two clamp255)
__device__ __forceinline__ uint32_t add_residual_x4( const uint32_t p4, const int16_t * res )
{
  const uint2 r = *reinterpret_cast<const uint2 *>( res );
  const pk2 s0 = pk_add( __builtin_amdgcn_perm( 0u, p4, 0x0c010c00u ), r.x ), s1 = pk_add( __builtin_amdgcn_perm( 0u, p4, 0x0c030c02u ), r.y );
  uint32_t b0, b1;
  asm( "v_sat_pk_u8_i16 %0, %1" : "=v"( b0 ) : "v"( s0 ) );
  asm( "v_sat_pk_u8_i16 %0, %1" : "=v"( b1 ) : "v"( s1 ) );
  return b0 | ( b1 << 16 );
}


} // namespace recon
} // namespace aa
