// Hardware probe (gfx950): checks v_alignbyte / v_dot4_i32_i8 primitives and the packed six-tap routine of kernels.hip against a
// scalar host model on random inputs.  This is how the hipcc 7.2 issue around v_ashr_pk_u8_i32 (a pair of clamp255(x >> 7)
// folded into one instruction whose upper 16 result bits are then assumed zero) was isolated; see sixtap_x4 in kernels.hip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/hw_probe_sixtap.hip -o /tmp/probe && /tmp/probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../alfalfa_amd/csrc/vp8_math.hh"
using namespace aa;
__device__ __forceinline__ void pack_taps( const int frac, uint32_t & t0123, uint32_t & t45 )
{
  t0123 = ( sixtap_coeff( frac, 0 ) & 0xFF ) | ( ( sixtap_coeff( frac, 1 ) & 0xFF ) << 8 ) | ( ( sixtap_coeff( frac, 2 ) & 0xFF ) << 16 )
          | ( static_cast<uint32_t>( sixtap_coeff( frac, 3 ) & 0xFF ) << 24 );
  t45 = ( sixtap_coeff( frac, 4 ) & 0xFF ) | ( ( sixtap_coeff( frac, 5 ) & 0xFF ) << 8 );
}
__device__ __forceinline__ uint32_t bytes_at( const uint32_t d0, const uint32_t d1, const uint32_t d2, const int s )
{
  const int q = s >> 2, sh = s & 3;
  const uint32_t lo = q == 0 ? d0 : ( q == 1 ? d1 : d2 );
  const uint32_t hi = q == 0 ? d1 : ( q == 1 ? d2 : 0u );
  return __builtin_amdgcn_alignbyte( hi, lo, sh );
}
// outputs k = 0..3 take source bytes o+k .. o+k+5 (o = 0..3) of the 12-byte string d0 d1 d2; frac 0 = identity (its
// centre tap 128 does not fit int8).  sum(taps) = 128, so sum t_i*p_i = sum t_i*(p_i - 128) + 16384: the pixels are
// re-biased to signed bytes with one xor per dword and each output is two v_dot4_i32_i8 on taps packed as signed bytes.
// NOTE: hipcc 7.2 folds a pair of `clamp255( x >> 7 )` into v_ashr_pk_u8_i32 and then assumes the upper 16 result bits
// are zero; on gfx950 they are not (found with tools/_t6.hip on hardware) -- the empty asm keeps shift and clamp apart.
__device__ __forceinline__ uint32_t sixtap_x4( uint32_t d0, uint32_t d1, uint32_t d2, const int o, const int frac, const uint32_t t0123, const uint32_t t45 )
{
  if ( frac == 0 ) return bytes_at( d0, d1, d2, o + 2 );
  d0 ^= 0x80808080u; d1 ^= 0x80808080u; d2 ^= 0x80808080u;
  uint32_t out = 0;
#pragma unroll
  for ( int k = 0; k < 4; k++ ) {
    const int a = static_cast<int>( bytes_at( d0, d1, d2, o + k ) ), b = static_cast<int>( bytes_at( d0, d1, d2, o + k + 4 ) );
    int v = __builtin_amdgcn_sdot4( a, static_cast<int>( t0123 ), __builtin_amdgcn_sdot4( b, static_cast<int>( t45 ), 16384 + 64, false ), false ) >> 7;
    asm volatile( "" : "+v"( v ) );
    out |= static_cast<uint32_t>( clamp255( v ) ) << ( 8 * k );
  }
  return out;
}
__global__ void k( const uint32_t * in, uint32_t * out, int n )
{
  int i = blockIdx.x * blockDim.x + threadIdx.x; if ( i >= n ) return;
  uint32_t d0 = in[i*4], d1 = in[i*4+1], d2 = in[i*4+2]; int o = in[i*4+3] & 3, frac = ( in[i*4+3] >> 2 ) & 7;
  // raw probes
  out[i*8+0] = bytes_at( d0, d1, d2, o + 2 );
  out[i*8+1] = bytes_at( d0, d1, d2, o + 6 );
  out[i*8+2] = bytes_at( d0, d1, d2, o + 7 );
  uint32_t t0, t1; pack_taps( frac, t0, t1 );
  out[i*8+3] = t0; out[i*8+4] = t1;
  out[i*8+5] = __builtin_amdgcn_sdot4( (int) d0, (int) t0, 100, false );
  out[i*8+6] = __builtin_amdgcn_sdot4( (int) d1, (int) t1, 16448, false );
  out[i*8+7] = sixtap_x4( d0, d1, d2, o, frac, t0, t1 );
}
int main() {
  const int n = 4096; std::vector<uint32_t> h( n * 4 ), o( n * 8 );
  srand( 2 ); for ( auto & x : h ) x = ( (uint32_t) rand() << 16 ) ^ rand();
  uint32_t * di, * dout; hipMalloc( &di, n * 16 ); hipMalloc( &dout, n * 32 ); hipMemcpy( di, h.data(), n * 16, hipMemcpyHostToDevice );
  hipLaunchKernelGGL( k, dim3( n / 64 ), dim3( 64 ), 0, 0, di, dout, n ); hipMemcpy( o.data(), dout, n * 32, hipMemcpyDeviceToHost );
  auto ab = []( uint32_t hi, uint32_t lo, int sh ) { uint64_t v = ( (uint64_t) hi << 32 ) | lo; return (uint32_t) ( v >> ( 8 * sh ) ); };
  auto ba = [&]( uint32_t d0, uint32_t d1, uint32_t d2, int s ) { int q = s >> 2, sh = s & 3; uint32_t lo = q == 0 ? d0 : ( q == 1 ? d1 : d2 ), hi = q == 0 ? d1 : ( q == 1 ? d2 : 0u ); return ab( hi, lo, sh ); };
  auto sd = []( uint32_t a, uint32_t b, int c ) { int s = c; for ( int i = 0; i < 4; i++ ) s += (int) (int8_t) ( a >> ( 8 * i ) ) * (int) (int8_t) ( b >> ( 8 * i ) ); return s; };
  int bad[8] = { 0 };
  for ( int i = 0; i < n; i++ ) {
    uint32_t d0 = h[i*4], d1 = h[i*4+1], d2 = h[i*4+2]; int oo = h[i*4+3] & 3, frac = ( h[i*4+3] >> 2 ) & 7;
    uint32_t t0 = 0, t1 = 0; for ( int t = 0; t < 4; t++ ) t0 |= (uint32_t) ( sixtap_coeff( frac, t ) & 0xFF ) << ( 8 * t ); t1 = ( sixtap_coeff( frac, 4 ) & 0xFF ) | ( ( sixtap_coeff( frac, 5 ) & 0xFF ) << 8 );
    uint32_t want[7] = { ba( d0, d1, d2, oo + 2 ), ba( d0, d1, d2, oo + 6 ), ba( d0, d1, d2, oo + 7 ), t0, t1, (uint32_t) sd( d0, t0, 100 ), (uint32_t) sd( d1, t1, 16448 ) };
    { uint8_t pb[12]; for ( int b = 0; b < 4; b++ ) { pb[b] = d0 >> ( 8 * b ); pb[4 + b] = d1 >> ( 8 * b ); pb[8 + b] = d2 >> ( 8 * b ); }
      uint32_t w = 0; for ( int kk = 0; kk < 4; kk++ ) { int ssum = 64; for ( int t = 0; t < 6; t++ ) ssum += pb[oo + kk + t] * sixtap_coeff( frac, t ); int v = ssum >> 7; v = v < 0 ? 0 : ( v > 255 ? 255 : v ); w |= (uint32_t) v << ( 8 * kk ); }
      if ( o[i*8+7] != w ) { if ( bad[7] < 5 ) printf( "sixtap_x4: o=%d frac=%d got %08x want %08x\n", oo, frac, o[i*8+7], w ); bad[7]++; } }
    for ( int j = 0; j < 7; j++ ) if ( o[i*8+j] != want[j] ) { if ( bad[j] < 3 ) printf( "probe %d: i=%d o=%d frac=%d got %08x want %08x (d=%08x %08x %08x)\n", j, i, oo, frac, o[i*8+j], want[j], d0, d1, d2 ); bad[j]++; }
  }
  for ( int j = 0; j < 8; j++ ) printf( "probe %d bad %d\n", j, bad[j] );
}
