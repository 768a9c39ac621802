// Per-stage GPU test harness (TEST INFRASTRUCTURE; built by tests/test_gpu_stages.py into tests/cpp/_build/libstage_test.so).
//
// The product's reconstruction kernels are assembled from device functions in alfalfa_amd/csrc/vp8_math.hh (per-element arithmetic)
// and alfalfa_amd/csrc/recon_inl.hh (whole 4x4 IDCTs in registers, the packed six-tap pass, packed residual addition, packed
// coefficient reads).  The kernels below run exactly those functions ON THE GPU, one stage at a time, on numbers the test supplies;
// tests/test_gpu_stages.py compares the results with the oracle's functions for the same stage (oracle/vp8_oracle.h vp8o_stage_*).
// Every entry point takes HOST pointers and moves the data itself.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <vector>

#include "../../alfalfa_amd/csrc/recon_inl.hh"

namespace {
using namespace aa;
using namespace aa::recon;

template <class T> struct DevBuf {
  T * p = nullptr; size_t n = 0;
  DevBuf( const T * host, size_t count ) : n( count ) { if ( hipMalloc( reinterpret_cast<void **>( &p ), std::max<size_t>( 1, n ) * sizeof( T ) ) != hipSuccess ) p = nullptr; else if ( host ) (void) hipMemcpy( p, host, n * sizeof( T ), hipMemcpyHostToDevice ); }
  ~DevBuf() { if ( p ) (void) hipFree( p ); }
  bool back( T * host ) const { return hipMemcpy( host, p, n * sizeof( T ), hipMemcpyDeviceToHost ) == hipSuccess; }
};
int finish() { return hipDeviceSynchronize() == hipSuccess && hipGetLastError() == hipSuccess ? 0 : 1; }

// ---- dequantise + 4x4 inverse DCT + add to a prediction (quantization.cc:95-126, transform.cc:100-137) ----
// block i: coeff[16] raster order (dense) -> idct_block_regs (both passes in registers, packed dequantisation) -> add_residual_x4
__global__ void k_residual( int n, const int16_t * coeff, const int * q, const uint8_t * pred, uint8_t * out )
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  uint32_t d[8];
  for ( int k = 0; k < 8; k++ ) d[k] = static_cast<uint16_t>( coeff[16 * i + 2 * k] ) | ( static_cast<uint32_t>( static_cast<uint16_t>( coeff[16 * i + 2 * k + 1] ) ) << 16 );
  int r[16];
  idct_block_regs( d, q[2 * i], q[2 * i + 1], false, 0, r );
  alignas( 8 ) int16_t res[16];
  for ( int k = 0; k < 16; k++ ) res[k] = static_cast<int16_t>( r[k] );
  for ( int row = 0; row < 4; row++ ) {
    uint32_t p4; memcpy( &p4, pred + 16 * i + 4 * row, 4 );
    const uint32_t o = add_residual_x4( p4, res + 4 * row );
    memcpy( out + 16 * i + 4 * row, &o, 4 );
  }
}
// the same from PACKED storage (coeff_pack.hh): words[i] = where block i's values start, mask[i] its mask word.  Whole waves
// (load_packed_block skips positions no lane of the wave has a coefficient at)
__global__ void k_residual_packed( int n, const uint32_t * mask, const uint32_t * first, const int16_t * values, const int * q, const uint8_t * pred, uint8_t * out )
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool on = i < n;
  uint32_t d[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  load_packed_block( on ? mask[i] : 0u, values + ( on ? first[i] : 0u ), d );
  if ( !on ) return;
  int r[16];
  idct_block_regs( d, q[2 * i], q[2 * i + 1], false, 0, r );
  alignas( 8 ) int16_t res[16];
  for ( int k = 0; k < 16; k++ ) res[k] = static_cast<int16_t>( r[k] );
  for ( int row = 0; row < 4; row++ ) {
    uint32_t p4; memcpy( &p4, pred + 16 * i + 4 * row, 4 );
    const uint32_t o = add_residual_x4( p4, res + 4 * row );
    memcpy( out + 16 * i + 4 * row, &o, 4 );
  }
}
// ---- Y2: dequantise + inverse Walsh-Hadamard (transform.cc:47-88), the way residual_x4 runs it: 16 lanes, two passes ----
__global__ void k_iwht( int n, const int16_t * coeff, const int * q, int16_t * out )
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  int c[16], im[16];
  for ( int l = 0; l < 16; l++ ) c[l] = static_cast<int16_t>( dequant( coeff[16 * i + l], q[2 * i + ( l ? 1 : 0 )] ) );
  for ( int l = 0; l < 4; l++ ) { const Quad v = iwht_pass1( c[l], c[l + 4], c[l + 8], c[l + 12] ); im[l] = static_cast<int16_t>( v.v0 ); im[l + 4] = static_cast<int16_t>( v.v1 ); im[l + 8] = static_cast<int16_t>( v.v2 ); im[l + 12] = static_cast<int16_t>( v.v3 ); }
  for ( int l = 0; l < 4; l++ ) {
    const int o = 4 * l;
    const Quad v = iwht_pass2( im[o], im[o + 1], im[o + 2], im[o + 3] );
    out[16 * i + o] = static_cast<int16_t>( v.v0 ); out[16 * i + o + 1] = static_cast<int16_t>( v.v1 ); out[16 * i + o + 2] = static_cast<int16_t>( v.v2 ); out[16 * i + o + 3] = static_cast<int16_t>( v.v3 );
  }
}
// ---- the ten 4x4 intra predictors, table form (what k_recon_intra4 evaluates) and switch form: E[13] as vp8_math.hh lays it out ----
__global__ void k_bpred( int n, const uint8_t * mode, const uint8_t * E, uint8_t * out_table, uint8_t * out_switch )
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  const uint8_t * e = E + 13 * i;
  int dc = 4; for ( int k = 0; k < 4; k++ ) dc += e[5 + k] + e[k];
  dc >>= 3;
  for ( int r = 0; r < 4; r++ ) for ( int c = 0; c < 4; c++ ) {
    const uint32_t t = bpred_entry( mode[i], c, r );
    out_table[16 * i + 4 * r + c] = static_cast<uint8_t>( bpred_eval( t >> 24, e[t & 0xFF], e[( t >> 8 ) & 0xFF], e[( t >> 16 ) & 0xFF], dc ) );
    out_switch[16 * i + 4 * r + c] = static_cast<uint8_t>( bpred_pixel( mode[i], e, c, r ) );
  }
}
// ---- 16x16 / 8x8 predictors (prediction.cc:385-431, 469-559): above[size], left[size], corner; interior block (both edges there) ----
__global__ void k_bigpred( int n, int size, const uint8_t * mode, const uint8_t * above, const uint8_t * left, const uint8_t * corner, uint8_t * out )
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  int sa = 0, sl = 0;
  for ( int k = 0; k < size; k++ ) { sa += above[size * i + k]; sl += left[size * i + k]; }
  const int dc = bigpred_dc( sa, sl, true, true, size == 16 ? 4 : 3 );
  for ( int r = 0; r < size; r++ ) for ( int c = 0; c < size; c++ )
    out[size * size * i + size * r + c] = static_cast<uint8_t>( bigpred_pixel( mode[i], above[size * i + c], left[size * i + r], corner[i], dc ) );
}
// ---- one six-tap pass on packed bytes (sixtap_x4_lane: v_dot4 on re-biased pixels, v_ashr_pk_u8_i32): 12 source bytes -> 4 outputs ----
__global__ void k_sixtap( int n, const uint32_t * src, const uint8_t * offset, const uint8_t * frac, uint32_t * out )
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  uint32_t t0, t1;
  pack_taps( frac[i], t0, t1 );
  out[i] = sixtap_x4_lane( src[3 * i], src[3 * i + 1], src[3 * i + 2], offset[i], frac[i], t0, t1 );
}
// ---- one loop-filter edge on two positions at once (packed int16: lf_edge_pk), limits from lf_params ----
__global__ void k_lf_edge( int n, const uint8_t * px, const uint8_t * level, const uint8_t * sharp, const uint8_t * key, const uint8_t * mb_edge, uint8_t * out )
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  const LfParamsPk P = lf_params_pk( lf_params( level[i], sharp[i], key[i] != 0 ) );
  pk2 v[8];
  for ( int k = 0; k < 8; k++ ) v[k] = static_cast<uint32_t>( px[16 * i + k] ) | ( static_cast<uint32_t>( px[16 * i + 8 + k] ) << 16 );
  lf_edge_pk( P, mb_edge[i] != 0, ~0u, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7] );
  for ( int k = 0; k < 8; k++ ) { out[16 * i + k] = static_cast<uint8_t>( v[k] & 0xFFu ); out[16 * i + 8 + k] = static_cast<uint8_t>( ( v[k] >> 16 ) & 0xFFu ); }
}

int blocks_for( int n ) { return ( n + 63 ) / 64; }
} // namespace

extern "C" {

int stage_device_count( void ) { int n = 0; return hipGetDeviceCount( &n ) == hipSuccess ? n : 0; }

int stage_residual( int n, const int16_t * coeff, const int * q, const uint8_t * pred, uint8_t * out )
{
  DevBuf<int16_t> c( coeff, size_t( n ) * 16 ); DevBuf<int> dq( q, size_t( n ) * 2 ); DevBuf<uint8_t> p( pred, size_t( n ) * 16 ), o( nullptr, size_t( n ) * 16 );
  hipLaunchKernelGGL( k_residual, dim3( blocks_for( n ) ), dim3( 64 ), 0, nullptr, n, c.p, dq.p, p.p, o.p );
  return finish() || !o.back( out );
}
int stage_residual_packed( int n, const uint32_t * mask, const uint32_t * first, const int16_t * values, int n_values, const int * q, const uint8_t * pred, uint8_t * out )
{
  DevBuf<uint32_t> m( mask, n ), f( first, n ); DevBuf<int16_t> v( values, n_values ); DevBuf<int> dq( q, size_t( n ) * 2 ); DevBuf<uint8_t> p( pred, size_t( n ) * 16 ), o( nullptr, size_t( n ) * 16 );
  hipLaunchKernelGGL( k_residual_packed, dim3( blocks_for( n ) ), dim3( 64 ), 0, nullptr, n, m.p, f.p, v.p, dq.p, p.p, o.p );
  return finish() || !o.back( out );
}
int stage_iwht( int n, const int16_t * coeff, const int * q, int16_t * out )
{
  DevBuf<int16_t> c( coeff, size_t( n ) * 16 ), o( nullptr, size_t( n ) * 16 ); DevBuf<int> dq( q, size_t( n ) * 2 );
  hipLaunchKernelGGL( k_iwht, dim3( blocks_for( n ) ), dim3( 64 ), 0, nullptr, n, c.p, dq.p, o.p );
  return finish() || !o.back( out );
}
int stage_bpred( int n, const uint8_t * mode, const uint8_t * E, uint8_t * out_table, uint8_t * out_switch )
{
  DevBuf<uint8_t> m( mode, n ), e( E, size_t( n ) * 13 ), a( nullptr, size_t( n ) * 16 ), b( nullptr, size_t( n ) * 16 );
  hipLaunchKernelGGL( k_bpred, dim3( blocks_for( n ) ), dim3( 64 ), 0, nullptr, n, m.p, e.p, a.p, b.p );
  return finish() || !a.back( out_table ) || !b.back( out_switch );
}
int stage_bigpred( int n, int size, const uint8_t * mode, const uint8_t * above, const uint8_t * left, const uint8_t * corner, uint8_t * out )
{
  DevBuf<uint8_t> m( mode, n ), a( above, size_t( n ) * size ), l( left, size_t( n ) * size ), c( corner, n ), o( nullptr, size_t( n ) * size * size );
  hipLaunchKernelGGL( k_bigpred, dim3( blocks_for( n ) ), dim3( 64 ), 0, nullptr, n, size, m.p, a.p, l.p, c.p, o.p );
  return finish() || !o.back( out );
}
int stage_sixtap( int n, const uint32_t * src, const uint8_t * offset, const uint8_t * frac, uint32_t * out )
{
  DevBuf<uint32_t> s( src, size_t( n ) * 3 ), o( nullptr, n ); DevBuf<uint8_t> of( offset, n ), fr( frac, n );
  hipLaunchKernelGGL( k_sixtap, dim3( blocks_for( n ) ), dim3( 64 ), 0, nullptr, n, s.p, of.p, fr.p, o.p );
  return finish() || !o.back( out );
}
int stage_lf_edge( int n, const uint8_t * px, const uint8_t * level, const uint8_t * sharp, const uint8_t * key, const uint8_t * mb_edge, uint8_t * out )
{
  DevBuf<uint8_t> p( px, size_t( n ) * 16 ), lv( level, n ), sh( sharp, n ), k( key, n ), mb( mb_edge, n ), o( nullptr, size_t( n ) * 16 );
  hipLaunchKernelGGL( k_lf_edge, dim3( blocks_for( n ) ), dim3( 64 ), 0, nullptr, n, p.p, lv.p, sh.p, k.p, mb.p, o.p );
  return finish() || !o.back( out );
}

} // extern "C"
