// Hardware probe (not product code): what a token step costs when the lane's probability comes out of a per-lane table in
// GLOBAL memory (L1 / L2) instead of the lane's LDS slice -- the question behind "64 chains per wave" (VERDICT round 3 item 4):
// the 1056-byte table is what bounds a wave to 22 lanes.  The probe keeps the shape of the real step: one dependent chain per
// lane, ~120 VALU instructions per step, the NEXT table address known only at the end of a step, every lane in its own table.
//   mode 0  table in LDS (today; only fits <= 24 lanes)        mode 1  global_load_ubyte, one per step, on the chain
//   mode 2  ... + a scattered 2-byte store every 4th step (the packed coefficient store as it is today)
//   mode 3  ... stores staged in LDS instead, flushed as 16-byte stores every 32 steps (all lanes together)
//   mode 4  mode 3 with nontemporal loads
//   mode 5  the current ROW of the table (16 bytes) in registers, loaded when a token ends (every ~4th step, per lane), the step's
//           probability picked out of it by arithmetic; stores staged as in mode 3 -- what a design with fewer, wider loads costs
//   mode 6  mode 5 with the row load issued one step AHEAD of its first use (the row of the next position is known when a
//           token's last bool is about to be decoded -- speculation on the common successor)
//   hipcc --offload-arch=gfx950 -O3 tools/hw_probe_probs.hip -o /tmp/probe_probs && /tmp/probe_probs
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK( x ) do { hipError_t e = ( x ); if ( e != hipSuccess ) { printf( "%s: %s\n", #x, hipGetErrorString( e ) ); exit( 1 ); } } while ( 0 )

constexpr uint32_t kTable = 1088;      // 1056 probabilities + the constant extra-bit ones, padded
constexpr uint32_t kStage = 64;        // bytes of LDS staging per lane (mode 3, 4)

__device__ inline uint32_t work( uint32_t x, uint32_t prob )
{
  // ~110 dependent single-rate VALU operations standing for refill, split, compare, renormalise, record decode
  #pragma unroll
  for ( int k = 0; k < 36; k++ ) x = ( ( x ^ ( x >> 7 ) ) + prob ) ^ ( x << 3 );
  return x;
}

template <int MODE>
__global__ __launch_bounds__( 64 ) void k_probe( const uint8_t * tables, uint8_t * out, int lanes, int iters, unsigned long long * cycles, uint32_t * sink )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) uint8_t smem[];
  const int lane = threadIdx.x;
  if ( lane >= lanes ) return;
  const uint32_t slot = blockIdx.x * 64u + lane;
  const uint8_t * tab = tables + size_t( slot ) * kTable;
  uint8_t * dst = out + size_t( slot ) * 65536u;
  uint32_t lds_base = 0;
  if ( MODE == 0 ) { lds_base = lane * kTable; for ( uint32_t k = 0; k < kTable; k += 4 ) *reinterpret_cast<uint32_t *>( smem + lds_base + k ) = *reinterpret_cast<const uint32_t *>( tab + k ); }
  if ( MODE >= 3 ) lds_base = lane * kStage;
  uint32_t x = slot * 2654435761u, paddr = x % 1056u, wpos = 0, flushed = 0;
  uint4 row = *reinterpret_cast<const uint4 *>( tab + ( paddr & ~15u ) ), next_row = row;
  const unsigned long long t0 = wall_clock64();
  for ( int i = 0; i < iters; i++ ) {
    uint32_t prob;
    if ( MODE == 5 || MODE == 6 ) {
      // byte (paddr & 15) of the row in registers: select the dword, shift
      const uint32_t k = paddr & 15u;
      const uint32_t w = k < 8u ? ( k < 4u ? row.x : row.y ) : ( k < 12u ? row.z : row.w );
      prob = ( w >> ( 8u * ( k & 3u ) ) ) & 255u;
    } else
    if ( MODE == 0 ) prob = smem[lds_base + paddr];
    else if ( MODE == 4 ) prob = __builtin_nontemporal_load( tab + paddr );
    else prob = tab[paddr];
    x = work( x, prob );
    paddr = ( x >> 8 ) % 1056u;                      // where the next probability is: known only now
    const bool emit = ( x & 3u ) == 0u;              // a coefficient every ~4th step
    if ( MODE == 5 ) { if ( emit ) row = *reinterpret_cast<const uint4 *>( tab + ( paddr & ~15u ) ); }       // the token ended: the next position's row
    if ( MODE == 6 ) { if ( emit ) row = next_row; if ( ( ( x >> 2 ) & 3u ) == 0u ) next_row = *reinterpret_cast<const uint4 *>( tab + ( paddr & ~15u ) ); }
    if ( MODE == 2 ) { if ( emit ) { *reinterpret_cast<volatile uint16_t *>( dst + ( wpos & 65534u ) ) = static_cast<uint16_t>( x ); wpos += 2; } }
    if ( MODE >= 3 ) {
      if ( emit ) { *reinterpret_cast<uint16_t *>( smem + lds_base + ( wpos & ( kStage - 1 ) ) ) = static_cast<uint16_t>( x ); wpos += 2; }
      if ( ( i & 31 ) == 31 ) {
        // all lanes together: whole 16-byte pieces out
        while ( __builtin_amdgcn_ballot_w64( wpos - flushed >= 16u ) ) {
          if ( wpos - flushed >= 16u ) {
            const uint4 v = *reinterpret_cast<const uint4 *>( smem + lds_base + ( flushed & ( kStage - 1 ) ) );
            *reinterpret_cast<uint4 *>( dst + ( flushed & 65520u ) ) = v;
            flushed += 16;
          }
        }
      }
    }
  }
  asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
  if ( lane == 0 ) cycles[blockIdx.x] = wall_clock64() - t0;
  if ( x == 0x12345u ) sink[0] = x;
}

template <int MODE>
static void run( const char * what, const uint8_t * tables, uint8_t * out, unsigned long long * d_cyc, uint32_t * sink, int waves, int lanes, int iters )
{
  const uint32_t lds = MODE == 0 ? lanes * kTable : MODE >= 3 ? 64 * kStage : 0;
  if ( lds > 65536 ) return;
  // the real workers ask for 33 KB of LDS per workgroup so that exactly four fit a CU: do the same
  const uint32_t request = lds > 33280u ? lds : 33280u;
  hipLaunchKernelGGL( k_probe<MODE>, dim3( waves ), dim3( 64 ), request, 0, tables, out, lanes, iters, d_cyc, sink );
  CK( hipDeviceSynchronize() );
  std::vector<unsigned long long> c( waves );
  CK( hipMemcpy( c.data(), d_cyc, waves * 8, hipMemcpyDeviceToHost ) );
  double s = 0; for ( auto v : c ) s += v;
  const double us = s / waves / iters / 100.0;
  printf( "%-62s waves %4d lanes %2d: %.4f us per step -> %.1f G steps/s\n", what, waves, lanes, us, waves * double( lanes ) / us / 1e3 );
}

int main()
{
  const int iters = 20000;
  const int max_waves = 1024;
  uint8_t * tables, * out; unsigned long long * d_cyc; uint32_t * sink;
  CK( hipMalloc( &tables, size_t( max_waves ) * 64 * kTable ) );
  CK( hipMalloc( &out, size_t( max_waves ) * 64 * 65536 ) );
  CK( hipMalloc( &d_cyc, max_waves * 8 ) ); CK( hipMalloc( &sink, 64 ) );
  { std::vector<uint8_t> h( size_t( max_waves ) * 64 * kTable ); for ( size_t i = 0; i < h.size(); i++ ) h[i] = static_cast<uint8_t>( 1 + ( i * 2654435761u >> 13 ) % 255 ); CK( hipMemcpy( tables, h.data(), h.size(), hipMemcpyHostToDevice ) ); }
  const bool only_rows = getenv( "PROBE_ROWS_ONLY" ) != nullptr;
  for ( int waves : { 1024, 16 } ) {
    for ( int lanes : { 1, 22, 32, 48, 64 } ) {
      run<5>( "5 16-byte row per token in registers, staged stores", tables, out, d_cyc, sink, waves, lanes, iters );
      run<6>( "6 as 5, row loaded one step ahead", tables, out, d_cyc, sink, waves, lanes, iters );
      if ( only_rows ) { run<0>( "0 table in LDS", tables, out, d_cyc, sink, waves, lanes, iters ); continue; }
      run<0>( "0 table in LDS", tables, out, d_cyc, sink, waves, lanes, iters );
      run<1>( "1 global_load_ubyte per step", tables, out, d_cyc, sink, waves, lanes, iters );
      run<2>( "2 ... + scattered 2-byte store every 4th step", tables, out, d_cyc, sink, waves, lanes, iters );
      run<3>( "3 ... stores staged in LDS, 16-byte flushes every 32 steps", tables, out, d_cyc, sink, waves, lanes, iters );
      run<4>( "4 as 3, nontemporal loads", tables, out, d_cyc, sink, waves, lanes, iters );
    }
  }
  return 0;
}
