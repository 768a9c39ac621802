// Hardware semantics probe (gfx950): v_ashr_pk_u8_i32 and v_sat_pk_u8_i16 -- which source lands in which byte, what the
// upper 16 result bits hold.   hipcc --offload-arch=gfx950 -O2 -o /tmp/p tools/hw_probe_pk.hip && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k( const int * in, unsigned * out )
{
  const int t = threadIdx.x;
  const int a = in[2 * t], b = in[2 * t + 1];
  unsigned r, s;
  asm volatile( "v_mov_b32 %0, 0xdeadbeef\n\tv_ashr_pk_u8_i32 %0, %1, %2, 7" : "=&v"( r ) : "v"( a ), "v"( b ) );
  const unsigned pk = ( static_cast<unsigned>( a ) & 0xFFFFu ) | ( static_cast<unsigned>( b ) << 16 );
  asm volatile( "v_mov_b32 %0, 0xdeadbeef\n\tv_sat_pk_u8_i16 %0, %1" : "=&v"( s ) : "v"( pk ) );
  out[2 * t] = r; out[2 * t + 1] = s;
}
int main()
{
  const int n = 8;
  int h[2 * n] = { 128 * 5 + 64, 128 * 9, -300, 128 * 300, 128 * 255 + 127, 128 * 256, 0, -1, 200, -200, 32767, -32768, 70000, 65, 255 * 128, 1 * 128 };
  int * d; unsigned * o; unsigned r[2 * n];
  hipMalloc( &d, sizeof h ); hipMalloc( &o, sizeof r );
  hipMemcpy( d, h, sizeof h, hipMemcpyHostToDevice );
  hipLaunchKernelGGL( k, dim3( 1 ), dim3( n ), 0, 0, d, o );
  hipMemcpy( r, o, sizeof r, hipMemcpyDeviceToHost );
  for ( int i = 0; i < n; i++ ) std::printf( "a=%d b=%d  ashr_pk(a,b,7)=%08x   sat_pk(lo=a16,hi=b16)=%08x\n", h[2 * i], h[2 * i + 1], r[2 * i], r[2 * i + 1] );
  return 0;
}
