// Hardware probe (not product code): what a wave pays for draining its stores (s_waitcnt vmcnt(0)) when its lanes write small
// pieces into chunks scattered over a large heap -- by KIND of memory: many 256-MB hipMalloc slabs, one big hipMalloc, memory
// mapped into a reserved virtual range (hipMemCreate / hipMemMap).  The token workers' lanes do exactly this.
//   hipcc --offload-arch=gfx950 -O3 tools/hw_probe_stores.hip -o /tmp/probe_stores && /tmp/probe_stores
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK( x ) do { hipError_t e = ( x ); if ( e != hipSuccess ) { printf( "%s: %s\n", #x, hipGetErrorString( e ) ); exit( 1 ); } } while ( 0 )

// every lane < lanes: `iters` 2-byte stores walking through its own 64-KB chunk (+ a 32-byte store every 16); drain every `drain` iterations
__global__ __launch_bounds__( 64 ) void k_probe( uint8_t * const * chunk_of, int lanes, int iters, int drain, unsigned long long * cycles )
{
  const int lane = threadIdx.x;
  if ( lane >= lanes ) return;
  uint8_t * p = chunk_of[blockIdx.x * lanes + lane];
  const unsigned long long t0 = wall_clock64();
  for ( int i = 0; i < iters; i++ ) {
    *reinterpret_cast<volatile uint16_t *>( p + ( ( i * 2 ) & 65535 ) ) = static_cast<uint16_t>( i );
    if ( ( i & 15 ) == 0 ) { *reinterpret_cast<volatile unsigned long long *>( p + ( ( i * 2 + 32 ) & 65520 ) ) = 0ull; *reinterpret_cast<volatile unsigned long long *>( p + ( ( i * 2 + 40 ) & 65528 ) ) = 0ull; }
    if ( drain && ( i % drain ) == drain - 1 ) asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
    // ~70 dependent ALU operations stand for the decode step between stores
    unsigned x = i;
    #pragma unroll
    for ( int k = 0; k < 35; k++ ) x = x * 1664525u + 1013904223u;
    if ( x == 0x12345u ) p[1] = 1;
  }
  asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
  if ( lane == 0 ) cycles[blockIdx.x] = wall_clock64() - t0;
}

int main()
{
  const int waves = 1024, lanes = 26, iters = 20000;
  const size_t chunk = 65536, heap_bytes = size_t( 96 ) << 30;
  const size_t nchunks = size_t( waves ) * lanes;
  std::vector<uint8_t *> ptr( nchunks );
  uint8_t ** d_ptr; unsigned long long * d_cyc;
  CK( hipMalloc( &d_ptr, nchunks * sizeof( uint8_t * ) ) ); CK( hipMalloc( &d_cyc, waves * 8 ) );
  auto run = [&]( const char * what ) {
    CK( hipMemcpy( d_ptr, ptr.data(), nchunks * sizeof( uint8_t * ), hipMemcpyHostToDevice ) );
    for ( int drain : { 0, 16, 64 } ) {
      hipLaunchKernelGGL( k_probe, dim3( waves ), dim3( 64 ), 0, 0, d_ptr, lanes, iters, drain, d_cyc );
      CK( hipDeviceSynchronize() );
      std::vector<unsigned long long> c( waves );
      CK( hipMemcpy( c.data(), d_cyc, waves * 8, hipMemcpyDeviceToHost ) );
      double s = 0; for ( auto v : c ) s += v;
      printf( "%-40s drain every %2d: %.3f us per iteration (100 MHz clock)\n", what, drain, s / waves / iters / 100.0 );
    }
  };
  // (a) 256-MB slabs, chunks scattered over 384 of them (96 GB)
  {
    std::vector<uint8_t *> slabs( 384 );
    for ( auto & s : slabs ) CK( hipMalloc( &s, size_t( 256 ) << 20 ) );
    for ( size_t i = 0; i < nchunks; i++ ) ptr[i] = slabs[( i * 131 ) % slabs.size()] + ( ( i * 7919 ) % 4096 ) * chunk;
    run( "256-MB hipMalloc slabs (96 GB)" );
    for ( auto s : slabs ) CK( hipFree( s ) );
  }
  // (b) one hipMalloc of 96 GB
  {
    uint8_t * big; CK( hipMalloc( &big, heap_bytes ) );
    for ( size_t i = 0; i < nchunks; i++ ) ptr[i] = big + ( ( i * 1000003 ) % ( heap_bytes / chunk ) ) * chunk;
    run( "one 96-GB hipMalloc" );
    CK( hipFree( big ) );
  }
  // (c) reserved range, 1-GB handles mapped; (d) the same with 1-GB aligned virtual range
  for ( size_t align : { size_t( 0 ), size_t( 1 ) << 30 } ) {
    void * va = nullptr;
    if ( hipMemAddressReserve( &va, heap_bytes, align, nullptr, 0 ) != hipSuccess ) { printf( "hipMemAddressReserve failed (align %zu)\n", align ); continue; }
    hipMemAllocationProp prop {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0; CK( hipMemGetAllocationGranularity( &gran, &prop, hipMemAllocationGranularityRecommended ) );
    std::vector<hipMemGenericAllocationHandle_t> hs;
    const size_t piece = size_t( 1 ) << 30;
    for ( size_t off = 0; off < heap_bytes; off += piece ) {
      hipMemGenericAllocationHandle_t h; CK( hipMemCreate( &h, piece, &prop, 0 ) );
      CK( hipMemMap( static_cast<uint8_t *>( va ) + off, piece, 0, h, 0 ) );
      hs.push_back( h );
    }
    hipMemAccessDesc acc {}; acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK( hipMemSetAccess( va, heap_bytes, &acc, 1 ) );
    for ( size_t i = 0; i < nchunks; i++ ) ptr[i] = static_cast<uint8_t *>( va ) + ( ( i * 1000003 ) % ( heap_bytes / chunk ) ) * chunk;
    char name[96]; snprintf( name, sizeof name, "hipMemMap 1-GB pieces, va align %zu MB, gran %zu KB", align >> 20, gran >> 10 );
    run( name );
    CK( hipMemUnmap( va, heap_bytes ) );
    for ( auto h : hs ) CK( hipMemRelease( h ) );
    CK( hipMemAddressFree( va, heap_bytes ) );
  }
  return 0;
}
