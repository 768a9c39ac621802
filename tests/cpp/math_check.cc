// Host check of the packed (2 x int16) loop-filter arithmetic of alfalfa_amd/csrc/vp8_math.hh against the scalar
// functions of the same header (which the GPU parity tests pin to the oracle).  Built and run by tests/test_math_check.py.
#include <cstdio>
#include <cstdlib>
#include "../../alfalfa_amd/csrc/vp8_math.hh"

using namespace aa;

static uint32_t rng_state = 12345;
static uint32_t rnd() { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

int main()
{
  long checked = 0;
  for ( int level = 1; level <= 63; level++ ) for ( int sharp = 0; sharp < 8; sharp += ( level % 3 ) + 1 ) for ( int key = 0; key < 2; key++ ) {
    const LfParams P = lf_params( level, sharp, key );
    const LfParamsPk Q = lf_params_pk( P );
    for ( int it = 0; it < 1500; it++ ) {
      int px[2][8];
      const int spread = 1 << ( rnd() % 9 );
      for ( int h = 0; h < 2; h++ ) {
        const int base = rnd() & 255;
        for ( int i = 0; i < 8; i++ ) { int v = base + static_cast<int>( rnd() % ( 2 * spread + 1 ) ) - spread; px[h][i] = v < 0 ? 0 : v > 255 ? 255 : v; }
        if ( ( rnd() & 15 ) == 0 ) for ( int i = 0; i < 8; i++ ) px[h][i] = rnd() & 255;
      }
      for ( int mb = 0; mb < 2; mb++ ) for ( int g = 0; g < 4; g++ ) {
        int want[2][8];
        for ( int h = 0; h < 2; h++ ) {
          int * p = want[h];
          for ( int i = 0; i < 8; i++ ) p[i] = px[h][i];
          if ( !( ( g >> h ) & 1 ) ) continue;
          const bool mask = lf_mask( P.interior_limit, mb ? P.mb_limit : P.sb_limit, p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7] );
          const bool hev = lf_hev( P.hev_threshold, p[2], p[3], p[4], p[5] );
          if ( mb ) lf_macroblock( mask, hev, p[1], p[2], p[3], p[4], p[5], p[6] );
          else lf_subblock( mask, hev, p[2], p[3], p[4], p[5] );
        }
        pk2 v[8];
        for ( int i = 0; i < 8; i++ ) v[i] = static_cast<uint32_t>( px[0][i] ) | ( static_cast<uint32_t>( px[1][i] ) << 16 );
        const pk2 gate = ( ( g & 1 ) ? 0xFFFFu : 0u ) | ( ( g & 2 ) ? 0xFFFF0000u : 0u );
        lf_edge_pk( Q, mb != 0, gate, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7] );
        for ( int i = 0; i < 8; i++ ) {
          if ( static_cast<int>( v[i] & 0xFFFFu ) != want[0][i] || static_cast<int>( v[i] >> 16 ) != want[1][i] ) {
            std::printf( "MISMATCH level %d sharp %d key %d mb %d gate %d pos %d: got %u,%u want %d,%d\n", level, sharp, key, mb, g, i,
                         v[i] & 0xFFFFu, v[i] >> 16, want[0][i], want[1][i] );
            return 1;
          }
        }
        checked++;
      }
    }
  }
  for ( int it = 0; it < 100000; it++ ) {          // byte <-> packed-halves shuffles
    const uint32_t a = rnd() * 2654435761u, b = rnd() * 40503u + it;
    const pk2 x0 = pk_from_bytes<0>( a, b ), x1 = pk_from_bytes<1>( a, b ), x2 = pk_from_bytes<2>( a, b ), x3 = pk_from_bytes<3>( a, b );
    if ( x0 != ( ( a & 0xFFu ) | ( ( b & 0xFFu ) << 16 ) ) || x3 != ( ( a >> 24 ) | ( ( b >> 24 ) << 16 ) ) ) { std::printf( "pk_from_bytes\n" ); return 1; }
    uint32_t ra, rb;
    pk_to_dwords( x0, x1, x2, x3, ra, rb );
    if ( ra != a || rb != b ) { std::printf( "pk_to_dwords\n" ); return 1; }
    const uint32_t w = a & 0xFFFFu;
    if ( pk_from_u16( w ) != ( ( w & 0xFFu ) | ( ( w >> 8 ) << 16 ) ) || pk_to_u16( pk_from_u16( w ) ) != w ) { std::printf( "pk_u16\n" ); return 1; }
  }
  for ( int it = 0; it < 20000; it++ ) {            // table form of the ten 4x4 predictors == the switch form
    uint8_t E[13];
    for ( int i = 0; i < 13; i++ ) E[i] = rnd() & 255;
    int dc = 4; for ( int i = 0; i < 4; i++ ) dc += E[5 + i] + E[i];
    dc >>= 3;
    for ( int mode = 0; mode < 10; mode++ ) for ( int r = 0; r < 4; r++ ) for ( int c = 0; c < 4; c++ ) {
      const uint32_t e = bpred_entry( mode, c, r );
      const int got = bpred_eval( e >> 24, E[e & 0xFF], E[( e >> 8 ) & 0xFF], E[( e >> 16 ) & 0xFF], dc );
      if ( ( e & 0xFF ) > 12 || ( ( e >> 8 ) & 0xFF ) > 12 || ( ( e >> 16 ) & 0xFF ) > 12 || got != bpred_pixel( mode, E, c, r ) ) {
        std::printf( "bpred table: mode %d r %d c %d got %d want %d\n", mode, r, c, got, bpred_pixel( mode, E, c, r ) ); return 1; }
    }
  }
  std::printf( "OK %ld edge pairs\n", checked );
  return 0;
}
