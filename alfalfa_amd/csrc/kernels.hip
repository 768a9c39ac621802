// HIP kernels (gfx950 / CDNA4) for the VP8 reconstruction hot path: one 64-lane wavefront per macroblock.
//
//   k_recon_inter   all inter-coded MBs of a batch of frames: six-tap motion compensation from LDS-staged,
//                   coordinate-clamped reference windows + dequant / iWHT / IDCT residual    (macroblock.cc:553-601)
//   k_recon_intra   intra MBs of one 2:1 anti-diagonal (col + 2*row == d): neighbours (left, above, above-left,
//                   above-right) are final when the diagonal is launched                          (macroblock.cc:523-551)
//   k_loopfilter    normal loop filter of one 2:1 anti-diagonal, in place on the output raster   (loopfilter.cc:133-154)
//
// No MFMA: nothing here is a dense contraction.  The path is integer stencil/gather work bounded by HBM traffic
// and by the raster-order dependency chains; every kernel stages its working set in LDS and touches each
// global byte once.  Batches of independent frames (streams / GOPs) fill the chip: grid.y = frame in batch.
#include <hip/hip_runtime.h>

#include "device_types.h"
#include "vp8_math.hh"

namespace aa {
namespace {

constexpr int kLanes = 64;

enum : int { DC_PRED, V_PRED, H_PRED, TM_PRED, B_PRED, NEARESTMV, NEARMV, ZEROMV, NEWMV, SPLITMV };

struct alignas( 16 ) ResidualLds {
  int16_t cf[25][16];    // dense, dequantised coefficients (block 24 = Y2)
  int16_t im[24][16];    // first-pass results (int16, Q5)
  int16_t res[24][16];   // residual to add: block b, [row*4+col]
  uint8_t map[32];       // rank of stored block -> dense block id
};

// Dequantise the MB's stored blocks, run iWHT (if Y2) and the 24 IDCTs; leaves residuals in L.res.
// Macroblock::apply_walsh / DCTCoefficients::{dequantize,iwht,idct_add}.  Called by the whole wave.
__device__ void compute_residual( const aa_mb_info & mb, const aa_dev_frame & f, ResidualLds & L, const int lane )
{
  uint32_t * z = reinterpret_cast<uint32_t *>( &L.cf[0][0] );
  for ( int i = lane; i < 200; i += kLanes ) z[i] = 0;
  const uint32_t mask = mb.nz_mask;
  if ( lane < 25 && ( ( mask >> lane ) & 1u ) ) L.map[__popc( mask & ( ( 1u << lane ) - 1u ) )] = static_cast<uint8_t>( lane );
  __syncthreads();
  const int n = __popc( mask ) * 16;
  const int16_t * src = f.coeffs + static_cast<size_t>( mb.coeff_index ) * 16;
  const uint16_t * q = f.quant[mb.segment_id & 3];
  for ( int k = lane; k < n; k += kLanes ) {
    const int blk = L.map[k >> 4], e = k & 15;
    const int base = blk < 16 ? 0 : ( blk < 24 ? 4 : 2 );          // {y_dc,y_ac,y2_dc,y2_ac,uv_dc,uv_ac}
    L.cf[blk][e] = static_cast<int16_t>( dequant( src[k], q[base + ( e ? 1 : 0 )] ) );
  }
  __syncthreads();
  if ( mb.flags & AA_MB_HAS_Y2 ) {
    if ( lane < 4 ) {
      const Quad v = iwht_pass1( L.cf[24][lane], L.cf[24][lane + 4], L.cf[24][lane + 8], L.cf[24][lane + 12] );
      L.im[0][lane] = static_cast<int16_t>( v.v0 ); L.im[0][lane + 4] = static_cast<int16_t>( v.v1 );
      L.im[0][lane + 8] = static_cast<int16_t>( v.v2 ); L.im[0][lane + 12] = static_cast<int16_t>( v.v3 );
    }
    __syncthreads();
    if ( lane < 4 ) {
      const int o = lane * 4;
      const Quad v = iwht_pass2( L.im[0][o], L.im[0][o + 1], L.im[0][o + 2], L.im[0][o + 3] );
      L.cf[o + 0][0] = static_cast<int16_t>( v.v0 ); L.cf[o + 1][0] = static_cast<int16_t>( v.v1 );
      L.cf[o + 2][0] = static_cast<int16_t>( v.v2 ); L.cf[o + 3][0] = static_cast<int16_t>( v.v3 );
    }
    __syncthreads();
  }
  for ( int t = lane; t < 96; t += kLanes ) {           // 24 blocks x 4 columns
    const int blk = t >> 2, i = t & 3;
    const Quad v = idct_pass1( L.cf[blk][i], L.cf[blk][i + 4], L.cf[blk][i + 8], L.cf[blk][i + 12] );
    L.im[blk][i * 4 + 0] = static_cast<int16_t>( v.v0 ); L.im[blk][i * 4 + 1] = static_cast<int16_t>( v.v1 );
    L.im[blk][i * 4 + 2] = static_cast<int16_t>( v.v2 ); L.im[blk][i * 4 + 3] = static_cast<int16_t>( v.v3 );
  }
  __syncthreads();
  for ( int t = lane; t < 96; t += kLanes ) {           // 24 blocks x 4 rows
    const int blk = t >> 2, i = t & 3;
    const Quad v = idct_pass2( L.im[blk][i], L.im[blk][i + 4], L.im[blk][i + 8], L.im[blk][i + 12] );
    L.res[blk][i * 4 + 0] = static_cast<int16_t>( v.v0 ); L.res[blk][i * 4 + 1] = static_cast<int16_t>( v.v1 );
    L.res[blk][i * 4 + 2] = static_cast<int16_t>( v.v2 ); L.res[blk][i * 4 + 3] = static_cast<int16_t>( v.v3 );
  }
  __syncthreads();
}

__device__ __forceinline__ int clampi( int v, int lo, int hi ) { return v < lo ? lo : ( v > hi ? hi : v ); }

__device__ __forceinline__ void load_taps( int frac, int ( &t )[6] )
{
  for ( int i = 0; i < 6; i++ ) t[i] = sixtap_coeff( frac, i );
}

// Motion-compensated prediction of one NxN block into LDS `dst` (row-major, stride N).
// VP8Raster::Block<N>::inter_predict (prediction.cc:655-674): every fetch is clamped to the padded plane, which
// equals the interior ("unsafe") path when the footprint is inside and EdgeExtendedRaster::at otherwise.
template <int N>
__device__ void mc_block( const uint8_t * __restrict__ ref, const int pw, const int ph, const int x0, const int y0,
                          const int mvx, const int mvy, uint8_t * win, uint8_t * im, uint8_t * dst, const int lane )
{
  const int sx = x0 + ( mvx >> 3 ), sy = y0 + ( mvy >> 3 );   // arithmetic shift (Q9)
  const int mx = mvx & 7, my = mvy & 7;
  if ( ( mx | my ) == 0 ) {
    for ( int i = lane; i < N * N; i += kLanes ) {
      const int r = i / N, c = i % N;
      dst[i] = ref[static_cast<size_t>( clampi( sy + r, 0, ph - 1 ) ) * pw + clampi( sx + c, 0, pw - 1 )];
    }
    __syncthreads();
    return;
  }
  constexpr int W = N + 5;
  for ( int i = lane; i < W * W; i += kLanes ) {
    const int r = i / W, c = i % W;
    win[i] = ref[static_cast<size_t>( clampi( sy - 2 + r, 0, ph - 1 ) ) * pw + clampi( sx - 2 + c, 0, pw - 1 )];
  }
  int hf[6], vf[6];
  load_taps( mx, hf ); load_taps( my, vf );
  __syncthreads();
  for ( int i = lane; i < W * N; i += kLanes ) {            // horizontal pass over N+5 rows, clamp to u8 (Q6)
    const int r = i / N, c = i % N;
    const uint8_t * p = win + r * W + c;
    im[i] = static_cast<uint8_t>( sixtap( p[0], p[1], p[2], p[3], p[4], p[5], hf[0], hf[1], hf[2], hf[3], hf[4], hf[5] ) );
  }
  __syncthreads();
  for ( int i = lane; i < N * N; i += kLanes ) {            // vertical pass
    const int r = i / N, c = i % N;
    const uint8_t * p = im + r * N + c;
    dst[i] = static_cast<uint8_t>( sixtap( p[0], p[N], p[2 * N], p[3 * N], p[4 * N], p[5 * N], vf[0], vf[1], vf[2], vf[3], vf[4], vf[5] ) );
  }
  __syncthreads();
}

struct alignas( 16 ) InterLds {
  ResidualLds r;
  uint8_t win[24 * 81];      // reference windows: 21x21 / 13x13 (whole-MB) or 24 x 9x9 (SPLITMV)
  uint8_t im[24 * 36];       // first-pass output
  uint8_t pred[384];         // Y 16x16 | U 8x8 | V 8x8, each row-major
  int16_t unit[24][4];       // SPLITMV: per 4x4 unit {sx-2, sy-2, mx, my}
};

// grid.x = macroblock (XCD-aware order), grid.y = frame in batch
__global__ __launch_bounds__( kLanes ) void k_recon_inter( const aa_frame_list list, const unsigned max_mbs )
{
  __shared__ InterLds L;
  const aa_dev_frame & f = *list.f[blockIdx.y];
  const unsigned total = static_cast<unsigned>( f.mbw ) * f.mbh;
  // workgroup b lands on XCD b % 8 (each XCD has its own L2): give each XCD a contiguous run of macroblocks so that
  // horizontally adjacent MBs, which share reference-window cache lines and output lines, hit the same L2.
  const unsigned chunk = ( max_mbs + 7u ) >> 3;
  const unsigned mi = ( blockIdx.x & 7u ) * chunk + ( blockIdx.x >> 3 );
  if ( mi >= total ) return;
  const aa_mb_info & mb = f.mbs[mi];
  if ( !( mb.flags & AA_MB_INTER ) ) return;
  const int lane = threadIdx.x;
  const int col = mi % f.mbw, row = mi / f.mbw;
  const int pw = f.mbw * 16, ph = f.mbh * 16, cw = pw >> 1, ch = ph >> 1;
  const bool has_res = mb.flags & AA_MB_HAS_NONZERO;
  if ( has_res ) compute_residual( mb, f, L.r, lane );

  const uint8_t * const * ref = f.ref[mb.ref_frame & 3];
  if ( mb.y_mode != SPLITMV ) {
    const int mvx = mb.u.mv[0][0], mvy = mb.u.mv[0][1];
    const int cmx = chroma_mv( 4 * mvx ), cmy = chroma_mv( 4 * mvy );
    mc_block<16>( ref[0], pw, ph, col * 16, row * 16, mvx, mvy, L.win, L.im, L.pred, lane );
    mc_block<8>( ref[1], cw, ch, col * 8, row * 8, cmx, cmy, L.win, L.im, L.pred + 256, lane );
    mc_block<8>( ref[2], cw, ch, col * 8, row * 8, cmx, cmy, L.win, L.im, L.pred + 320, lane );
  } else {
    // 16 luma + 4+4 chroma 4x4 units, each with its own vector; all units go through both filter passes
    // (fraction 0 = identity taps, bit-identical to a copy).
    if ( lane < 24 ) {
      int mvx, mvy, x0, y0;
      if ( lane < 16 ) {
        mvx = mb.u.mv[lane][0]; mvy = mb.u.mv[lane][1];
        x0 = col * 16 + ( lane & 3 ) * 4; y0 = row * 16 + ( lane >> 2 ) * 4;
      } else {
        const int b = ( lane - 16 ) & 3, i0 = ( b >> 1 ) * 8 + ( b & 1 ) * 2;
        mvx = chroma_mv( mb.u.mv[i0][0] + mb.u.mv[i0 + 1][0] + mb.u.mv[i0 + 4][0] + mb.u.mv[i0 + 5][0] );
        mvy = chroma_mv( mb.u.mv[i0][1] + mb.u.mv[i0 + 1][1] + mb.u.mv[i0 + 4][1] + mb.u.mv[i0 + 5][1] );
        x0 = col * 8 + ( b & 1 ) * 4; y0 = row * 8 + ( b >> 1 ) * 4;
      }
      L.unit[lane][0] = static_cast<int16_t>( x0 + ( mvx >> 3 ) - 2 ); L.unit[lane][1] = static_cast<int16_t>( y0 + ( mvy >> 3 ) - 2 );
      L.unit[lane][2] = static_cast<int16_t>( mvx & 7 ); L.unit[lane][3] = static_cast<int16_t>( mvy & 7 );
    }
    __syncthreads();
    for ( int i = lane; i < 24 * 81; i += kLanes ) {
      const int u = i / 81, e = i % 81, r = e / 9, c = e % 9;
      const uint8_t * plane = u < 16 ? ref[0] : ( u < 20 ? ref[1] : ref[2] );
      const int w = u < 16 ? pw : cw, h = u < 16 ? ph : ch;
      L.win[i] = plane[static_cast<size_t>( clampi( L.unit[u][1] + r, 0, h - 1 ) ) * w + clampi( L.unit[u][0] + c, 0, w - 1 )];
    }
    __syncthreads();
    for ( int i = lane; i < 24 * 36; i += kLanes ) {
      const int u = i / 36, e = i % 36, r = e >> 2, c = e & 3;
      int t[6]; load_taps( L.unit[u][2], t );
      const uint8_t * p = L.win + u * 81 + r * 9 + c;
      L.im[i] = static_cast<uint8_t>( sixtap( p[0], p[1], p[2], p[3], p[4], p[5], t[0], t[1], t[2], t[3], t[4], t[5] ) );
    }
    __syncthreads();
    for ( int i = lane; i < 24 * 16; i += kLanes ) {
      const int u = i >> 4, e = i & 15, r = e >> 2, c = e & 3;
      int t[6]; load_taps( L.unit[u][3], t );
      const uint8_t * p = L.im + u * 36 + r * 4 + c;
      const int v = sixtap( p[0], p[4], p[8], p[12], p[16], p[20], t[0], t[1], t[2], t[3], t[4], t[5] );
      // scatter into the MB-shaped prediction buffer
      if ( u < 16 ) L.pred[( ( u >> 2 ) * 4 + r ) * 16 + ( u & 3 ) * 4 + c] = static_cast<uint8_t>( v );
      else { const int b = ( u - 16 ) & 3; L.pred[256 + ( u >= 20 ? 64 : 0 ) + ( ( b >> 1 ) * 4 + r ) * 8 + ( b & 1 ) * 4 + c] = static_cast<uint8_t>( v ); }
    }
    __syncthreads();
  }

  // prediction + residual -> raster, 4 pixels (one dword) per lane
  {
    const int r = lane >> 2, c4 = ( lane & 3 ) * 4;
    const int blk = ( r >> 2 ) * 4 + ( c4 >> 2 );
    uint32_t out = 0;
    for ( int j = 0; j < 4; j++ ) {
      int v = L.pred[r * 16 + c4 + j];
      if ( has_res ) v = clamp255( v + L.r.res[blk][( r & 3 ) * 4 + j] );
      out |= static_cast<uint32_t>( v ) << ( 8 * j );
    }
    *reinterpret_cast<uint32_t *>( f.cur[0] + static_cast<size_t>( row * 16 + r ) * pw + col * 16 + c4 ) = out;
  }
  if ( lane < 32 ) {
    const int pl = lane >> 4, l = lane & 15;
    const int r = l >> 1, c4 = ( l & 1 ) * 4;
    const int blk = 16 + pl * 4 + ( r >> 2 ) * 2 + ( c4 >> 2 );
    uint32_t out = 0;
    for ( int j = 0; j < 4; j++ ) {
      int v = L.pred[256 + pl * 64 + r * 8 + c4 + j];
      if ( has_res ) v = clamp255( v + L.r.res[blk][( r & 3 ) * 4 + j] );
      out |= static_cast<uint32_t>( v ) << ( 8 * j );
    }
    *reinterpret_cast<uint32_t *>( f.cur[1 + pl] + static_cast<size_t>( row * 8 + r ) * cw + col * 8 + c4 ) = out;
  }
}

// ---- global accesses that other workgroups of the SAME launch consume / produced -----------------------------------
// Row-pipelined kernels hand pixels from one workgroup to another inside a launch.  Per-XCD L2s are not coherent and a
// CU's L1 is never refreshed by other CUs' stores, so (MI355X_MICROARCH.md "inter-workgroup visibility", valid form
// "{sc1 stores and sc1 loads on both sides}") shared pixels are written with write-through agent-scope stores, the
// producing wave drains them (s_waitcnt vmcnt(0)) before it publishes its progress word, and consumers read them with
// agent-scope (L1-bypassing) loads after one relaxed poll of that word.  No fences.
template <bool kShared>
__device__ __forceinline__ void store_u32( uint8_t * p, uint32_t v )
{
  if ( kShared ) __hip_atomic_store( reinterpret_cast<uint32_t *>( p ), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
  else *reinterpret_cast<uint32_t *>( p ) = v;
}
template <bool kShared>
__device__ __forceinline__ uint32_t load_u32( const uint8_t * p )
{
  if ( kShared ) return __hip_atomic_load( reinterpret_cast<const uint32_t *>( p ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
  return *reinterpret_cast<const uint32_t *>( p );
}

struct alignas( 16 ) IntraLds {
  ResidualLds r;
  alignas( 16 ) uint8_t y[17][24];    // [row+1][col+4]: row -1 = above (cols -4..19 incl. above-right), col -1 = left
  alignas( 16 ) uint8_t c[2][9][12];  // chroma: [plane][row+1][col+4]
};

// One intra macroblock, whole wave: neighbours -> LDS, predict (+ residual), write.  Macroblock::reconstruct_intra
// (macroblock.cc:523-551) with VP8Raster::Block<N>::predictors (prediction.cc:99-167).
// kShared: neighbours may have been produced by another workgroup of this launch (row-pipelined schedule).
template <bool kShared>
__device__ void intra_macroblock( const aa_dev_frame & f, const aa_mb_info & mb, const int col, const int row, IntraLds & L, const int lane )
{
  const int pw = f.mbw * 16, cw = pw >> 1;
  const bool has_res = mb.flags & AA_MB_HAS_NONZERO;
  if ( has_res ) compute_residual( mb, f, L.r, lane );

  const int x0 = col * 16, y0 = row * 16;
  const uint8_t * Y = f.cur[0];
  // above row as 6 dwords: x0-4 (corner in byte 3), x0..x0+15, x0+16 (above-right); left column: byte 3 of the dword at x0-4
  if ( lane < 6 ) {
    uint32_t v;
    if ( y0 == 0 ) v = 0x7F7F7F7Fu;
    else if ( lane == 0 ) v = x0 > 0 ? load_u32<kShared>( Y + static_cast<size_t>( y0 - 1 ) * pw + x0 - 4 ) : 0x81818181u;
    else if ( lane <= 4 ) v = load_u32<kShared>( Y + static_cast<size_t>( y0 - 1 ) * pw + x0 + ( lane - 1 ) * 4 );
    else if ( x0 + 16 >= pw ) v = 0x01010101u * ( load_u32<kShared>( Y + static_cast<size_t>( y0 - 1 ) * pw + pw - 4 ) >> 24 );   // replicate: prediction.cc:144-151
    else v = load_u32<kShared>( Y + static_cast<size_t>( y0 - 1 ) * pw + x0 + 16 );
    *reinterpret_cast<uint32_t *>( &L.y[0][lane * 4] ) = v;
  } else if ( lane >= 32 && lane < 48 ) {
    const int r = lane - 32;
    L.y[r + 1][3] = x0 > 0 ? static_cast<uint8_t>( load_u32<kShared>( Y + static_cast<size_t>( y0 + r ) * pw + x0 - 4 ) >> 24 ) : 129;
  }
  {
    const int cx0 = col * 8, cy0 = row * 8;
    const int pl = lane >> 5, l = lane & 31;
    const uint8_t * C = f.cur[1 + pl];
    if ( l < 3 ) {
      uint32_t v;
      if ( cy0 == 0 ) v = 0x7F7F7F7Fu;
      else if ( l == 0 ) v = cx0 > 0 ? load_u32<kShared>( C + static_cast<size_t>( cy0 - 1 ) * cw + cx0 - 4 ) : 0x81818181u;
      else v = load_u32<kShared>( C + static_cast<size_t>( cy0 - 1 ) * cw + cx0 + ( l - 1 ) * 4 );
      *reinterpret_cast<uint32_t *>( &L.c[pl][0][l * 4] ) = v;
    } else if ( l >= 16 && l < 24 ) {
      const int r = l - 16;
      L.c[pl][r + 1][3] = cx0 > 0 ? static_cast<uint8_t>( load_u32<kShared>( C + static_cast<size_t>( cy0 + r ) * cw + cx0 - 4 ) >> 24 ) : 129;
    }
  }
  __syncthreads();

  // ---- chroma: U then V, 8x8 (prediction.cc:435-450) ----
  if ( lane < 32 ) {
    const int pl = lane >> 4, l = lane & 15;
    const int r = l >> 1, c4 = ( l & 1 ) * 4;
    int sa = 0, sl = 0;
    for ( int i = 0; i < 8; i++ ) { sa += L.c[pl][0][i + 4]; sl += L.c[pl][i + 1][3]; }
    const int dc = bigpred_dc( sa, sl, row > 0, col > 0, 3 );
    const int corner = L.c[pl][0][3], left = L.c[pl][r + 1][3];
    const int blk = 16 + pl * 4 + ( r >> 2 ) * 2 + ( c4 >> 2 );
    uint32_t out = 0;
    for ( int j = 0; j < 4; j++ ) {
      int v = bigpred_pixel( mb.uv_mode, L.c[pl][0][c4 + j + 4], left, corner, dc );
      if ( has_res ) v = clamp255( v + L.r.res[blk][( r & 3 ) * 4 + j] );
      out |= static_cast<uint32_t>( v ) << ( 8 * j );
    }
    store_u32<kShared>( f.cur[1 + pl] + static_cast<size_t>( row * 8 + r ) * cw + col * 8 + c4, out );
  }

  // ---- luma ----
  if ( mb.y_mode != B_PRED ) {
    const int r = lane >> 2, c4 = ( lane & 3 ) * 4;
    int sa = 0, sl = 0;
    for ( int i = 0; i < 16; i++ ) { sa += L.y[0][i + 4]; sl += L.y[i + 1][3]; }
    const int dc = bigpred_dc( sa, sl, row > 0, col > 0, 4 );
    const int corner = L.y[0][3], left = L.y[r + 1][3];
    const int blk = ( r >> 2 ) * 4 + ( c4 >> 2 );
    uint32_t out = 0;
    for ( int j = 0; j < 4; j++ ) {
      int v = bigpred_pixel( mb.y_mode, L.y[0][c4 + j + 4], left, corner, dc );
      if ( has_res ) v = clamp255( v + L.r.res[blk][( r & 3 ) * 4 + j] );
      out |= static_cast<uint32_t>( v ) << ( 8 * j );
    }
    store_u32<kShared>( f.cur[0] + static_cast<size_t>( y0 + r ) * pw + x0 + c4, out );
    __syncthreads();      // LDS is reused by the next macroblock of a row-pipelined workgroup
    return;
  }
  // B_PRED: 16 sub-blocks in raster order, each predicted from already reconstructed pixels then + residual
  // (macroblock.cc:541-544).  Lanes 0..15 own one pixel of the current sub-block.
  for ( int b = 0; b < 16; b++ ) {
    const int bx = b & 3, by = b >> 2;
    int v = 0;
    if ( lane < 16 ) {
      uint8_t E[13];
      const int ar = by * 4, ac = bx * 4 + 3;     // LDS index of (row -1, col -1) of this sub-block
      for ( int i = 0; i < 4; i++ ) E[i] = L.y[ar + 4 - i][ac];
      E[4] = L.y[ar][ac];
      for ( int i = 0; i < 4; i++ ) E[5 + i] = L.y[ar][ac + 1 + i];
      for ( int i = 0; i < 4; i++ ) E[9 + i] = ( bx == 3 ) ? L.y[0][20 + i] : L.y[ar][ac + 5 + i];   // prediction.cc:140-164
      const int c = lane & 3, r = lane >> 2;
      v = bpred_pixel( mb.u.b_mode[b], E, c, r );
      if ( has_res ) v = clamp255( v + L.r.res[b][r * 4 + c] );
    }
    __syncthreads();
    if ( lane < 16 ) L.y[by * 4 + ( lane >> 2 ) + 1][bx * 4 + ( lane & 3 ) + 4] = static_cast<uint8_t>( v );
    __syncthreads();
  }
  {
    const int r = lane >> 2, c4 = ( lane & 3 ) * 4;
    uint32_t out = 0;
    for ( int j = 0; j < 4; j++ ) out |= static_cast<uint32_t>( L.y[r + 1][c4 + j + 4] ) << ( 8 * j );
    store_u32<kShared>( f.cur[0] + static_cast<size_t>( y0 + r ) * pw + x0 + c4, out );
  }
  __syncthreads();
}

// grid.x = position on the diagonal (row = row_lo + blockIdx.x, col = d - 2*row), grid.y = frame in batch
__global__ __launch_bounds__( kLanes ) void k_recon_intra( const aa_frame_list list, const int diagonal, const int row_lo )
{
  __shared__ IntraLds L;
  const aa_dev_frame & f = *list.f[blockIdx.y];
  if ( !f.has_intra ) return;
  const int row = row_lo + blockIdx.x, col = diagonal - 2 * row;
  if ( row >= f.mbh || col < 0 || col >= f.mbw ) return;
  const aa_mb_info & mb = f.mbs[row * f.mbw + col];
  if ( mb.flags & AA_MB_INTER ) return;
  intra_macroblock<false>( f, mb, col, row, L, threadIdx.x );
}

// ---- in-launch ordering for the row-pipelined kernels ------------------------------------------------------------
// One workgroup (one wave) owns one macroblock row of one frame and walks it left to right; row r may work on column c
// once row r-1 has finished column min(c+1, mbw-1) (left/above/above-right dependencies of intra prediction and of the
// loop filter: the same 2:1 wavefront as the per-diagonal launches, without 254 kernel boundaries).
// Deadlock freedom does not rely on residency or dispatch order: rows are handed out by an atomic TICKET in dependency
// order, so the row a workgroup waits for is always held by a workgroup that is already running.  Every spin is
// bounded; on expiry the kernel records an error code and carries on (the host reports AA_ERR_HIP, never a hang).
__device__ __forceinline__ int take_ticket( aa_sync_ws * ws, int * slot, const int lane )
{
  if ( lane == 0 ) *slot = atomicAdd( &ws->ticket, 1 );
  __syncthreads();
  return *slot;
}
__device__ __forceinline__ void publish_progress( int * progress, const int value, const int lane )
{
  asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );      // every store of this wave has been written through
  if ( lane == 0 ) __hip_atomic_store( progress, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
}
__device__ __forceinline__ void wait_progress( const int * progress, const int need, aa_sync_ws * ws )
{
  int spins = 0;
  while ( __hip_atomic_load( progress, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) < need ) {
    __builtin_amdgcn_s_sleep( 2 );
    ++spins;
    // watchdog: sticky error word; once any wait has expired every other wait gives up within 1024 polls
    if ( ( spins & 1023 ) == 0 && __hip_atomic_load( &ws->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) ) break;
    if ( spins > ( 1 << 21 ) ) { __hip_atomic_store( &ws->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ); break; }
  }
}

// grid.x = n_frames * mbh_max workgroups; ticket t -> (frame t / mbh_max, row t % mbh_max)
__global__ __launch_bounds__( kLanes ) void k_recon_intra_rows( const aa_frame_list list, const int n_frames, const int mbh_max, aa_sync_ws * ws )
{
  __shared__ IntraLds L;
  __shared__ int s_ticket;
  const int lane = threadIdx.x;
  const int t = take_ticket( ws, &s_ticket, lane );
  const int fi = t / mbh_max, row = t % mbh_max;
  if ( fi >= n_frames ) return;
  const aa_dev_frame & f = *list.f[fi];
  if ( row >= f.mbh ) return;
  int * progress = ws->progress + fi * mbh_max;
  const int mbw = f.mbw;
  if ( f.has_intra ) {
    const int words = ( mbw + 63 ) >> 6;
    const unsigned long long * mask = f.intra_rows + static_cast<size_t>( row ) * words;
    for ( int w = 0; w < words; w++ ) {
      unsigned long long m = mask[w];
      while ( m ) {
        const int col = w * 64 + __ffsll( static_cast<long long>( m ) ) - 1;
        m &= m - 1;
        // everything left of `col` in this row is final; then wait for the row above
        publish_progress( &progress[row], col, lane );
        if ( row > 0 ) wait_progress( &progress[row - 1], min( col + 2, mbw ), ws );
        intra_macroblock<true>( f, f.mbs[row * mbw + col], col, row, L, lane );
      }
    }
  }
  publish_progress( &progress[row], mbw, lane );
}

struct alignas( 16 ) LfLds {
  alignas( 16 ) uint8_t y[20][20];      // rows -4..15, cols -4..15
  alignas( 16 ) uint8_t c[2][12][12];   // rows -4..7, cols -4..7
};

// One edge position handled by one lane: p = pointer to the first q-side pixel, s = step across the edge.
__device__ __forceinline__ void lf_edge( uint8_t * p, const int s, const bool mb_edge, const LfParams & P )
{
  int p3 = p[-4 * s], p2 = p[-3 * s], p1 = p[-2 * s], p0 = p[-s], q0 = p[0], q1 = p[s], q2 = p[2 * s], q3 = p[3 * s];
  const bool mask = lf_mask( P.interior_limit, mb_edge ? P.mb_limit : P.sb_limit, p3, p2, p1, p0, q0, q1, q2, q3 );
  const bool hev = lf_hev( P.hev_threshold, p1, p0, q0, q1 );
  if ( mb_edge ) {
    lf_macroblock( mask, hev, p2, p1, p0, q0, q1, q2 );
    p[-3 * s] = static_cast<uint8_t>( p2 ); p[2 * s] = static_cast<uint8_t>( q2 );
  } else {
    lf_subblock( mask, hev, p1, p0, q0, q1 );
  }
  p[-2 * s] = static_cast<uint8_t>( p1 ); p[-s] = static_cast<uint8_t>( p0 ); p[0] = static_cast<uint8_t>( q0 ); p[s] = static_cast<uint8_t>( q1 );
}

// The eight dependent edge passes of NormalLoopFilter::filter (loopfilter.cc:133-154) on the LDS copy of one MB:
// left MB edge, inner vertical edges, top MB edge, inner horizontal edges.  Lanes 0..15 luma line, 16..23 U, 24..31 V.
__device__ void lf_passes( LfLds & L, const bool have_left, const bool have_top, const bool inner, const LfParams & P, const int lane )
{
  const bool is_y = lane < 16, is_c = lane >= 16 && lane < 32;
  const int cl = ( lane - 16 ) & 7, cp = ( lane - 16 ) >> 3;
  if ( have_left ) {
    if ( is_y ) lf_edge( &L.y[4 + lane][4], 1, true, P );
    else if ( is_c ) lf_edge( &L.c[cp][4 + cl][4], 1, true, P );
  }
  __syncthreads();
  if ( inner ) {
    if ( is_y ) lf_edge( &L.y[4 + lane][8], 1, false, P );
    else if ( is_c ) lf_edge( &L.c[cp][4 + cl][8], 1, false, P );
    __syncthreads();
    if ( is_y ) lf_edge( &L.y[4 + lane][12], 1, false, P );
    __syncthreads();
    if ( is_y ) lf_edge( &L.y[4 + lane][16], 1, false, P );
    __syncthreads();
  }
  if ( have_top ) {
    if ( is_y ) lf_edge( &L.y[4][4 + lane], 20, true, P );
    else if ( is_c ) lf_edge( &L.c[cp][4][4 + cl], 12, true, P );
  }
  __syncthreads();
  if ( inner ) {
    if ( is_y ) lf_edge( &L.y[8][4 + lane], 20, false, P );
    else if ( is_c ) lf_edge( &L.c[cp][8][4 + cl], 12, false, P );
    __syncthreads();
    if ( is_y ) lf_edge( &L.y[12][4 + lane], 20, false, P );
    __syncthreads();
    if ( is_y ) lf_edge( &L.y[16][4 + lane], 20, false, P );
    __syncthreads();
  }
}

// grid as k_recon_intra.  All MBs with col + 2*row == d are independent: their read/write footprints
// ([x0-4,x0+15] x [y0-4,y0+15]) are disjoint and everything they read was finished by diagonals < d.
__global__ __launch_bounds__( kLanes ) void k_loopfilter( const aa_frame_list list, const int diagonal, const int row_lo )
{
  __shared__ LfLds L;
  const aa_dev_frame & f = *list.f[blockIdx.y];
  if ( !f.loop_filter_level ) return;
  const int row = row_lo + blockIdx.x, col = diagonal - 2 * row;
  if ( row >= f.mbh || col < 0 || col >= f.mbw ) return;
  const aa_mb_info & mb = f.mbs[row * f.mbw + col];
  const int level = mb.lf_level;
  if ( level == 0 ) return;
  const int lane = threadIdx.x;
  const int pw = f.mbw * 16, cw = pw >> 1;
  const int x0 = col * 16, y0 = row * 16, cx0 = col * 8, cy0 = row * 8;
  const LfParams P = lf_params( level, f.sharpness, f.key_frame );

  // ---- stage: 20 rows x 5 dwords (Y), 2 x 12 rows x 3 dwords (U,V) ----
  uint8_t * Y = f.cur[0];
  for ( int i = lane; i < 100; i += kLanes ) {
    const int r = i / 5, d = i % 5;
    const int gy = y0 - 4 + r, gx = x0 - 4 + d * 4;
    uint32_t v = 0;
    if ( gy >= 0 && gx >= 0 ) v = *reinterpret_cast<const uint32_t *>( Y + static_cast<size_t>( gy ) * pw + gx );
    *reinterpret_cast<uint32_t *>( &L.y[r][d * 4] ) = v;
  }
  for ( int i = lane; i < 72; i += kLanes ) {
    const int pl = i / 36, e = i % 36, r = e / 3, d = e % 3;
    const int gy = cy0 - 4 + r, gx = cx0 - 4 + d * 4;
    uint32_t v = 0;
    if ( gy >= 0 && gx >= 0 ) v = *reinterpret_cast<const uint32_t *>( f.cur[1 + pl] + static_cast<size_t>( gy ) * cw + gx );
    *reinterpret_cast<uint32_t *>( &L.c[pl][r][d * 4] ) = v;
  }
  __syncthreads();

  lf_passes( L, col > 0, row > 0, !( mb.flags & AA_MB_LF_SKIP_INNER ), P, lane );

  // ---- write back what this MB may have modified: rows/cols -3..15 minus the untouched corner.
  // Dword stores over [-4,15] are safe: nothing else touches that footprint during this launch.
  for ( int i = lane; i < 100; i += kLanes ) {
    const int r = i / 5, d = i % 5;
    const int gy = y0 - 4 + r, gx = x0 - 4 + d * 4;
    if ( r == 0 || gy < 0 || gx < 0 ) continue;            // row -4 is never modified
    if ( r < 4 && d == 0 ) continue;                        // corner block
    *reinterpret_cast<uint32_t *>( Y + static_cast<size_t>( gy ) * pw + gx ) = *reinterpret_cast<const uint32_t *>( &L.y[r][d * 4] );
  }
  for ( int i = lane; i < 72; i += kLanes ) {
    const int pl = i / 36, e = i % 36, r = e / 3, d = e % 3;
    const int gy = cy0 - 4 + r, gx = cx0 - 4 + d * 4;
    if ( r == 0 || gy < 0 || gx < 0 ) continue;
    if ( r < 4 && d == 0 ) continue;
    *reinterpret_cast<uint32_t *>( f.cur[1 + pl] + static_cast<size_t>( gy ) * cw + gx ) = *reinterpret_cast<const uint32_t *>( &L.c[pl][r][d * 4] );
  }
}

// Row-pipelined loop filter: ticket -> (frame, MB row); the wave walks its row left to right.
//   * per-MB loop-filter level/flags of the whole row are preloaded into LDS once;
//   * the macroblock's own 16 rows were produced by earlier launches -> plain loads, PREFETCHED into registers while
//     the previous macroblock is being filtered;
//   * the 4 rows above belong to the previous MB row, filtered by another workgroup of this launch -> agent-scope
//     loads after progress[row-1] >= min(col+2, mbw);
//   * the 4 columns to the left are the wave's own previous macroblock -> carried over in LDS, never re-read;
//   * every store is write-through; progress[row] = col+1 is published one step late, after the first poll for the
//     next macroblock has returned (its s_waitcnt also drains the stores), so the drain overlaps the prefetch + poll.
constexpr int kMaxMbw = 1024;     // 16383 px / 16

__global__ __launch_bounds__( kLanes ) void k_loopfilter_rows( const aa_frame_list list, const int n_frames, const int mbh_max, aa_sync_ws * ws )
{
  __shared__ LfLds L;
  __shared__ uint16_t s_info[kMaxMbw];     // lf_level | flags << 8
  __shared__ int s_ticket;
  const int lane = threadIdx.x;
  const int t = take_ticket( ws, &s_ticket, lane );
  const int fi = t / mbh_max, row = t % mbh_max;
  if ( fi >= n_frames ) return;
  const aa_dev_frame & f = *list.f[fi];
  if ( row >= f.mbh || !f.loop_filter_level ) return;
  int * progress = ws->progress + fi * mbh_max;
  const int mbw = f.mbw, pw = mbw * 16, cw = pw >> 1;
  const int y0 = row * 16, cy0 = row * 8;
  uint8_t * Y = f.cur[0];
  for ( int c = lane; c < mbw; c += kLanes ) {
    const aa_mb_info & m = f.mbs[row * mbw + c];
    s_info[c] = static_cast<uint16_t>( m.lf_level | ( m.flags << 8 ) );
  }
  __syncthreads();

  // lane roles for the bulk transfers: luma 16 rows x 4 dwords = 64 lanes; chroma 2 planes x 8 rows x 2 dwords = 32 lanes
  const int yr = 4 + ( lane >> 2 ), yd = 1 + ( lane & 3 );
  const int cpl = ( lane >> 4 ) & 1, cr = 4 + ( ( lane >> 1 ) & 7 ), cd = 1 + ( lane & 1 );
  const uint8_t * yrow = Y + static_cast<size_t>( y0 - 4 + yr ) * pw - 4 + yd * 4;
  const uint8_t * crow = f.cur[1 + cpl] + static_cast<size_t>( cy0 - 4 + cr ) * cw - 4 + cd * 4;

  bool carried = false;            // LDS columns -4..-1 hold the previous macroblock's filtered right edge
  bool prefetched = false;
  uint32_t pre_y = 0, pre_c = 0;
  int pending = -1;                // progress value not yet published (its stores may still be in flight)
  for ( int col = 0; col < mbw; col++ ) {
    const int info = s_info[col];
    const int level = info & 0xFF;
    if ( level == 0 ) {
      carried = false; prefetched = false;
      publish_progress( &progress[row], col + 1, lane ); pending = -1;
      continue;
    }
    const int x0 = col * 16, cx0 = col * 8;
    if ( !prefetched ) {
      pre_y = *reinterpret_cast<const uint32_t *>( yrow + x0 );
      if ( lane < 32 ) pre_c = *reinterpret_cast<const uint32_t *>( crow + cx0 );
    }
    if ( !carried && col > 0 ) {   // left neighbour columns straight from memory (previous MB was not filtered)
      if ( lane < 16 ) *reinterpret_cast<uint32_t *>( &L.y[4 + lane][0] ) = *reinterpret_cast<const uint32_t *>( Y + static_cast<size_t>( y0 + lane ) * pw + x0 - 4 );
      else if ( lane < 32 ) { const int l = lane - 16; *reinterpret_cast<uint32_t *>( &L.c[l >> 3][4 + ( l & 7 )][0] ) = *reinterpret_cast<const uint32_t *>( f.cur[1 + ( l >> 3 )] + static_cast<size_t>( cy0 + ( l & 7 ) ) * cw + cx0 - 4 ); }
    }
    if ( row > 0 ) {
      const int need = min( col + 2, mbw );
      int seen = __hip_atomic_load( &progress[row - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
      if ( pending >= 0 ) { publish_progress( &progress[row], pending, lane ); pending = -1; }
      if ( seen < need ) wait_progress( &progress[row - 1], need, ws );
      if ( lane < 16 ) {             // 4 rows x 4 dwords above the luma block
        const int r = lane >> 2, d = 1 + ( lane & 3 );
        *reinterpret_cast<uint32_t *>( &L.y[r][d * 4] ) = load_u32<true>( Y + static_cast<size_t>( y0 - 4 + r ) * pw + x0 - 4 + d * 4 );
      } else if ( lane < 32 ) {      // 2 planes x 4 rows x 2 dwords above the chroma blocks
        const int l = lane - 16, pl = l >> 3, r = ( l >> 1 ) & 3, d = 1 + ( l & 1 );
        *reinterpret_cast<uint32_t *>( &L.c[pl][r][d * 4] ) = load_u32<true>( f.cur[1 + pl] + static_cast<size_t>( cy0 - 4 + r ) * cw + cx0 - 4 + d * 4 );
      }
    } else if ( pending >= 0 ) { publish_progress( &progress[row], pending, lane ); pending = -1; }
    *reinterpret_cast<uint32_t *>( &L.y[yr][yd * 4] ) = pre_y;
    if ( lane < 32 ) *reinterpret_cast<uint32_t *>( &L.c[cpl][cr][cd * 4] ) = pre_c;
    __syncthreads();

    lf_passes( L, col > 0, row > 0, !( ( info >> 8 ) & AA_MB_LF_SKIP_INNER ), lf_params( level, f.sharpness, f.key_frame ), lane );

    // write-through stores: rows -3..-1 x cols 0..15 (previous MB row), rows 0..15 x cols -4..15
    for ( int i = lane; i < 100; i += kLanes ) {
      const int r = i / 5, d = i % 5;
      if ( r == 0 || ( r < 4 && ( d == 0 || row == 0 ) ) || ( d == 0 && col == 0 ) ) continue;
      store_u32<true>( Y + static_cast<size_t>( y0 - 4 + r ) * pw + x0 - 4 + d * 4, *reinterpret_cast<const uint32_t *>( &L.y[r][d * 4] ) );
    }
    for ( int i = lane; i < 72; i += kLanes ) {
      const int pl = i / 36, e = i % 36, r = e / 3, d = e % 3;
      if ( r == 0 || ( r < 4 && ( d == 0 || row == 0 ) ) || ( d == 0 && col == 0 ) ) continue;
      store_u32<true>( f.cur[1 + pl] + static_cast<size_t>( cy0 - 4 + r ) * cw + cx0 - 4 + d * 4, *reinterpret_cast<const uint32_t *>( &L.c[pl][r][d * 4] ) );
    }
    pending = col + 1;
    // prefetch the next macroblock's own rows while these stores drain
    prefetched = col + 1 < mbw && ( s_info[col + 1] & 0xFF ) != 0;
    if ( prefetched ) {
      pre_y = *reinterpret_cast<const uint32_t *>( yrow + x0 + 16 );
      if ( lane < 32 ) pre_c = *reinterpret_cast<const uint32_t *>( crow + cx0 + 8 );
    }
    // carry the filtered right edge (cols 12..15 / 4..7) over as the next macroblock's left neighbour columns
    if ( lane < 16 ) *reinterpret_cast<uint32_t *>( &L.y[4 + lane][0] ) = *reinterpret_cast<const uint32_t *>( &L.y[4 + lane][16] );
    else if ( lane < 32 ) { const int l = lane - 16; *reinterpret_cast<uint32_t *>( &L.c[l >> 3][4 + ( l & 7 )][0] ) = *reinterpret_cast<const uint32_t *>( &L.c[l >> 3][4 + ( l & 7 )][8] ); }
    carried = true;
    __syncthreads();
  }
  if ( pending >= 0 ) publish_progress( &progress[row], pending, lane );
}

} // namespace

int launch_recon_inter( const aa_frame_list & list, int n, unsigned max_mbs, void * stream )
{
  const unsigned blocks = ( ( max_mbs + 7u ) >> 3 ) * 8u;
  hipLaunchKernelGGL( k_recon_inter, dim3( blocks, n ), dim3( kLanes ), 0, static_cast<hipStream_t>( stream ), list, max_mbs );
  return static_cast<int>( hipGetLastError() );
}
int launch_recon_intra_diagonal( const aa_frame_list & list, int n, int diagonal, int row_lo, int rows, void * stream )
{
  hipLaunchKernelGGL( k_recon_intra, dim3( rows, n ), dim3( kLanes ), 0, static_cast<hipStream_t>( stream ), list, diagonal, row_lo );
  return static_cast<int>( hipGetLastError() );
}
int launch_loopfilter_diagonal( const aa_frame_list & list, int n, int diagonal, int row_lo, int rows, void * stream )
{
  hipLaunchKernelGGL( k_loopfilter, dim3( rows, n ), dim3( kLanes ), 0, static_cast<hipStream_t>( stream ), list, diagonal, row_lo );
  return static_cast<int>( hipGetLastError() );
}
int launch_recon_intra_rows( const aa_frame_list & list, int n, int mbh_max, aa_sync_ws * ws, void * stream )
{
  hipLaunchKernelGGL( k_recon_intra_rows, dim3( n * mbh_max ), dim3( kLanes ), 0, static_cast<hipStream_t>( stream ), list, n, mbh_max, ws );
  return static_cast<int>( hipGetLastError() );
}
int launch_loopfilter_rows( const aa_frame_list & list, int n, int mbh_max, aa_sync_ws * ws, void * stream )
{
  hipLaunchKernelGGL( k_loopfilter_rows, dim3( n * mbh_max ), dim3( kLanes ), 0, static_cast<hipStream_t>( stream ), list, n, mbh_max, ws );
  return static_cast<int>( hipGetLastError() );
}

} // namespace aa
