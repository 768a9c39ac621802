// HIP kernels (gfx950 / CDNA4) for the VP8 reconstruction hot path.  64-lane waves, 16 lanes per macroblock: every unit of
// work here is 16 wide (a 4x4 sub-block, the rows of a 16x16 prediction, 2 x 8 chroma rows, 16 luma IDCTs, the lines of
// a filter edge), so a wave carries four independent macroblocks / frames.
//
//   k_recon_inter4      whole-vector inter MBs, four consecutive MBs per wave: in-register dequant/IDCT, six-tap motion
//                       compensation from LDS-staged reference windows (integer vectors: a copy)     (macroblock.cc:553-601)
//   k_recon_inter       SPLITMV macroblocks, one per wave (launched only when a frame has any)
//   k_recon_intra4      row-pipelined intra prediction, four frames per wave                       (macroblock.cc:523-551)
//   k_loopfilter_rows4  row-pipelined normal loop filter, four frames per wave, packed int16 arithmetic, strip-staged
//                       whole-line I/O                                                              (loopfilter.cc:133-154)
//   k_recon_intra / k_loopfilter   the first schedule (one launch per 2:1 anti-diagonal, one MB per wave), kept as an
//                       independent implementation for A/B runs (ALFALFA_AMD_SCHEDULE=diagonal)
//
// No MFMA: nothing here is a dense contraction.  The path is integer stencil/gather work; what binds it is instruction
// issue and the raster-order dependency wavefront, HBM traffic stays near the algorithmic bytes.  Batches of independent
// frames (streams / GOPs) fill the chip.  The row-pipelined kernels are XCD-affine (see take_ticket).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstddef>

#include <cstdlib>

#include "device_types.h"
#include "vp8_math.hh"
#include "coeff_pack.hh"
#include "recon_inl.hh"

namespace aa {
namespace {
using namespace recon;

constexpr int kLanes = 64;

enum : int { DC_PRED, V_PRED, H_PRED, TM_PRED, B_PRED, NEARESTMV, NEARMV, ZEROMV, NEWMV, SPLITMV };

struct alignas( 16 ) ResidualLds {
  union {
    int16_t cf[25][16];  // dense, dequantised coefficients (block 24 = Y2); dead once the first IDCT pass has run
    int16_t res[24][16]; // residual to add: block b, [row*4+col] -- written by the second pass over the same storage
  };
  int16_t im[24][16];    // first-pass results (int16, Q5)
  uint8_t map[32];       // rank of stored block -> dense block id
};

// Dequantise the MB's stored blocks, run iWHT (if Y2) and the 24 IDCTs; leaves residuals in L.res.
// Macroblock::apply_walsh / DCTCoefficients::{dequantize,iwht,idct_add}.  Called by the whole wave.
__device__ void compute_residual( const aa_mb_info & mb, const aa_dev_frame & f, ResidualLds & L, const int lane )
{
  uint32_t * z = reinterpret_cast<uint32_t *>( &L.cf[0][0] );
  for ( int i = lane; i < 200; i += kLanes ) z[i] = 0;
  const uint32_t mask = mb.nz_mask;
  // storage order: Y2 (bit 24) first, then bit order
  if ( lane < 25 && ( ( mask >> lane ) & 1u ) ) L.map[lane == 24 ? 0 : __popc( mask & ( ( 1u << lane ) - 1u ) ) + static_cast<int>( mask >> 24 )] = static_cast<uint8_t>( lane );
  __syncthreads();
  const uint16_t * q = f.quant[mb.segment_id & 3];
  if ( f.packed ) {
    // packed storage (coeff_pack.hh): 25 mask slots, then the values of the stored blocks in parse order.  Masks and the offsets
    // of the blocks' values go through LDS (L.im is free until the first IDCT pass), then every lane fetches the coefficients of
    // its raster positions one by one
    const int16_t * const w = f.coeffs + ( static_cast<size_t>( mb.reserved ) << 32 | mb.coeff_index );
    uint16_t * const scr = reinterpret_cast<uint16_t *>( &L.im[0][0] );          // [0, 25) masks, [32, 57) offsets
    if ( lane < 25 ) scr[lane] = ( ( mask >> lane ) & 1u ) ? static_cast<uint16_t>( w[lane] ) : 0u;
    __syncthreads();
    if ( lane < 25 ) {
      uint32_t off = pack::kMaskSlots;
      if ( lane != 24 ) { off += __popc( scr[24] ); for ( int b = 0; b < lane; b++ ) off += __popc( scr[b] ); }
      scr[32 + lane] = static_cast<uint16_t>( off );
    }
    __syncthreads();
    for ( int k = lane; k < 25 * 16; k += kLanes ) {
      const int blk = k >> 4, e = k & 15;
      const uint32_t m = scr[blk], zz = pack::zigzag_of( e );
      if ( ( m >> zz ) & 1u ) {
        const int base = blk < 16 ? 0 : ( blk < 24 ? 4 : 2 );          // {y_dc,y_ac,y2_dc,y2_ac,uv_dc,uv_ac}
        L.cf[blk][e] = static_cast<int16_t>( dequant( w[scr[32 + blk] + __popc( m & ( ( 1u << zz ) - 1u ) )], q[base + ( e ? 1 : 0 )] ) );
      }
    }
  } else {
    const int n = __popc( mask ) * 16;
    const int16_t * src = f.coeffs + static_cast<size_t>( mb.coeff_index ) * 16;
    for ( int k = lane; k < n; k += kLanes ) {
      const int blk = L.map[k >> 4], e = k & 15;
      const int base = blk < 16 ? 0 : ( blk < 24 ? 4 : 2 );          // {y_dc,y_ac,y2_dc,y2_ac,uv_dc,uv_ac}
      L.cf[blk][e] = static_cast<int16_t>( dequant( src[k], q[base + ( e ? 1 : 0 )] ) );
    }
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

// ---- six-tap motion compensation on packed bytes ------------------------------------------------------------------
// One lane produces 4 adjacent outputs of a filter pass from 12 consecutive source bytes held in three dwords:
// about 7 VALU instructions per output instead of ~25 for byte-at-a-time code.
// 4 bytes starting at byte offset s (0..11) of the 16-byte string d0 d1 d2 0
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

struct alignas( 16 ) InterLds {     // ~5 KB: 32 single-wave workgroups fit one CU (the kernel is latency bound: occupancy matters)
  ResidualLds r;
  alignas( 16 ) uint8_t pred[384];   // Y 16x16 | U 8x8 | V 8x8, each row-major
  union {
    struct {                         // SPLITMV: 24 units of 4x4 with their own vectors
      uint32_t sw[24][9][3];         // 9-row reference windows: 12 bytes per row (aligned dwords; the window starts at byte sx & 3)
      uint32_t st[24][4][3];         // first-pass output TRANSPOSED: per unit and column, 9 rows in 12 bytes
      int16_t unit[24][6];           // per unit {sx-2, sy-2, mx, my, byte offset of the window in its row, inside the plane}
    };
    struct {                         // whole-MB vector: reference windows as dword rows, first-pass output stored
      uint32_t wy[21][6], wc[2][13][4];   // TRANSPOSED (column-major) so that the vertical pass also reads 12
      uint32_t ty[16][6], tc[2][8][4];    // consecutive bytes per lane
    };
  };
};

// One inter macroblock by one wave.  `bx` = workgroup index within the frame's run of blocks (XCD-aware order).
__device__ __forceinline__ void recon_inter_body( const aa_dev_frame & f, const unsigned bx, const unsigned max_mbs, InterLds & L, const bool split_only )
{
  const unsigned total = static_cast<unsigned>( f.mbw ) * f.mbh;
  // workgroup b lands on XCD b % 8 (each XCD has its own L2): give each XCD a contiguous run of macroblocks so that
  // horizontally adjacent MBs, which share reference-window cache lines and output lines, hit the same L2.
  const unsigned chunk = ( max_mbs + 7u ) >> 3;
  const unsigned mi = ( bx & 7u ) * chunk + ( bx >> 3 );
  if ( mi >= total ) return;
  const aa_mb_info & mb = f.mbs[mi];
  if ( !( mb.flags & AA_MB_INTER ) ) return;
  if ( split_only && mb.y_mode != SPLITMV ) return;       // whole-vector macroblocks are k_recon_inter4's
  const int lane = threadIdx.x;
  const int col = mi % f.mbw, row = mi / f.mbw;
  const int pw = f.mbw * 16, ph = f.mbh * 16, cw = pw >> 1, ch = ph >> 1;
  const bool has_res = mb.flags & AA_MB_HAS_NONZERO;
  const uint8_t * const * ref = f.ref[mb.ref_frame & 3];
  const bool whole = mb.y_mode != SPLITMV;
  // Y 16x16 with the MB vector, U/V 8x8 with the derived chroma vector (macroblock.cc:583-586)
  const int mvx = mb.u.mv[0][0], mvy = mb.u.mv[0][1];
  const int cmx = chroma_mv( 4 * mvx ), cmy = chroma_mv( 4 * mvy );
  const int sxy = col * 16 + ( mvx >> 3 ) - 2, syy = row * 16 + ( mvy >> 3 ) - 2;     // window origins (Q9: arithmetic shift)
  const int sxc = col * 8 + ( cmx >> 3 ) - 2, syc = row * 8 + ( cmy >> 3 ) - 2;
  // aligned-dword staging when the windows lie inside the planes, else byte-wise coordinate clamping (EdgeExtendedRaster)
  const int axy = sxy & ~3, axc = sxc & ~3;
  const bool inside = whole && sxy >= 0 && syy >= 0 && axy + 24 <= pw && syy + 21 <= ph && sxc >= 0 && syc >= 0 && axc + 16 <= cw && syc + 13 <= ch;
  // issue the reference-window loads before the residual work so that their latency overlaps it
  uint32_t wreg[4] = { 0, 0, 0, 0 };
  if ( inside ) {
#pragma unroll
    for ( int k = 0; k < 4; k++ ) {
      const int i = lane + k * kLanes;
      if ( i < 126 ) { const int r = i / 6, d = i % 6; wreg[k] = *reinterpret_cast<const uint32_t *>( ref[0] + static_cast<size_t>( syy + r ) * pw + axy + d * 4 ); }
      else if ( i < 230 ) { const int j = i - 126, pl = j / 52, e = j % 52, r = e >> 2, d = e & 3;
                            wreg[k] = *reinterpret_cast<const uint32_t *>( ref[1 + pl] + static_cast<size_t>( syc + r ) * cw + axc + d * 4 ); }
    }
  }
  if ( has_res ) compute_residual( mb, f, L.r, lane );

  if ( whole ) {
    int oy, oc;
    if ( inside ) {
      oy = sxy & 3; oc = sxc & 3;
      uint32_t * flat = &L.wy[0][0];          // wy (126 dwords) is directly followed by wc (104 dwords)
#pragma unroll
      for ( int k = 0; k < 4; k++ ) { const int i = lane + k * kLanes; if ( i < 230 ) flat[i] = wreg[k]; }
    } else {
      oy = 0; oc = 0;
      uint8_t * by = reinterpret_cast<uint8_t *>( &L.wy[0][0] );
      for ( int i = lane; i < 21 * 24; i += kLanes ) {
        const int r = i / 24, c = i % 24;
        by[i] = ref[0][static_cast<size_t>( clampi( syy + r, 0, ph - 1 ) ) * pw + clampi( sxy + c, 0, pw - 1 )];
      }
      uint8_t * bc = reinterpret_cast<uint8_t *>( &L.wc[0][0][0] );
      for ( int i = lane; i < 2 * 13 * 16; i += kLanes ) {
        const int pl = i / 208, e = i % 208, r = e >> 4, c = e & 15;
        bc[i] = ref[1 + pl][static_cast<size_t>( clampi( syc + r, 0, ch - 1 ) ) * cw + clampi( sxc + c, 0, cw - 1 )];
      }
    }
    uint32_t hy0, hy1, vy0, vy1, hc0, hc1, vc0, vc1;
    pack_taps( mvx & 7, hy0, hy1 ); pack_taps( mvy & 7, vy0, vy1 ); pack_taps( cmx & 7, hc0, hc1 ); pack_taps( cmy & 7, vc0, vc1 );
    __syncthreads();
    // horizontal pass over N+5 rows; result bytes go to the transposed buffers t[column][row]
    for ( int t = lane; t < 84 + 52; t += kLanes ) {
      if ( t < 84 ) {
        const int r = t >> 2, g = t & 3;
        const uint32_t o4 = sixtap_x4( L.wy[r][g], L.wy[r][g + 1], L.wy[r][g + 2], oy, mvx & 7, hy0, hy1 );
        uint8_t * tb = reinterpret_cast<uint8_t *>( &L.ty[g * 4][0] ) + r;
        tb[0] = static_cast<uint8_t>( o4 ); tb[24] = static_cast<uint8_t>( o4 >> 8 ); tb[48] = static_cast<uint8_t>( o4 >> 16 ); tb[72] = static_cast<uint8_t>( o4 >> 24 );
      } else {
        const int j = t - 84, pl = j / 26, e = j % 26, r = e >> 1, g = e & 1;
        const uint32_t o4 = sixtap_x4( L.wc[pl][r][g], L.wc[pl][r][g + 1], L.wc[pl][r][g + 2], oc, cmx & 7, hc0, hc1 );
        uint8_t * tb = reinterpret_cast<uint8_t *>( &L.tc[pl][g * 4][0] ) + r;
        tb[0] = static_cast<uint8_t>( o4 ); tb[16] = static_cast<uint8_t>( o4 >> 8 ); tb[32] = static_cast<uint8_t>( o4 >> 16 ); tb[48] = static_cast<uint8_t>( o4 >> 24 );
      }
    }
    __syncthreads();
    // vertical pass: lane = (column c, row group i) -> rows 4i..4i+3 of column c
    for ( int t = lane; t < 64 + 32; t += kLanes ) {
      if ( t < 64 ) {
        const int c = t >> 2, i = t & 3;
        const uint32_t o4 = sixtap_x4( L.ty[c][i], L.ty[c][i + 1], L.ty[c][i + 2], 0, mvy & 7, vy0, vy1 );
        uint8_t * pb = L.pred + ( i * 4 ) * 16 + c;
        pb[0] = static_cast<uint8_t>( o4 ); pb[16] = static_cast<uint8_t>( o4 >> 8 ); pb[32] = static_cast<uint8_t>( o4 >> 16 ); pb[48] = static_cast<uint8_t>( o4 >> 24 );
      } else {
        const int j = t - 64, pl = j >> 4, c = ( j >> 1 ) & 7, i = j & 1;
        const uint32_t o4 = sixtap_x4( L.tc[pl][c][i], L.tc[pl][c][i + 1], L.tc[pl][c][i + 2], 0, cmy & 7, vc0, vc1 );
        uint8_t * pb = L.pred + 256 + pl * 64 + ( i * 4 ) * 8 + c;
        pb[0] = static_cast<uint8_t>( o4 ); pb[8] = static_cast<uint8_t>( o4 >> 8 ); pb[16] = static_cast<uint8_t>( o4 >> 16 ); pb[24] = static_cast<uint8_t>( o4 >> 24 );
      }
    }
    __syncthreads();
  } else {
    // 16 luma + 4+4 chroma 4x4 units, each with its own vector (macroblock.cc:553-601, prediction.cc:813-971); all units go
    // through both filter passes (fraction 0 = a copy).  Same packed arithmetic as the whole-vector path: a task makes 4
    // outputs from 12 source bytes (sixtap_x4), 9 first-pass tasks and 4 second-pass tasks per unit.
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
      const int sx = x0 + ( mvx >> 3 ) - 2, sy = y0 + ( mvy >> 3 ) - 2;
      const int w = lane < 16 ? pw : cw, h = lane < 16 ? ph : ch;
      const bool in = sx >= 0 && sy >= 0 && ( sx & ~3 ) + 12 <= w && sy + 9 <= h;
      L.unit[lane][0] = static_cast<int16_t>( sx ); L.unit[lane][1] = static_cast<int16_t>( sy );
      L.unit[lane][2] = static_cast<int16_t>( mvx & 7 ); L.unit[lane][3] = static_cast<int16_t>( mvy & 7 );
      L.unit[lane][4] = static_cast<int16_t>( in ? ( sx & 3 ) : 0 ); L.unit[lane][5] = in;
    }
    __syncthreads();
    for ( int i = lane; i < 24 * 27; i += kLanes ) {          // window rows as aligned dwords (coordinate clamping outside the plane)
      const int u = i / 27, e = i % 27, r = e / 3, d = e % 3;
      const uint8_t * plane = u < 16 ? ref[0] : ( u < 20 ? ref[1] : ref[2] );
      const int w = u < 16 ? pw : cw, h = u < 16 ? ph : ch;
      const int sx = L.unit[u][0], sy = L.unit[u][1];
      uint32_t v;
      if ( L.unit[u][5] ) v = *reinterpret_cast<const uint32_t *>( plane + static_cast<size_t>( sy + r ) * w + ( sx & ~3 ) + d * 4 );
      else {
        const uint8_t * line = plane + static_cast<size_t>( clampi( sy + r, 0, h - 1 ) ) * w;
        v = 0;
        for ( int k = 0; k < 4; k++ ) v |= static_cast<uint32_t>( line[clampi( sx + d * 4 + k, 0, w - 1 )] ) << ( 8 * k );
      }
      L.sw[u][r][d] = v;
    }
    __syncthreads();
    for ( int t = lane; t < 24 * 9; t += kLanes ) {           // first pass: unit u, row r -> 4 bytes, stored transposed
      const int u = t / 9, r = t % 9, fx = L.unit[u][2];
      uint32_t t0, t1;
      pack_taps( fx, t0, t1 );
      const uint32_t o4 = sixtap_x4( L.sw[u][r][0], L.sw[u][r][1], L.sw[u][r][2], L.unit[u][4], fx, t0, t1 );
      uint8_t * tb = reinterpret_cast<uint8_t *>( &L.st[u][0][0] ) + r;
      tb[0] = static_cast<uint8_t>( o4 ); tb[12] = static_cast<uint8_t>( o4 >> 8 ); tb[24] = static_cast<uint8_t>( o4 >> 16 ); tb[36] = static_cast<uint8_t>( o4 >> 24 );
    }
    __syncthreads();
    for ( int t = lane; t < 24 * 4; t += kLanes ) {           // second pass: unit u, column c -> rows 0..3
      const int u = t >> 2, c = t & 3, fy = L.unit[u][3];
      uint32_t t0, t1;
      pack_taps( fy, t0, t1 );
      const uint32_t o4 = sixtap_x4( L.st[u][c][0], L.st[u][c][1], L.st[u][c][2], 0, fy, t0, t1 );
      uint8_t * pb;
      int stride;
      if ( u < 16 ) { pb = L.pred + ( ( u >> 2 ) * 4 ) * 16 + ( u & 3 ) * 4 + c; stride = 16; }
      else { const int b = ( u - 16 ) & 3; pb = L.pred + 256 + ( u >= 20 ? 64 : 0 ) + ( ( b >> 1 ) * 4 ) * 8 + ( b & 1 ) * 4 + c; stride = 8; }
      pb[0] = static_cast<uint8_t>( o4 ); pb[stride] = static_cast<uint8_t>( o4 >> 8 ); pb[2 * stride] = static_cast<uint8_t>( o4 >> 16 ); pb[3 * stride] = static_cast<uint8_t>( o4 >> 24 );
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

// grid.x = macroblock (XCD-aware order), grid.y = frame in batch
__global__ __launch_bounds__( kLanes ) void k_recon_inter( const aa_frame_list list, const unsigned max_mbs, const int split_only )
{
  __builtin_amdgcn_s_setprio( 3 );      // reconstruction shares the SIMDs with seconds-long entropy-decode waves: win the issue arbitration

  __shared__ InterLds L;
  recon_inter_body( *list.f[blockIdx.y], blockIdx.x, max_mbs, L, split_only != 0 );
}

// ---- global accesses that other workgroups of the SAME launch consume / produced -----------------------------------
// Row-pipelined kernels hand pixels from one workgroup to another inside a launch.  Per-XCD L2s are not coherent with
// each other and a CU's L1 is never refreshed by other CUs' stores (MI355X_MICROARCH.md "inter-workgroup visibility").
// The row kernels are therefore XCD-AFFINE: every macroblock row of a frame is processed by a workgroup of the SAME XCD
// (per-XCD ticket queues indexed by the hardware XCC_ID, see take_ticket), so the XCD's L2 is the coherence point:
//   producer: ordinary (sc0) stores -- the line STAYS in the XCD's L2 -- drained with s_waitcnt vmcnt(0) before the
//             progress word is stored;
//   consumer: one relaxed poll of the progress word, then sc1 loads (bypass the CU's L1, served by the L2).
// No fences, no write-through to HBM, no fabric round trip per hand-off (measured with sc1 stores: the loop filter moved
// 5x its algorithmic bytes and was transaction bound).
template <bool kShared>
__device__ __forceinline__ void store_u32( uint8_t * p, uint32_t v )
{
  *reinterpret_cast<uint32_t *>( p ) = v;        // kShared: same instruction -- an ordinary store is what keeps the line in the XCD's L2
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
__device__ void intra_macroblock( const aa_dev_frame & f, const aa_mb_info & mb, const int col, const int row, IntraLds & L, const int lane,
                                  const bool residual_ready = false )
{
  const int pw = f.mbw * 16, cw = pw >> 1;
  const bool has_res = mb.flags & AA_MB_HAS_NONZERO;
  if ( has_res && !residual_ready ) compute_residual( mb, f, L.r, lane );

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
// One workgroup (one wave) owns one macroblock row of one unit (a frame, or a group of four frames) and walks it left to
// right; row r may work on column c once row r-1 has finished column min(c+1, mbw-1) (left/above/above-right
// dependencies of intra prediction and of the loop filter: the same 2:1 wavefront as the per-diagonal launches, without
// 254 kernel boundaries).
// Work is handed out by TICKET, one queue per XCD: unit u belongs to XCD u % n_xcd, and a workgroup only ever takes
// tickets of the XCD it really runs on (HW_REG_XCC_ID), in dependency order: ROW-major (row 0 of every unit of the XCD,
// then row 1, ...), so that when fewer workgroups are resident than there are rows the chip sweeps all units top to
// bottom ONCE instead of running the 2:1 wavefront's critical path once per batch of resident units.  So
//   * all rows of a unit run on one XCD -> hand-offs go through that XCD's L2 (see above);
//   * deadlock freedom does not rely on residency or dispatch order: the row a workgroup waits for has a lower ticket
//     of the same queue and is therefore held by a workgroup that is already running;
//   * a workgroup keeps taking tickets until its queue is empty, so every queue is drained as long as ONE workgroup
//     lands on each XCD (the grid has n_xcd x the work of the fullest queue; with the observed round-robin placement
//     block b -> XCD b % 8 every workgroup takes exactly one ticket).
// Every spin is bounded; on expiry the kernel records an error code and carries on (the host reports AA_ERR_HIP).
// Bound of a hand-off wait.  The wait cannot deadlock (see above) but the workgroup waited for may be slow: reconstruction
// runs beside thousands of entropy-decode waves that hold their SIMDs for seconds.  A wait gives up when it has lasted TWO SECONDS
// of the 100 MHz clock AND the wave has polled four million times: a row kernel lasts milliseconds, so a broken hand-off is
// reported in seconds (round 3 bounded the polls alone at 2^26: minutes).  Time alone is not a bound: the first version of this
// round had only the clock and expired once in a 20-step run ("needed 12 saw 11") -- wall time also passes while the whole GPU is
// held up (the host mapping another GiB into the coefficient heap, pinning an arena), polls do not.  The clock is read once per
// 1024 polls.  (Lowering the waiting wave's issue priority was tried with it and dropped: no measured gain.)
// Round 6: SIXTY seconds (and 2^24 polls): ten were run out once in a priming pass of session 13 (1 run in ~60 of the round); the wave
// that gives up now also leaves what it saw of its unit's rows (wait_expired).
// Round 5: TEN seconds and 2^24 polls.  Two seconds were run out once more, in a priming pass (heap being mapped, arenas being pinned,
// sixteen host-lane threads uploading from pageable memory -- since removed): the wait is a safety net against a broken hand-off, and
// a net that tears under a slow but correct run costs a whole job.
constexpr unsigned long long kMaxWaitTicks = 6000000000ull;
constexpr int kMinPolls = 1 << 24;
// The slow path's second look at the row above (every 1024 polls, ~0.5 ms into a wait).  Session 16's dump of an expired wait showed the
// row above COMPLETE (120 columns) while the waiting wave's polls -- one agent-scope load of one address for the whole wave -- had
// returned "38" for sixty seconds ("needed N saw N - 1" every time such a wait has expired, rounds 5 and 6, always in a priming pass):
// a hand-off that was written and never seen, not a workgroup that was not running.  Whatever held the stale value (the dump's own loads,
// a microsecond later, from the same CU, were fresh), a second look that shares nothing with the first ends it: the caches invalidated
// (buffer_inv sc1), the address per lane (a vector-addressed load, not the scalar-base form of the poll), system scope, and a
// read-modify-write that the L2 itself executes.  progress only grows, so the larger value is the truth.  Counted (ws->dump[RESCUES..]):
// how many waits ended this way, and by which read.
#ifndef AA_HANDOFF_PUBLISH_AGENT
#define AA_HANDOFF_PUBLISH_AGENT 0    /* build parameter (A/B runs): 1 = a row's progress word is stored with agent scope (sc1: written through), not workgroup scope */
#endif
#if AA_HANDOFF_PUBLISH_AGENT
#define AA_PUBLISH_SCOPE __HIP_MEMORY_SCOPE_AGENT
#else
#define AA_PUBLISH_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
#ifndef AA_HANDOFF_POLL_FORM
#define AA_HANDOFF_POLL_FORM 0        /* build parameter (A/B runs): 1 = the poll itself is a vector-addressed load */
#endif
#ifndef AA_HANDOFF_LOOK_EVERY
#define AA_HANDOFF_LOOK_EVERY 1024    /* build parameter (A/B runs): polls between two second looks (a power of two) */
#endif
__device__ __forceinline__ int poll_progress( const int * p )
{
#if AA_HANDOFF_POLL_FORM
  int off = 0;
  asm volatile( "" : "+v"( off ) );
  p += off;
#endif
  return __hip_atomic_load( p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
}
__device__ __forceinline__ int reread_progress( aa_sync_ws * ws, const int * p, const int seen, const int need, const int kernel )
{
  int off = 0;
  asm volatile( "" : "+v"( off ) );                       // (opaque: keeps the address in vector registers)
  int * q = const_cast<int *>( p ) + off;
  __builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "agent" );
  const int a = __hip_atomic_load( q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
  const int b = __hip_atomic_fetch_or( q, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
  const int best = max( seen, max( a, b ) );
  if ( seen < need && best >= need ) {
    // (... and the poll's own form once more, behind the reads that saw the value: still short of it = the poll was STALE, not early)
    const int c = poll_progress( p );
    if ( ( threadIdx.x & 63 ) == __builtin_ctzll( __ballot( 1 ) ) ) {       // one count per wave
      atomicAdd( &ws->dump[AA_SYNC_WS_RESCUES], 1 );
      if ( a >= need ) atomicAdd( &ws->dump[AA_SYNC_WS_RESCUES + 1], 1 );
      if ( b >= need ) atomicAdd( &ws->dump[AA_SYNC_WS_RESCUES + 2], 1 );
      if ( c < need ) atomicAdd( &ws->dump[AA_SYNC_WS_RESCUES + 3], 1 );
      atomicAdd( &ws->dump[AA_SYNC_WS_RESCUES + 3 + kernel], 1 );          // (1: k_recon_intra4, 2: k_loopfilter_rows4)
    }
  }
  return best;
}
// The first wave whose wait expires says where, and leaves what it sees of its unit's rows at that moment (the row that is not
// moving while the row above it is far ahead, or done, is the one to look at)
__device__ __noinline__ void wait_expired( aa_sync_ws * ws, const int code, const int group, const int row, const int need, const int seen, const int * progress, const int mbh,
                                           const int spins, const unsigned long long waited )
{
  int won = 0;
  if ( threadIdx.x == 0 ) won = atomicCAS( &ws->error, 0, code ) == 0;
  if ( !__shfl( won, 0 ) ) return;
  progress = reinterpret_cast<const int *>( static_cast<uintptr_t>( __shfl( static_cast<unsigned long long>( reinterpret_cast<uintptr_t>( progress ) ), 0 ) ) );   // (k_recon_intra4: a row array per frame of the wave -- lane 0's)
  for ( int i = threadIdx.x; i < 128; i += 64 ) ws->dump[i] = i < mbh ? __hip_atomic_load( &progress[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) : -1;
  if ( threadIdx.x == 0 ) {
    ws->where[0] = group; ws->where[1] = row; ws->where[2] = ( need << 16 ) | ( seen & 0xFFFF );
    ws->dump[128] = spins; ws->dump[129] = static_cast<int>( waited / 100000ull ); ws->dump[130] = mbh;      // (ms)
  }
}
// Test hooks riding in k_loopfilter_rows4's `dbg` argument (ALFALFA_AMD_LF_DEBUG, see lf_debug_bits): bit 5 = FAULT INJECTION -- row 1 of every
// unit never publishes its progress, so row 2 waits until its wait expires (tests/test_gpu_parity.py: the error, its dump and that nothing
// hangs); bits 8-15 = the wait's time bound in quarters of a second (0: kMaxWaitTicks), bits 16-20 = log2 of its poll bound (0: kMinPolls).
__device__ __forceinline__ bool lf_fault_injected( const int dbg, const int row ) { return ( dbg & 32 ) && row == 1; }
__device__ __forceinline__ unsigned long long lf_max_wait_ticks( const int dbg ) { const int q = ( dbg >> 8 ) & 255; return q ? static_cast<unsigned long long>( q ) * 25000000ull : kMaxWaitTicks; }
__device__ __forceinline__ int lf_min_polls( const int dbg ) { const int e = ( dbg >> 16 ) & 31; return e ? 1 << e : kMinPolls; }
__device__ __forceinline__ int xcc_id() { return static_cast<int>( __builtin_amdgcn_s_getreg( 20 | ( 0 << 6 ) | ( 3 << 11 ) ) ); }   // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ int take_ticket( aa_sync_ws * ws, const int xcc, int * slot, const int lane )
{
  __syncthreads();                 // the previous ticket's readers are done with *slot
  if ( lane == 0 ) *slot = __hip_atomic_fetch_add( &ws->ticket[xcc], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP );   // performed in this XCD's L2
  __syncthreads();
  return *slot;
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

// ---- row-pipelined intra prediction, FOUR frames per wave ---------------------------------------------------------------
// 16 lanes per frame ("slot"); each slot walks the intra macroblocks of ITS frame's row on its own (key frames: all four
// in lock step; inter frames: whatever sparse columns each frame has).  A 4x4 sub-block is 16 pixels = 16 lanes, a 16x16
// prediction is 16 rows = 16 lanes, the two chroma planes are 2 x 8 rows = 16 lanes, the 16 luma IDCTs are 16 lanes: the
// per-macroblock version of this kernel kept 16 of 64 lanes busy through the serial B_PRED chain.  The 4x4 predictors are
// evaluated from a table (vp8_math.hh bpred_entry) so that four macroblocks with four different modes share one
// instruction stream; each IDCT runs both passes in one lane's registers (no LDS round trip, no barrier).
struct alignas( 16 ) Intra4Slot {
  alignas( 16 ) int16_t res[24][16];   // residual of block b at [row*4+col]
  alignas( 16 ) uint8_t y[17][48];     // [row+1][col+16]: row 0 = the row above (cols -4..19), byte 15 = the column to the left
  alignas( 16 ) uint8_t c[2][9][32];   // chroma likewise: cols -4..7 at bytes 12..23
  alignas( 16 ) int16_t y2[32];        // Y2: dequantised coefficients / block DCs [0..15], first-pass results [16..31]
};
struct alignas( 16 ) Intra4Lds { Intra4Slot slot[4]; uint32_t tab[160]; };

// Residual of up to four macroblocks (one per 16-lane slot): dequantise, inverse WHT of Y2 (through S.y2), then whole
// 4x4 inverse DCTs in each lane's registers -- 16 luma blocks = 16 lanes, then 8 chroma blocks -- into S.res[block][row*4+col].
// Macroblock::apply_walsh / DCTCoefficients::{dequantize,iwht,idct_add} (macroblock.cc:504-521, quantization.cc:95-126,
// transform.cc:47-137).  Called by the whole wave (contains barriers).
// `off_hi`: the record's `reserved` byte -- packed storage: bits 32-39 of the macroblock's word offset (coeff_index = bits 0-31)
template <class Slot>
__device__ __forceinline__ void residual_x4( Slot & S, const aa_dev_frame & f, const bool has_res, const bool has_y2, const uint32_t nz_mask,
                                             const uint32_t coeff_index, const uint32_t off_hi, const int segment, const int l )
{
  if ( __any( has_res ) ) {
    const bool pk = f.packed != 0;            // (the same for the 16 lanes of a slot; slots of one wave may differ: k_recon_intra4)
    const int16_t * const src = pk ? f.coeffs + ( static_cast<size_t>( off_hi ) << 32 | coeff_index ) : f.coeffs + static_cast<size_t>( coeff_index ) * 16;
    const uint16_t * const q = f.quant[segment];
    const bool y2_stored = has_y2 && ( ( nz_mask >> 24 ) & 1u );
    // packed: where the values of the blocks start = behind the mask slots and the values of the blocks before them in parse order
    // (Y2, then Y in raster order, then U, V)
    uint32_t vbase = pack::kMaskSlots;
    if ( __any( y2_stored ) ) {
      if ( y2_stored ) {
        if ( pk ) {
          const uint32_t m24 = static_cast<uint16_t>( src[24] );
          const uint32_t zz = pack::zigzag_of( static_cast<uint32_t>( l ) );
          const int c = ( ( m24 >> zz ) & 1u ) ? src[pack::kMaskSlots + __popc( m24 & ( ( 1u << zz ) - 1u ) )] : 0;
          S.y2[l] = static_cast<int16_t>( dequant( c, q[l ? 3 : 2] ) );
          vbase += __popc( m24 );
        } else S.y2[l] = static_cast<int16_t>( dequant( src[l], q[l ? 3 : 2] ) );     // a stored Y2 block comes first
      }
      __syncthreads();
      if ( y2_stored && l < 4 ) {
        const Quad v = iwht_pass1( S.y2[l], S.y2[l + 4], S.y2[l + 8], S.y2[l + 12] );
        S.y2[16 + l] = static_cast<int16_t>( v.v0 ); S.y2[16 + l + 4] = static_cast<int16_t>( v.v1 );
        S.y2[16 + l + 8] = static_cast<int16_t>( v.v2 ); S.y2[16 + l + 12] = static_cast<int16_t>( v.v3 );
      }
      __syncthreads();
      if ( y2_stored && l < 4 ) {
        const int o = l * 4;
        const Quad v = iwht_pass2( S.y2[16 + o], S.y2[16 + o + 1], S.y2[16 + o + 2], S.y2[16 + o + 3] );
        S.y2[o] = static_cast<int16_t>( v.v0 ); S.y2[o + 1] = static_cast<int16_t>( v.v1 );
        S.y2[o + 2] = static_cast<int16_t>( v.v2 ); S.y2[o + 3] = static_cast<int16_t>( v.v3 );
      }
      __syncthreads();
    }
#pragma unroll
    for ( int round = 0; round < 2; round++ ) {
      const int blk = round == 0 ? l : 16 + ( l & 7 );
      const bool mine = has_res && ( round == 0 || l < 8 );
      const bool stored = mine && ( ( nz_mask >> blk ) & 1u );
      const bool wht_dc = round == 0 && has_y2;
      const int dc = ( wht_dc && y2_stored ) ? S.y2[blk] : 0;
      if ( !__any( mine ) ) continue;
      if ( __any( stored ) ) {
        uint32_t d[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
        // packed: this lane's mask, and -- a prefix sum over the slot's lanes -- where its values start
        const uint32_t pmask = ( stored && pk ) ? static_cast<uint16_t>( src[blk] ) : 0u;
        const int pcount = __popc( pmask ), pincl = slot_scan16( pcount );
        if ( __any( stored && pk ) ) load_packed_block( pmask, src + vbase + ( pincl - pcount ), d );
        vbase += static_cast<uint32_t>( __shfl( pincl, 15, 16 ) );           // (the chroma round: behind all the luma values)
        if ( stored && !pk ) {
          const uint4 * p = reinterpret_cast<const uint4 *>( src + ( __popc( nz_mask & ( ( 1u << blk ) - 1u ) ) + static_cast<int>( nz_mask >> 24 ) ) * 16 );
          const uint4 a = p[0], b = p[1];
          d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
        }
        const int base = round == 0 ? 0 : 4;
        int r[16];
        idct_block_regs( d, q[base], q[base + 1], wht_dc, dc, r );
        if ( mine ) {
          uint32_t o[8];
#pragma unroll
          for ( int i = 0; i < 8; i++ ) o[i] = ( static_cast<uint32_t>( r[2 * i] ) & 0xFFFFu ) | ( static_cast<uint32_t>( r[2 * i + 1] ) << 16 );
          uint4 * dst = reinterpret_cast<uint4 *>( &S.res[blk][0] );
          dst[0] = make_uint4( o[0], o[1], o[2], o[3] ); dst[1] = make_uint4( o[4], o[5], o[6], o[7] );
        }
      } else if ( mine ) {
        // no block of this round carries coefficients in any of the four macroblocks (low-entropy streams: Y2-only or
        // chroma-less macroblocks): the inverse DCT of a block that is nothing but its DC is that DC, (dc + 4) >> 3,
        // everywhere (both passes of transform.cc:100-137 with c[1..15] = 0)
        const pk2 v = pk_splat( ( static_cast<int16_t>( dc ) + 4 ) >> 3 );
        uint4 * dst = reinterpret_cast<uint4 *>( &S.res[blk][0] );
        dst[0] = make_uint4( v, v, v, v ); dst[1] = make_uint4( v, v, v, v );
      }
    }
  }
}

// n pixels of a 16x16 / 8x8 prediction row (+ residual): above = packed above pixels, res = pointer to 4 int16 residuals
__device__ __forceinline__ uint32_t bigpred_x4( const int mode, const uint32_t above, const int left, const int corner, const int dc, const bool has_res, const int16_t * res )
{
  uint32_t out = 0;
#pragma unroll
  for ( int j = 0; j < 4; j++ ) {
    int v = bigpred_pixel( mode, ( above >> ( 8 * j ) ) & 0xFF, left, corner, dc );
    if ( has_res ) v = clamp255( v + res[j] );
    out |= static_cast<uint32_t>( v ) << ( 8 * j );
  }
  return out;
}

__device__ __forceinline__ void recon_intra4_row( const aa_frame_list & list, const int group, const int row, const int mbh_max, aa_sync_ws * ws, Intra4Lds & L,
                                                  const int home_xcc )
{
  const int lane = threadIdx.x, slot = lane >> 4, l = lane & 15;
  Intra4Slot & S = L.slot[slot];
  const aa_dev_frame * const fp = list.f[group * 4 + slot];
  const aa_dev_frame & f = fp ? *fp : *list.f[group * 4];
  const bool frame_on = fp != nullptr && row < fp->mbh;
  const int mbw = f.mbw, pw = mbw * 16, cw = pw >> 1;
  const int y0 = row * 16, cy0 = row * 8;
  int * const progress = ws->progress + ( group * 4 + slot ) * mbh_max;
  uint8_t * const Y = f.cur[0];
  uint8_t * const Cl = f.cur[1 + ( l >> 3 )];       // the chroma plane this lane serves in the left-column and row roles
  const int words = ( frame_on && f.has_intra ) ? ( mbw + 63 ) >> 6 : 0;
  const unsigned long long * const mask = f.intra_rows + static_cast<size_t>( row ) * ( ( mbw + 63 ) >> 6 );
  int w = 0;
  unsigned long long m = words ? mask[0] : 0ull;
  bool row_done = false;
  int seen = row > 0 ? 0 : 0x7FFFFFFF;     // columns of the row above known to be final (progress only grows)

  for ( ;; ) {
    while ( m == 0 && w + 1 < words ) { ++w; m = mask[w]; }
    const bool on = m != 0;
    if ( !__any( on ) ) break;
    const int col = on ? w * 64 + __ffsll( static_cast<long long>( m ) ) - 1 : 0;
    m &= m - 1;
    // everything left of `col` in this row is final: the previous macroblock's stores have reached the L2.  A slot that
    // has run out of intra macroblocks publishes the whole row at once -- it must not hold the rows below it back until
    // the OTHER three frames of the wave are through.
    asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
    if ( frame_on && l == 0 && ( on || !row_done ) ) __hip_atomic_store( &progress[row], on ? col : mbw, __ATOMIC_RELAXED, AA_PUBLISH_SCOPE );
    row_done = row_done || !on;

    const aa_mb_info * const mb = f.mbs + static_cast<size_t>( row ) * mbw + col;
    uint4 hd = make_uint4( 0, 0, 0, 0 ), bm = make_uint4( 0, 0, 0, 0 );
    if ( on ) { hd = *reinterpret_cast<const uint4 *>( mb ); bm = *reinterpret_cast<const uint4 *>( mb->u.b_mode ); }
    const int y_mode = hd.x & 0xFF, uv_mode = ( hd.x >> 8 ) & 0xFF, segment = ( hd.x >> 24 ) & 3, flags = hd.y & 0xFF;
    const uint32_t nz_mask = hd.z, coeff_index = hd.w;
    const bool has_res = on && ( flags & AA_MB_HAS_NONZERO );
    const bool has_y2 = has_res && ( flags & AA_MB_HAS_Y2 );
    const int need = min( col + 2, mbw );
    if ( on && row > 0 && seen < need ) seen = poll_progress( &progress[row - 1] );

    // ---- residual: needs no neighbour, runs before the wait for the row above ----
    residual_x4( S, f, has_res, has_y2, nz_mask, coeff_index, hd.y >> 24, segment, l );

    // ---- wait for the row above, then stage the neighbours (sc1 loads: L1-bypassing, served by this XCD's L2) ----
    if ( row > 0 ) {
      int spins = 0;
      unsigned long long wait_t0 = 0;
      while ( !__all( !on || seen >= need ) ) {
        __builtin_amdgcn_s_sleep( 4 );
        if ( on && seen < need ) seen = poll_progress( &progress[row - 1] );
        ++spins;
        if ( ( spins & ( AA_HANDOFF_LOOK_EVERY - 1 ) ) == 0 && on ) seen = reread_progress( ws, &progress[row - 1], seen, need, 1 );
        if ( ( spins & 1023 ) == 0 ) {
          if ( __hip_atomic_load( &ws->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) ) break;
          // the hand-off is only coherent inside the XCD the ticket was taken on: a wave that finds itself elsewhere says so
          if ( xcc_id() != home_xcc ) { if ( lane == 0 && atomicCAS( &ws->error, 0, 4 ) == 0 ) { ws->where[0] = group; ws->where[1] = row; ws->where[2] = ( home_xcc << 16 ) | xcc_id(); } break; }
          const unsigned long long now = wall_clock64();
          if ( !wait_t0 ) wait_t0 = now;
          else if ( spins > kMinPolls && now - wait_t0 > kMaxWaitTicks ) { wait_expired( ws, 1, group, row, need, seen, progress, mbh_max, spins, now - wait_t0 ); break; }
        }
      }

    }
    const int x0 = col * 16, cx0 = col * 8;
    if ( on ) {
      if ( l < 6 ) {               // the row above, cols -4..19 as six dwords (prediction.cc:99-167 edge rules)
        uint32_t v;
        const uint8_t * a = Y + static_cast<size_t>( y0 - 1 ) * pw;
        if ( y0 == 0 ) v = 0x7F7F7F7Fu;
        else if ( l == 0 ) v = x0 > 0 ? load_u32<true>( a + x0 - 4 ) : 0x81818181u;
        else if ( l <= 4 ) v = load_u32<true>( a + x0 + ( l - 1 ) * 4 );
        else if ( x0 + 16 >= pw ) v = 0x01010101u * ( load_u32<true>( a + pw - 4 ) >> 24 );      // replicate: prediction.cc:144-151
        else v = load_u32<true>( a + x0 + 16 );
        *reinterpret_cast<uint32_t *>( &S.y[0][12 + l * 4] ) = v;
      } else if ( l < 12 ) {
        const int pl = ( l - 6 ) / 3, i = ( l - 6 ) % 3;
        const uint8_t * a = f.cur[1 + pl] + static_cast<size_t>( cy0 - 1 ) * cw;
        uint32_t v;
        if ( cy0 == 0 ) v = 0x7F7F7F7Fu;
        else if ( i == 0 ) v = cx0 > 0 ? load_u32<true>( a + cx0 - 4 ) : 0x81818181u;
        else v = load_u32<true>( a + cx0 + ( i - 1 ) * 4 );
        *reinterpret_cast<uint32_t *>( &S.c[pl][0][12 + i * 4] ) = v;
      }
      S.y[l + 1][15] = x0 > 0 ? static_cast<uint8_t>( load_u32<true>( Y + static_cast<size_t>( y0 + l ) * pw + x0 - 4 ) >> 24 ) : 129;
      S.c[l >> 3][( l & 7 ) + 1][15] = cx0 > 0 ? static_cast<uint8_t>( load_u32<true>( Cl + static_cast<size_t>( cy0 + ( l & 7 ) ) * cw + cx0 - 4 ) >> 24 ) : 129;
    }
    __syncthreads();

    // ---- chroma: lane = (plane l >> 3, row l & 7), 8 pixels (prediction.cc:435-450) ----
    if ( on ) {
      const int pl = l >> 3, r = l & 7;
      const uint32_t a0 = *reinterpret_cast<const uint32_t *>( &S.c[pl][0][16] ), a1 = *reinterpret_cast<const uint32_t *>( &S.c[pl][0][20] );
      const int sa = absdiff_sum4( a0 ) + absdiff_sum4( a1 );
      int sl = 0;
#pragma unroll
      for ( int i = 0; i < 8; i++ ) sl += S.c[pl][i + 1][15];
      const int dc = bigpred_dc( sa, sl, row > 0, col > 0, 3 );
      const int corner = S.c[pl][0][15], left = S.c[pl][r + 1][15];
      const int blk = 16 + pl * 4 + ( r >> 2 ) * 2;
      const uint32_t o0 = bigpred_x4( uv_mode, a0, left, corner, dc, has_res, &S.res[blk][( r & 3 ) * 4] );
      const uint32_t o1 = bigpred_x4( uv_mode, a1, left, corner, dc, has_res, &S.res[blk + 1][( r & 3 ) * 4] );
      *reinterpret_cast<uint2 *>( Cl + static_cast<size_t>( cy0 + r ) * cw + cx0 ) = make_uint2( o0, o1 );
    }
    // ---- luma 16x16: lane = row ----
    const bool bp = on && y_mode == B_PRED;
    if ( on && !bp ) {
      const int r = l;
      uint32_t a[4];
      int sa = 0, sl = 0;
#pragma unroll
      for ( int k = 0; k < 4; k++ ) { a[k] = *reinterpret_cast<const uint32_t *>( &S.y[0][16 + 4 * k] ); sa += absdiff_sum4( a[k] ); }
#pragma unroll
      for ( int i = 0; i < 16; i++ ) sl += S.y[i + 1][15];
      const int dc = bigpred_dc( sa, sl, row > 0, col > 0, 4 );
      const int corner = S.y[0][15], left = S.y[r + 1][15];
      uint32_t o[4];
#pragma unroll
      for ( int k = 0; k < 4; k++ ) o[k] = bigpred_x4( y_mode, a[k], left, corner, dc, has_res, &S.res[( r >> 2 ) * 4 + k][( r & 3 ) * 4] );
      *reinterpret_cast<uint4 *>( Y + static_cast<size_t>( y0 + r ) * pw + x0 ) = make_uint4( o[0], o[1], o[2], o[3] );
    }
    // ---- luma B_PRED: 16 sub-blocks in raster order, lane = pixel (macroblock.cc:541-544) ----
    if ( __any( bp ) ) {
#pragma unroll                // sub-block position known at compile time: mode byte, tile addresses and the column-3 rule fold away
      for ( int b = 0; b < 16; b++ ) {
        const int bx = b & 3, by = b >> 2;
        const int ar = by * 4, ac = bx * 4 + 15;          // tile position of (row -1, col -1) of this sub-block
        if ( bp ) {
          const uint32_t word = by == 0 ? bm.x : ( by == 1 ? bm.y : ( by == 2 ? bm.z : bm.w ) );
          const int mode = ( word >> ( 8 * bx ) ) & 0xFF;
          // L.tab: bpred_entry with each E index already turned into a tile offset from (row -1, col -1) of the sub-block:
          // i < 4 the column to the left (bottom up), 4 the corner, 5..12 the row above -- of which 9..12 (above-right,
          // flag bit) come from the row above the MACROBLOCK in sub-block column 3 (prediction.cc:153-160).  A sub-block's
          // taps never lie inside the sub-block itself, so one barrier per step.
          const uint32_t e = L.tab[mode * 16 + l];
          const uint8_t * const org = &S.y[ar][ac];
          const int lift = bx == 3 ? ar * 48 : 0;
          const int e0 = org[static_cast<int>( e & 0xFF ) - ( ( e >> 24 ) & 1 ? lift : 0 )];
          const int e1 = org[static_cast<int>( ( e >> 8 ) & 0xFF ) - ( ( e >> 25 ) & 1 ? lift : 0 )];
          const int e2 = org[static_cast<int>( ( e >> 16 ) & 0xFF ) - ( ( e >> 26 ) & 1 ? lift : 0 )];
          const uint32_t above4 = *reinterpret_cast<const uint32_t *>( &S.y[ar][ac + 1] );
          const int dc = ( absdiff_sum4( above4 ) + S.y[ar + 1][ac] + S.y[ar + 2][ac] + S.y[ar + 3][ac] + S.y[ar + 4][ac] + 4 ) >> 3;
          int v = bpred_eval( e >> 27, e0, e1, e2, dc );
          if ( has_res ) v = clamp255( v + S.res[b][l] );
          S.y[ar + 1 + ( l >> 2 )][ac + 1 + ( l & 3 )] = static_cast<uint8_t>( v );
        }
        __syncthreads();
      }
      if ( bp ) *reinterpret_cast<uint4 *>( Y + static_cast<size_t>( y0 + l ) * pw + x0 ) = *reinterpret_cast<const uint4 *>( &S.y[l + 1][16] );
    }
    __syncthreads();
  }
  asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
  if ( frame_on && l == 0 && !row_done ) __hip_atomic_store( &progress[row], mbw, __ATOMIC_RELAXED, AA_PUBLISH_SCOPE );
}

__global__ __launch_bounds__( kLanes ) void k_recon_intra4( const aa_frame_list list, const int n_groups, const int mbh_max, aa_sync_ws * ws, const int n_xcd )
{
  __builtin_amdgcn_s_setprio( 3 );      // reconstruction shares the SIMDs with seconds-long entropy-decode waves: win the issue arbitration

  __shared__ Intra4Lds L;
  __shared__ int s_ticket;
  for ( int i = threadIdx.x; i < 160; i += kLanes ) {       // (mode, pixel) -> three tile offsets, above-right flags, kind
    const uint32_t e = bpred_entry( i >> 4, i & 3, ( i >> 2 ) & 3 );
    uint32_t packed = ( e >> 24 ) << 27;
    for ( int k = 0; k < 3; k++ ) {
      const uint32_t idx = ( e >> ( 8 * k ) ) & 0xFF;
      packed |= ( idx < 4 ? ( 4 - idx ) * 48 : idx - 4 ) << ( 8 * k );
      packed |= ( idx >= 9 ? 1u : 0u ) << ( 24 + k );
    }
    L.tab[i] = packed;
  }
  const int xcc = xcc_id();
  if ( xcc >= n_xcd ) { if ( threadIdx.x == 0 ) atomicExch( &ws->error, 3 ); return; }
  for ( ;; ) {
    const int t = take_ticket( ws, xcc, &s_ticket, threadIdx.x );
    const int mine = xcc < n_groups ? ( n_groups - xcc + n_xcd - 1 ) / n_xcd : 0;     // groups of this XCD: xcc, xcc + n_xcd, ...
    if ( t >= mine * mbh_max ) return;
    recon_intra4_row( list, ( t % mine ) * n_xcd + xcc, t / mine, mbh_max, ws, L, xcc );    // ROW-major: see take_ticket
  }
}

// ---- inter macroblocks, FOUR per wave ----------------------------------------------------------------------------------
// 16 lanes per macroblock ("slot"), four consecutive macroblocks of a frame per wave.  The one-macroblock-per-wave body
// above (still used for SPLITMV) left lanes idle in every phase -- 96 IDCT tasks, 136 + 96 filter tasks and 96 output
// dwords on 64 lanes, the inverse WHT on 4 -- and paid its scalar set-up per macroblock.  Here:
//   * residual: each lane runs whole 4x4 IDCTs in its registers (16 luma blocks = 16 lanes, 8 chroma blocks = 8 lanes);
//   * six-tap passes: luma and chroma tasks of a macroblock form ONE list (136 horizontal, 96 vertical) walked 16 at a
//     time with per-lane source / taps / destination, so partially filled rounds are shared by all four macroblocks;
//   * output: lane = pixel row (luma 16 B per lane, chroma 8 B per lane).
// Round 6: what is never alive at the same time shares its bytes -- the prediction takes the place of the reference windows (which
// the horizontal pass and the copies have read, into the transposed buffers and into registers, by the time it is written), the Y2
// scratch of the residual stage that of the first-pass output.  2 320 instead of 2 776 bytes per slot: 9 344 B per workgroup, so that
// SIX of them fit into the 58.6 KB of LDS the token workers leave of a CU where five of 11 200 B did (the kernel is latency bound:
// beside the workers its pace is the number of its waves a CU holds).
struct alignas( 16 ) Inter4Slot {
  alignas( 16 ) int16_t res[24][16];        // residual of block b at [row*4+col]
  union {
    struct {
      alignas( 16 ) uint8_t wy[21][24];     // luma reference window rows -2..18, 24 bytes from the aligned column
      uint8_t wc[2][13][16];                // chroma windows (directly follows wy: the staging loop treats both as one dword array)
    };
    alignas( 16 ) uint8_t pred[384];        // Y 16x16 | U 8x8 | V 8x8, row-major (born when the windows are dead)
  };
  union {
    alignas( 16 ) uint8_t ty[16][24];       // first-pass output, TRANSPOSED (column-major): the vertical pass also reads consecutive bytes
    alignas( 16 ) int16_t y2[32];           // (residual stage only)
  };
  alignas( 16 ) uint8_t tc[2][8][16];
};
struct alignas( 16 ) Inter4Lds { Inter4Slot slot[4]; uint32_t taps[16]; };
static_assert( offsetof( Inter4Slot, wc ) == offsetof( Inter4Slot, wy ) + 21 * 24, "wy and wc must be contiguous" );

// `bq` = workgroup index within the frame's run of blocks (XCD-aware order); macroblocks 4 q .. 4 q + 3
__device__ __forceinline__ void recon_inter4_body( const aa_dev_frame & f, const unsigned bq, const unsigned max_quads, Inter4Lds & L )
{
  const int lane = threadIdx.x, slot = lane >> 4, l = lane & 15;
  Inter4Slot & S = L.slot[slot];
  if ( lane < 16 ) {            // packed six-tap coefficients of the 8 fractions: taps[2 f] = t0..t3, taps[2 f + 1] = t4,t5
    uint32_t a, b; pack_taps( lane >> 1, a, b ); L.taps[lane] = ( lane & 1 ) ? b : a;
  }
  const unsigned total = static_cast<unsigned>( f.mbw ) * f.mbh;
  const unsigned chunk = ( max_quads + 7u ) >> 3;          // each XCD (block b -> XCD b % 8) gets a contiguous run of quads
  const unsigned qi = ( bq & 7u ) * chunk + ( bq >> 3 );
  const unsigned mi = qi * 4u + slot;
  const int mbw = f.mbw, pw = mbw * 16, ph = f.mbh * 16, cw = pw >> 1, ch = ph >> 1;
  bool on = qi < max_quads && mi < total;
  const aa_mb_info * const mb = f.mbs + ( on ? mi : 0u );
  uint4 hd = make_uint4( 0, 0, 0, 0 ); uint32_t mvw = 0;
  if ( on ) { hd = *reinterpret_cast<const uint4 *>( mb ); mvw = *reinterpret_cast<const uint32_t *>( &mb->u.mv[0][0] ); }
  const int y_mode = hd.x & 0xFF, ref_frame = ( hd.x >> 16 ) & 3, segment = ( hd.x >> 24 ) & 3, flags = hd.y & 0xFF;
  const uint32_t nz_mask = hd.z, coeff_index = hd.w;
  on = on && ( flags & AA_MB_INTER ) && y_mode != SPLITMV;
  if ( !__any( on ) ) return;
  const int col = on ? static_cast<int>( mi % static_cast<unsigned>( mbw ) ) : 0, row = on ? static_cast<int>( mi / static_cast<unsigned>( mbw ) ) : 0;
  const bool has_res = on && ( flags & AA_MB_HAS_NONZERO );
  const bool has_y2 = has_res && ( flags & AA_MB_HAS_Y2 );
  const uint8_t * const * ref = f.ref[on ? ref_frame : 1];
  const uint8_t * const ref_y = ref[0], * const ref_u = ref[1], * const ref_v = ref[2];
  // Y 16x16 with the MB vector, U/V 8x8 with the derived chroma vector (macroblock.cc:583-586)
  const int mvx = static_cast<int16_t>( mvw & 0xFFFFu ), mvy = static_cast<int>( mvw ) >> 16;
  const int cmx = chroma_mv( 4 * mvx ), cmy = chroma_mv( 4 * mvy );
  const int sxy = col * 16 + ( mvx >> 3 ) - 2, syy = row * 16 + ( mvy >> 3 ) - 2;     // window origins (Q9: arithmetic shift)
  const int sxc = col * 8 + ( cmx >> 3 ) - 2, syc = row * 8 + ( cmy >> 3 ) - 2;
  // aligned-dword staging when the windows lie inside the planes, else byte-wise coordinate clamping (EdgeExtendedRaster)
  const int axy = sxy & ~3, axc = sxc & ~3;
  const bool inside = sxy >= 0 && syy >= 0 && axy + 24 <= pw && syy + 21 <= ph && sxc >= 0 && syc >= 0 && axc + 16 <= cw && syc + 13 <= ch;
  const bool fast = on && inside;
  // ---- reference windows: 126 + 104 dwords per macroblock, 15 per lane; issued before the residual work ----
  // dword i = l + 16 k of the macroblock's 230: luma row i / 6, dword i % 6; written so that the per-k part is constant:
  // i = 6 (l/6 + 2k) + (l%6 + 4k)  ->  row = l/6 + 2k + (4k)/6 + (l%6 + (4k)%6 >= 6), dword likewise
  uint32_t wreg[15];
  const int l6 = l / 6, lm = l - 6 * l6;
  const uint8_t * const wy0 = ref_y + static_cast<ptrdiff_t>( syy ) * pw + axy;
  const uint8_t * const wu0 = ref_u + static_cast<ptrdiff_t>( syc ) * cw + axc;
  const uint8_t * const wv0 = ref_v + static_cast<ptrdiff_t>( syc ) * cw + axc;
#pragma unroll
  for ( int k = 0; k < 15; k++ ) {
    const int i = l + 16 * k;
    wreg[k] = 0;
    if ( fast ) {
      if ( k < 7 || ( k == 7 && i < 126 ) ) {
        const int carry = lm + ( 4 * k ) % 6 >= 6 ? 1 : 0;
        const int r = l6 + 2 * k + ( 4 * k ) / 6 + carry, d = lm + ( 4 * k ) % 6 - 6 * carry;
        wreg[k] = *reinterpret_cast<const uint32_t *>( wy0 + static_cast<ptrdiff_t>( r ) * pw + d * 4 );
      } else if ( i < 230 ) {
        const int j = i - 126, pl = j >= 52 ? 1 : 0, e = j - 52 * pl;
        wreg[k] = *reinterpret_cast<const uint32_t *>( ( pl ? wv0 : wu0 ) + static_cast<ptrdiff_t>( e >> 2 ) * cw + ( e & 3 ) * 4 );
      }
    }
  }

  // ---- residual ----
  residual_x4( S, f, has_res, has_y2, nz_mask, coeff_index, hd.y >> 24, segment, l );

  // ---- windows -> LDS ----
  {
    uint32_t * flat = reinterpret_cast<uint32_t *>( &S.wy[0][0] );          // wy (126 dwords) is directly followed by wc (104 dwords)
    if ( fast ) {
#pragma unroll
      for ( int k = 0; k < 15; k++ ) { const int i = l + 16 * k; if ( i < 230 ) flat[i] = wreg[k]; }
    }
    if ( __any( on && !inside ) ) {
      if ( on && !inside ) {
        uint8_t * by = &S.wy[0][0];
        for ( int i = l; i < 21 * 24; i += 16 ) {
          const int r = i / 24, c = i % 24;
          by[i] = ref_y[static_cast<size_t>( clampi( syy + r, 0, ph - 1 ) ) * pw + clampi( sxy + c, 0, pw - 1 )];
        }
        uint8_t * bc = &S.wc[0][0][0];
        for ( int i = l; i < 2 * 13 * 16; i += 16 ) {
          const int pl = i / 208, e = i % 208, r = e >> 4, c = e & 15;
          bc[i] = ( pl ? ref_v : ref_u )[static_cast<size_t>( clampi( syc + r, 0, ch - 1 ) ) * cw + clampi( sxc + c, 0, cw - 1 )];
        }
      }
    }
  }
  const int oy = inside ? ( sxy & 3 ) : 0, oc = inside ? ( sxc & 3 ) : 0;
  __syncthreads();

  // A plane whose vector has no fractional part is a COPY of the window (both six-tap passes are the identity then: the
  // reference always runs them, prediction.cc:875-915, libvpx-style decoders branch to a copy).  Decided per wave: the
  // general task lists below only run for the planes that some macroblock of the wave really filters.
  const bool luma_general = __any( on && ( ( mvx | mvy ) & 7 ) != 0 ), chroma_general = __any( on && ( ( cmx | cmy ) & 7 ) != 0 );
  const uint8_t * const wyb = &S.wy[0][0]; const uint8_t * const wcb = &S.wc[0][0][0];
  uint8_t * const tyb = &S.ty[0][0]; uint8_t * const tcb = &S.tc[0][0][0];
  // ---- horizontal pass: 84 luma + 52 chroma tasks of four outputs; results to the transposed buffers t[column][row].
  //      Two unrolled task lists (instead of one 136-task loop) so that plane, strides and tap registers are compile-time ----
  if ( on && luma_general ) {
    const int frac = mvx & 7;
    const uint32_t t0 = L.taps[2 * frac], t1 = L.taps[2 * frac + 1];
#pragma unroll
    for ( int k = 0; k < 6; k++ ) {
      const int t = l + 16 * k;
      if ( k == 5 && t >= 84 ) continue;
      const int r = t >> 2, g = t & 3;
      const uint8_t * sp = wyb + r * 24 + g * 4;
      const uint32_t o4 = sixtap_x4_lane( *reinterpret_cast<const uint32_t *>( sp ), *reinterpret_cast<const uint32_t *>( sp + 4 ), *reinterpret_cast<const uint32_t *>( sp + 8 ), oy, frac, t0, t1 );
      uint8_t * tb = tyb + ( g * 4 ) * 24 + r;
      tb[0] = static_cast<uint8_t>( o4 ); tb[24] = static_cast<uint8_t>( o4 >> 8 ); tb[48] = static_cast<uint8_t>( o4 >> 16 ); tb[72] = static_cast<uint8_t>( o4 >> 24 );
    }
  }
  if ( on && chroma_general ) {
    const int frac = cmx & 7;
    const uint32_t t0 = L.taps[2 * frac], t1 = L.taps[2 * frac + 1];
#pragma unroll
    for ( int k = 0; k < 4; k++ ) {
      const int j = l + 16 * k;
      if ( k == 3 && j >= 52 ) continue;
      const int pl = j >= 26 ? 1 : 0, e = j - 26 * pl, r = e >> 1, g = e & 1;
      const uint8_t * sp = wcb + pl * 208 + r * 16 + g * 4;
      const uint32_t o4 = sixtap_x4_lane( *reinterpret_cast<const uint32_t *>( sp ), *reinterpret_cast<const uint32_t *>( sp + 4 ), *reinterpret_cast<const uint32_t *>( sp + 8 ), oc, frac, t0, t1 );
      uint8_t * tb = tcb + pl * 128 + ( g * 4 ) * 16 + r;
      tb[0] = static_cast<uint8_t>( o4 ); tb[16] = static_cast<uint8_t>( o4 >> 8 ); tb[32] = static_cast<uint8_t>( o4 >> 16 ); tb[48] = static_cast<uint8_t>( o4 >> 24 );
    }
  }
  // ---- planes that are copied: their window rows into registers now (the prediction shares the windows' bytes: nothing of it may be
  //      written before every lane has read what it needs of the windows -- the horizontal pass above, the copies here) ----
  uint4 copy_y = make_uint4( 0, 0, 0, 0 ); uint2 copy_c = make_uint2( 0, 0 );
  if ( !luma_general && on ) {          // lane = pixel row: bytes oy+2 .. oy+17 of window row l+2
    const uint2 * rp = reinterpret_cast<const uint2 *>( wyb + ( l + 2 ) * 24 );
    const uint2 a = rp[0], b = rp[1], c = rp[2];
    const bool q = oy + 2 >= 4; const int sft = ( oy + 2 ) & 3;
    const uint32_t e0 = q ? a.y : a.x, e1 = q ? b.x : a.y, e2 = q ? b.y : b.x, e3 = q ? c.x : b.y, e4 = q ? c.y : c.x;
    copy_y = make_uint4( __builtin_amdgcn_alignbyte( e1, e0, sft ), __builtin_amdgcn_alignbyte( e2, e1, sft ),
                         __builtin_amdgcn_alignbyte( e3, e2, sft ), __builtin_amdgcn_alignbyte( e4, e3, sft ) );
  }
  if ( !chroma_general && on ) {        // lane = (plane, pixel row): bytes oc+2 .. oc+9 of window row r+2
    const int pl = l >> 3, r = l & 7;
    const uint2 * rp = reinterpret_cast<const uint2 *>( wcb + pl * 208 + ( r + 2 ) * 16 );
    const uint2 a = rp[0], b = rp[1];
    const bool q = oc + 2 >= 4; const int sft = ( oc + 2 ) & 3;
    const uint32_t e0 = q ? a.y : a.x, e1 = q ? b.x : a.y, e2 = q ? b.y : b.x;
    copy_c = make_uint2( __builtin_amdgcn_alignbyte( e1, e0, sft ), __builtin_amdgcn_alignbyte( e2, e1, sft ) );
  }
  __syncthreads();
  if ( !luma_general && on ) *reinterpret_cast<uint4 *>( S.pred + l * 16 ) = copy_y;
  if ( !chroma_general && on ) *reinterpret_cast<uint2 *>( S.pred + 256 + ( l >> 3 ) * 64 + ( l & 7 ) * 8 ) = copy_c;
  // ---- vertical pass: 64 luma + 32 chroma tasks: (column c, row group i) -> rows 4i..4i+3 of column c ----
  if ( on && luma_general ) {
    const int frac = mvy & 7;
    const uint32_t t0 = L.taps[2 * frac], t1 = L.taps[2 * frac + 1];
#pragma unroll
    for ( int k = 0; k < 4; k++ ) {
      const int t = l + 16 * k, c = t >> 2, i = t & 3;
      const uint8_t * sp = tyb + c * 24 + i * 4;
      const uint32_t o4 = sixtap_x4_lane( *reinterpret_cast<const uint32_t *>( sp ), *reinterpret_cast<const uint32_t *>( sp + 4 ), *reinterpret_cast<const uint32_t *>( sp + 8 ), 0, frac, t0, t1 );
      uint8_t * pb = S.pred + ( i * 4 ) * 16 + c;
      pb[0] = static_cast<uint8_t>( o4 ); pb[16] = static_cast<uint8_t>( o4 >> 8 ); pb[32] = static_cast<uint8_t>( o4 >> 16 ); pb[48] = static_cast<uint8_t>( o4 >> 24 );
    }
  }
  if ( on && chroma_general ) {
    const int frac = cmy & 7;
    const uint32_t t0 = L.taps[2 * frac], t1 = L.taps[2 * frac + 1];
#pragma unroll
    for ( int k = 0; k < 2; k++ ) {
      const int j = l + 16 * k, pl = j >> 4, c = ( j >> 1 ) & 7, i = j & 1;
      const uint8_t * sp = tcb + pl * 128 + c * 16 + i * 4;
      const uint32_t o4 = sixtap_x4_lane( *reinterpret_cast<const uint32_t *>( sp ), *reinterpret_cast<const uint32_t *>( sp + 4 ), *reinterpret_cast<const uint32_t *>( sp + 8 ), 0, frac, t0, t1 );
      uint8_t * pb = S.pred + 256 + pl * 64 + ( i * 4 ) * 8 + c;
      pb[0] = static_cast<uint8_t>( o4 ); pb[8] = static_cast<uint8_t>( o4 >> 8 ); pb[16] = static_cast<uint8_t>( o4 >> 16 ); pb[24] = static_cast<uint8_t>( o4 >> 24 );
    }
  }
  __syncthreads();

  // ---- prediction + residual -> raster: lane = luma row l (16 B), and chroma plane l >> 3 row l & 7 (8 B) ----
  if ( on ) {
    uint32_t o[4];
#pragma unroll
    for ( int k = 0; k < 4; k++ ) {
      const uint32_t p4 = *reinterpret_cast<const uint32_t *>( S.pred + l * 16 + 4 * k );
      const uint32_t out = has_res ? add_residual_x4( p4, &S.res[( l >> 2 ) * 4 + k][( l & 3 ) * 4] ) : p4;
      o[k] = out;
    }
    *reinterpret_cast<uint4 *>( f.cur[0] + static_cast<size_t>( row * 16 + l ) * pw + col * 16 ) = make_uint4( o[0], o[1], o[2], o[3] );
    const int pl = l >> 3, r = l & 7;
    uint32_t oc2[2];
#pragma unroll
    for ( int k = 0; k < 2; k++ ) {
      const uint32_t p4 = *reinterpret_cast<const uint32_t *>( S.pred + 256 + pl * 64 + r * 8 + 4 * k );
      const uint32_t out = has_res ? add_residual_x4( p4, &S.res[16 + pl * 4 + ( r >> 2 ) * 2 + k][( r & 3 ) * 4] ) : p4;
      oc2[k] = out;
    }
    *reinterpret_cast<uint2 *>( f.cur[1 + pl] + static_cast<size_t>( row * 8 + r ) * cw + col * 8 ) = make_uint2( oc2[0], oc2[1] );
  }
}

// grid.x = quad of macroblocks (XCD-aware order), grid.y = frame in batch
__global__ __launch_bounds__( kLanes ) void k_recon_inter4( const aa_frame_list list, const unsigned max_quads )
{
  __builtin_amdgcn_s_setprio( 3 );      // reconstruction shares the SIMDs with seconds-long entropy-decode waves: win the issue arbitration

  __shared__ Inter4Lds L;
  recon_inter4_body( *list.f[blockIdx.y], blockIdx.x, max_quads, L );
}

// ---- row-pipelined loop filter: packed arithmetic, FOUR frames per wave, strip-staged I/O ----------------------------
// 16 lanes per frame ("slot").  A lane filters TWO positions of an edge at once in packed int16 (vp8_math.hh pk2):
//   vertical edges   (V phase): lane j < 8 owns luma pixel rows 2j, 2j+1; lane 8+k owns chroma rows 2(k&3), +1 of plane k>>2;
//   horizontal edges (H phase): lane j < 8 owns luma columns 2j, 2j+1;    lane 8+k the chroma columns likewise.
// A lane keeps its two 20-pixel lines (12 for chroma) in registers for all four (two) edges of a phase.  Chroma lanes
// run the SAME instruction stream as luma lanes (same LDS row stride, edges 2 and 3 gated off).
//
// Memory traffic is what bounded the per-macroblock version of this kernel (every lane of every load/store touched its
// own 128-byte line: ~130 L2 requests per macroblock).  Here a workgroup stages a STRIP of eight macroblocks (128 luma
// columns = one cache line per pixel row) in LDS:
//   * the strip's own 16 rows are loaded with lanes running ALONG rows (whole lines), one strip ahead, into registers;
//   * the filter phases work in place on the strip;
//   * rows [16r-4, 16r+12) of the frame -- final once this macroblock row has passed -- are stored back whole-line at the
//     end of the strip; only the strip's last four columns, which the next strip's left MB edge still modifies, follow
//     as 4-byte fix-ups;
//   * the bottom four rows (chroma: four) are NOT written to the frame by this row: they go, one 128-byte line per
//     macroblock, to a BOUNDARY buffer; the workgroup of the next macroblock row reads that line as its rows -4..-1,
//     finishes them with its top MB edge and stores them with its own strip.  Frame rows are therefore written by
//     exactly one workgroup and cross-row hand-off costs one line per macroblock each way (+ a 4-byte-column fix-up).
#ifndef AA_LF_STRIP_MBS
#define AA_LF_STRIP_MBS 8
#endif
constexpr int kStripMbs = AA_LF_STRIP_MBS;                         // 8: whole 128-byte lines, 19 KB of LDS per wave (2 waves per SIMD); 4: half lines, 11 KB, 3 waves per SIMD -- measured: same speed, 1.35x the HBM traffic
constexpr int kStripRow = 16 + 16 * kStripMbs;        // LDS row: 16 bytes of padding + the strip's luma columns (U | V halves for chroma rows)
constexpr int kStripRpi = 16 / kStripMbs;             // pixel rows covered by one bulk load/store instruction (16 lanes = chunks x rows)
static_assert( ( kStripRow / 16 ) % 2 == 1, "odd multiple of 16: consecutive rows start on different banks" );
struct alignas( 16 ) LfStrip {
  uint8_t c[12][kStripRow];  // chroma rows -4..7 of the MB row: U strip columns at bytes 16.., V after them
  uint8_t y[20][kStripRow];  // luma rows -4..15: strip columns at bytes 16..
  uint8_t halo_y[16][4];     // columns -4..-1 of rows 0..15: the previous strip's right edge
  uint8_t halo_c[2][8][4];
  uint8_t pad[32];           // the four slots of a wave start on different banks
};
struct alignas( 16 ) LfStripLds { LfStrip slot[4]; };

__device__ __forceinline__ uint64_t load_u64_shared( const uint8_t * p )
{
  return __hip_atomic_load( reinterpret_cast<const uint64_t *>( p ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
}

// the (up to) four edges of a phase on a lane's two lines: v[i] = packed pixels at position -4 + i
__device__ __forceinline__ void lf_edges_pk( pk2 ( &v )[20], const LfParamsPk & P, const pk2 g0, const pk2 g1, const pk2 g23 )
{
  if ( __any( g0 != 0u ) ) lf_edge_pk( P, true, g0, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7] );
  if ( __any( g1 != 0u ) ) {
    lf_edge_pk( P, false, g1, v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11] );
    if ( __any( g23 != 0u ) ) {
      lf_edge_pk( P, false, g23, v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15] );
      lf_edge_pk( P, false, g23, v[12], v[13], v[14], v[15], v[16], v[17], v[18], v[19] );
    }
  }
}

// One macroblock row of a group of four frames of one geometry (list.f[4g] is never null, the host pads with null).
// bnd: boundary buffer, 128 bytes per (frame of the launch, MB row, MB column): luma rows 12..15 x 16 B, U rows 4..7 x 8 B,
// V rows 4..7 x 8 B of that macroblock after its own filtering.
__device__ __forceinline__ void loopfilter_strip_row( const aa_frame_list & list, const int group, const int row, const int mbh_max, const int mbw_max,
                                                       aa_sync_ws * ws, uint8_t * bnd, LfStripLds & S, const int dbg, const int home_xcc )
{
  const int lane = threadIdx.x, slot = lane >> 4, l = lane & 15;
  const aa_dev_frame & f0 = *list.f[group * 4];
  const int mbh = f0.mbh;
  if ( row >= mbh ) return;
  const aa_dev_frame * const fp = list.f[group * 4 + slot];
  const bool frame_on = fp != nullptr && fp->loop_filter_level != 0;
  if ( !__any( frame_on ) ) return;
  const aa_dev_frame & f = fp ? *fp : f0;
  LfStrip & T = S.slot[slot];
  int * const progress = ws->progress + group * mbh_max;
  const int mbw = f0.mbw, pw = mbw * 16, cw = pw >> 1;
  const int y0 = row * 16, cy0 = row * 8;
  const int sharp = f.sharpness; const bool key = f.key_frame;
  const bool last_row = row == mbh - 1;
  const bool luma = l < 8;
  const int k8 = l & 7, cp = k8 >> 2, cj = k8 & 3;     // chroma lanes (l >= 8): plane, line/column pair
  // ---- phase roles (pointers for strip position 0; + 16 k (luma) / 8 k (chroma) per macroblock) ----
  uint8_t * const vrow = luma ? &T.y[4 + 2 * l][16] : &T.c[4 + 2 * cj][16 + 8 * kStripMbs * cp];          // row A; row B at +kStripRow
  uint8_t * const vhalo = luma ? &T.halo_y[2 * l][0] : &T.halo_c[cp][2 * cj][0];                // row A; row B at +4
  uint8_t * const hcol = luma ? &T.y[0][16 + 2 * l] : &T.c[0][16 + 8 * kStripMbs * cp + 2 * cj];          // rows at +kStripRow r
  const int mbstep = luma ? 16 : 8;
  // ---- bulk roles: lanes along rows.  instr i: luma row 2i + (l>>3), 16-byte chunk l&7 (= MB of the strip);
  //      chroma plane i>>2, row 2(i&3) + (l>>3), 8-byte chunk l&7 ----
  const int brow = l / kStripMbs, bchunk = l % kStripMbs;
  uint8_t * const gy = f.cur[0] + static_cast<size_t>( y0 + brow ) * pw + 16 * bchunk;          // + kStripRpi i pw + 16 kStripMbs s
  uint8_t * const gu = f.cur[1] + static_cast<size_t>( cy0 + brow ) * cw + 8 * bchunk;          // + kStripRpi i cw + 8 kStripMbs s
  uint8_t * const gv = f.cur[2] + static_cast<size_t>( cy0 + brow ) * cw + 8 * bchunk;
  uint8_t * const sy = &T.y[4 + brow][16 + 16 * bchunk];                                        // + kStripRpi kStripRow i
  uint8_t * const sc = &T.c[4 + brow][16 + 8 * bchunk];                                         // + kStripRpi kStripRow i (+ 8 kStripMbs for V)
  // ---- boundary roles: lanes 0..3 luma rows 12..15 (16 B), 4,5 U row pairs (2 x 8 B), 6,7 V ----
  const int bl = l & 7, bq = ( bl - 4 ) & 3;
  uint8_t * const bsrc = bl < 4 ? &T.y[16 + bl][16] : &T.c[8 + 2 * ( bq & 1 )][16 + 8 * kStripMbs * ( bq >> 1 )];      // own bottom rows (2nd chroma row at +kStripRow)
  uint8_t * const btop = bl < 4 ? &T.y[bl][16] : &T.c[2 * ( bq & 1 )][16 + 8 * kStripMbs * ( bq >> 1 )];                // rows -4..-1
  const int bstep = bl < 4 ? 16 : 8;
  const size_t bnd_row = ( static_cast<size_t>( group * 4 + slot ) * mbh_max + row ) * mbw_max;             // in 128-byte lines
  uint8_t * const bnd_own = bnd + bnd_row * 128 + 16 * bl;
  const uint8_t * const bnd_top = bnd + ( bnd_row - mbw_max ) * 128 + 16 * bl;
  // fix-up of the previous macroblock's last four columns in its boundary line: lanes 0..3 luma rows 12..15, 4..7 U rows 4..7, 8..11 V
  const int fq = l >> 2, fr = l & 3;
  const int fix_off = fq == 0 ? 16 * fr + 12 : 64 + 32 * ( fq - 1 ) + 8 * fr + 4;
  uint8_t * const fix_src0 = fq == 0 ? &T.halo_y[12 + fr][0] : &T.halo_c[( fq - 1 ) & 1][4 + fr][0];       // strip position 0: the halo
  uint8_t * const fix_srck = fq == 0 ? &T.y[16 + fr][16 - 4] : &T.c[8 + fr][16 + 8 * kStripMbs * ( ( fq - 1 ) & 1 ) - 4];   // + 16 k / 8 k
  const int fix_step = fq == 0 ? 16 : 8;
  // frame fix-up of a finished strip's last four columns: lane l <-> luma row l, chroma plane l>>3 row l&7
  uint8_t * const fy = f.cur[0] + static_cast<size_t>( y0 + l ) * pw - 4;                        // + 16 kStripMbs s
  uint8_t * const fc = f.cur[1 + ( l >> 3 )] + static_cast<size_t>( cy0 + ( l & 7 ) ) * cw - 4;  // + 8 kStripMbs s
  const aa_mb_info * const mbrow = f.mbs + static_cast<size_t>( row ) * mbw;

  const int n_strips = ( mbw + kStripMbs - 1 ) / kStripMbs;
  uint4 py[kStripMbs]; uint2 pc[kStripMbs];
#pragma unroll
  for ( int i = 0; i < kStripMbs; i++ ) { py[i] = make_uint4( 0, 0, 0, 0 ); pc[i] = make_uint2( 0, 0 ); }
  // own rows of strip s -> registers (lanes along rows: whole lines)
  auto prefetch = [&]( const int s ) {
    const bool in = frame_on && !( dbg & 2 ) && s * kStripMbs + bchunk < mbw;
    if ( in ) {
#pragma unroll
      for ( int i = 0; i < kStripMbs; i++ ) py[i] = *reinterpret_cast<const uint4 *>( gy + static_cast<size_t>( kStripRpi * i ) * pw + 16 * kStripMbs * s );
#pragma unroll
      for ( int i = 0; i < kStripMbs / 2; i++ ) pc[i] = *reinterpret_cast<const uint2 *>( gu + static_cast<size_t>( kStripRpi * i ) * cw + 8 * kStripMbs * s );
#pragma unroll
      for ( int i = 0; i < kStripMbs / 2; i++ ) pc[kStripMbs / 2 + i] = *reinterpret_cast<const uint2 *>( gv + static_cast<size_t>( kStripRpi * i ) * cw + 8 * kStripMbs * s );
    }
  };
  prefetch( 0 );
  int info = frame_on ? *reinterpret_cast<const uint16_t *>( &mbrow[0].flags ) : 0;     // flags | lf_level << 8
  int seen = row > 0 ? 0 : mbw;        // boundary lines of the row above known to be complete
  LfParamsPk P = lf_params_pk( lf_params( 1, sharp, key ) );
  int p_level = -1;                    // the level P was computed for
  for ( int s = 0; s < n_strips; s++ ) {
    const int nmb = min( kStripMbs, mbw - s * kStripMbs );
    // ---- strip turn-over: keep the right edge as the new left neighbour, then drop the prefetched rows in ----
    if ( s > 0 ) {
      *reinterpret_cast<uint32_t *>( &T.halo_y[l][0] ) = *reinterpret_cast<const uint32_t *>( &T.y[4 + l][16 + 16 * kStripMbs - 4] );
      *reinterpret_cast<uint32_t *>( &T.halo_c[l >> 3][l & 7][0] ) = *reinterpret_cast<const uint32_t *>( &T.c[4 + ( l & 7 )][16 + 8 * kStripMbs * ( l >> 3 ) + 8 * kStripMbs - 4] );
      __syncthreads();
    }
#pragma unroll
    for ( int i = 0; i < kStripMbs; i++ ) *reinterpret_cast<uint4 *>( sy + kStripRpi * kStripRow * i ) = py[i];
#pragma unroll
    for ( int i = 0; i < kStripMbs / 2; i++ ) { *reinterpret_cast<uint2 *>( sc + kStripRpi * kStripRow * i ) = pc[i]; *reinterpret_cast<uint2 *>( sc + kStripRpi * kStripRow * i + 8 * kStripMbs ) = pc[kStripMbs / 2 + i]; }
    __syncthreads();

    for ( int k = 0; k < nmb; k++ ) {
      const int col = s * kStripMbs + k;
      const int level = info >> 8;
      const bool active = level != 0;
      const bool inner = !( info & AA_MB_LF_SKIP_INNER );
      const bool more = col + 1 < mbw;
      // in flight during the V phase: the next macroblock's level, the poll of the row above
      const int info_next = ( frame_on && more ) ? *reinterpret_cast<const uint16_t *>( &mbrow[col + 1].flags ) : 0;
      const int need = col + 1;           // boundary lines 0..col of the row above complete (the last one only at its row's end)
      // progress only grows: what an earlier poll saw stays valid, so a row that runs well behind the row above polls rarely
      if ( row > 0 && seen < need ) seen = poll_progress( &progress[row - 1] );
      const bool any_active = __any( active );
      if ( !__all( level == p_level ) ) { P = lf_params_pk( lf_params( active ? level : 1, sharp, key ) ); p_level = level; }     // levels rarely change along a row
      const pk2 g_on = active ? ~0u : 0u, g_in = ( active && inner ) ? ~0u : 0u, g_in23 = ( active && inner && luma ) ? ~0u : 0u;

      if ( any_active && !( dbg & 4 ) ) {   // ---- V phase: left MB edge, inner vertical edges ----
        uint8_t * const ra = vrow + mbstep * k;
        uint8_t * const ha = k == 0 ? vhalo : ra - 4;          // columns -4..-1: the halo at strip position 0, else the previous MB
        uint8_t * const hb = k == 0 ? vhalo + 4 : ra + kStripRow - 4;
        uint32_t a[5], b[5];
        a[0] = *reinterpret_cast<const uint32_t *>( ha ); b[0] = *reinterpret_cast<const uint32_t *>( hb );
        { const uint2 u = *reinterpret_cast<const uint2 *>( ra ); a[1] = u.x; a[2] = u.y; }
        { const uint2 u = *reinterpret_cast<const uint2 *>( ra + 8 ); a[3] = u.x; a[4] = u.y; }
        { const uint2 u = *reinterpret_cast<const uint2 *>( ra + kStripRow ); b[1] = u.x; b[2] = u.y; }
        { const uint2 u = *reinterpret_cast<const uint2 *>( ra + kStripRow + 8 ); b[3] = u.x; b[4] = u.y; }
        pk2 v[20];
#pragma unroll
        for ( int d = 0; d < 5; d++ ) {
          v[4 * d] = pk_from_bytes<0>( a[d], b[d] ); v[4 * d + 1] = pk_from_bytes<1>( a[d], b[d] );
          v[4 * d + 2] = pk_from_bytes<2>( a[d], b[d] ); v[4 * d + 3] = pk_from_bytes<3>( a[d], b[d] );
        }
        lf_edges_pk( v, P, col > 0 ? g_on : 0u, g_in, g_in23 );
#pragma unroll
        for ( int d = 0; d < 5; d++ ) pk_to_dwords( v[4 * d], v[4 * d + 1], v[4 * d + 2], v[4 * d + 3], a[d], b[d] );
        if ( active ) {
          *reinterpret_cast<uint32_t *>( ha ) = a[0]; *reinterpret_cast<uint32_t *>( hb ) = b[0];
          *reinterpret_cast<uint2 *>( ra ) = make_uint2( a[1], a[2] ); *reinterpret_cast<uint2 *>( ra + kStripRow ) = make_uint2( b[1], b[2] );
          if ( luma ) { *reinterpret_cast<uint2 *>( ra + 8 ) = make_uint2( a[3], a[4] ); *reinterpret_cast<uint2 *>( ra + kStripRow + 8 ) = make_uint2( b[3], b[4] ); }
        }
      }
      // The left MB edge just run was the last thing to touch the previous macroblock's right columns: complete its
      // boundary line now, so that the row below can use it one step earlier than if this waited for the end of the step.
      __syncthreads();
      if ( frame_on && !last_row && col > 0 && l < 12 && !( dbg & 1 ) ) {
        const uint8_t * src = k == 0 ? fix_src0 : fix_srck + fix_step * k;
        *reinterpret_cast<uint32_t *>( bnd + ( bnd_row + col - 1 ) * 128 + fix_off ) = *reinterpret_cast<const uint32_t *>( src );
      }

      if ( row > 0 ) {
        int spins = 0;
        unsigned long long wait_t0 = 0;
        while ( !__all( seen >= need ) && !( dbg & 16 ) ) {
          __builtin_amdgcn_s_sleep( 4 );
          if ( seen < need ) seen = poll_progress( &progress[row - 1] );
          ++spins;
          if ( ( spins & ( AA_HANDOFF_LOOK_EVERY - 1 ) ) == 0 ) seen = reread_progress( ws, &progress[row - 1], seen, need, 2 );
          if ( ( spins & 1023 ) == 0 ) {
            if ( __hip_atomic_load( &ws->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) ) break;
            // the hand-off is only coherent inside one XCD: a wave that finds itself on another one (context save / restore
            // under queue oversubscription) says so instead of waiting for the watchdog
            if ( xcc_id() != home_xcc ) { if ( lane == 0 && atomicCAS( &ws->error, 0, 4 ) == 0 ) { ws->where[0] = group; ws->where[1] = row; ws->where[2] = ( home_xcc << 16 ) | xcc_id(); } break; }
            const unsigned long long now = wall_clock64();
            if ( !wait_t0 ) wait_t0 = now;
            else if ( spins > lf_min_polls( dbg ) && now - wait_t0 > lf_max_wait_ticks( dbg ) ) { wait_expired( ws, 2, group, row, need, seen, progress, mbh, spins, now - wait_t0 ); break; }
          }
        }
  
        // rows -4..-1: the boundary line the row above left for this macroblock (sc1: bypass L1, served by the XCD's L2)
        if ( frame_on && l < 8 && !( dbg & 8 ) ) {
          const uint8_t * src = bnd_top + static_cast<size_t>( col ) * 128;
          const uint64_t lo = load_u64_shared( src ), hi = load_u64_shared( src + 8 );
          uint8_t * dst = btop + bstep * k;
          if ( bl < 4 ) { *reinterpret_cast<uint64_t *>( dst ) = lo; *reinterpret_cast<uint64_t *>( dst + 8 ) = hi; }
          else { *reinterpret_cast<uint64_t *>( dst ) = lo; *reinterpret_cast<uint64_t *>( dst + kStripRow ) = hi; }
        }
      }
      // every boundary line up to macroblock col-1 is complete and has reached the L2 (the fix-up above drained behind the
      // wait for the row above): publish
      asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
      if ( lane == 0 && col > 0 && !lf_fault_injected( dbg, row ) ) __hip_atomic_store( &progress[row], col, __ATOMIC_RELAXED, AA_PUBLISH_SCOPE );
      // the next strip's own rows: issued here so that no wait of THIS step covers them (vmcnt completes in order); they
      // have the rest of the strip to arrive
      if ( k == 0 && s + 1 < n_strips ) prefetch( s + 1 );
      __syncthreads();

      if ( any_active && !( dbg & 4 ) ) {   // ---- H phase: top MB edge, inner horizontal edges ----
        uint8_t * const hc = hcol + mbstep * k;
        pk2 v[20];
#pragma unroll
        for ( int r = 0; r < 20; r++ ) v[r] = pk_from_u16( *reinterpret_cast<const uint16_t *>( hc + kStripRow * r ) );
        lf_edges_pk( v, P, row > 0 ? g_on : 0u, g_in, g_in23 );
        if ( active ) {
#pragma unroll
          for ( int r = 1; r < 12; r++ ) *reinterpret_cast<uint16_t *>( hc + kStripRow * r ) = static_cast<uint16_t>( pk_to_u16( v[r] ) );
          if ( luma ) {
#pragma unroll
            for ( int r = 12; r < 20; r++ ) *reinterpret_cast<uint16_t *>( hc + kStripRow * r ) = static_cast<uint16_t>( pk_to_u16( v[r] ) );
          }
        }
      }
      __syncthreads();

      if ( frame_on && !( dbg & 1 ) ) {
        if ( !last_row ) {
          // this macroblock's bottom rows -> its boundary line (its last four columns are not final yet: fixed up by the next step)
          if ( l < 8 ) {
            const uint8_t * src = bsrc + bstep * k;
            uint4 q;
            if ( bl < 4 ) q = *reinterpret_cast<const uint4 *>( src );
            else { const uint2 u0 = *reinterpret_cast<const uint2 *>( src ), u1 = *reinterpret_cast<const uint2 *>( src + kStripRow ); q = make_uint4( u0.x, u0.y, u1.x, u1.y ); }
            *reinterpret_cast<uint4 *>( bnd_own + static_cast<size_t>( col ) * 128 ) = q;
          }
        }
        if ( k == 0 && s > 0 ) {
          // the previous strip's last four columns are final now
          if ( l < 12 || last_row ) *reinterpret_cast<uint32_t *>( fy + 16 * kStripMbs * s ) = *reinterpret_cast<const uint32_t *>( &T.halo_y[l][0] );
          if ( ( l & 7 ) < 4 || last_row ) *reinterpret_cast<uint32_t *>( fc + 8 * kStripMbs * s ) = *reinterpret_cast<const uint32_t *>( &T.halo_c[l >> 3][l & 7][0] );
        }
        if ( k == nmb - 1 && bchunk < nmb ) {
          // ---- the strip is done: rows -4..11 (chroma -4..3) back to the frame, whole lines; the last MB row also its bottom rows ----
#pragma unroll
          for ( int i = 0; i < 20 / kStripRpi; i++ ) {          // strip rows kStripRpi i + brow = frame rows y0 - 4 + kStripRpi i + brow
            if ( ( ( i + 1 ) * kStripRpi > 4 || row > 0 ) && ( i * kStripRpi < 16 || last_row ) )
              *reinterpret_cast<uint4 *>( gy + ( static_cast<ptrdiff_t>( kStripRpi * i ) - 4 ) * pw + 16 * kStripMbs * s ) = *reinterpret_cast<const uint4 *>( sy + kStripRpi * kStripRow * i - 4 * kStripRow );
          }
#pragma unroll
          for ( int i = 0; i < 12 / kStripRpi; i++ ) {          // chroma strip rows kStripRpi i + brow = chroma rows cy0 - 4 + kStripRpi i + brow
            if ( ( ( i + 1 ) * kStripRpi > 4 || row > 0 ) && ( i * kStripRpi < 8 || last_row ) ) {
              *reinterpret_cast<uint2 *>( gu + ( static_cast<ptrdiff_t>( kStripRpi * i ) - 4 ) * cw + 8 * kStripMbs * s ) = *reinterpret_cast<const uint2 *>( sc + kStripRpi * kStripRow * i - 4 * kStripRow );
              *reinterpret_cast<uint2 *>( gv + ( static_cast<ptrdiff_t>( kStripRpi * i ) - 4 ) * cw + 8 * kStripMbs * s ) = *reinterpret_cast<const uint2 *>( sc + kStripRpi * kStripRow * i - 4 * kStripRow + 8 * kStripMbs );
            }
          }
        }
      }
      info = info_next;
      __syncthreads();
    }
  }
  asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );       // the last macroblock's line has no right neighbour to wait for
  if ( lane == 0 && !lf_fault_injected( dbg, row ) ) __hip_atomic_store( &progress[row], mbw, __ATOMIC_RELAXED, AA_PUBLISH_SCOPE );
}

__device__ __forceinline__ void loopfilter_rows4_body( const aa_frame_list & list, const int n_groups, const int mbh_max, const int mbw_max, aa_sync_ws * ws, uint8_t * bnd,
                                                        const int n_xcd, LfStripLds & S, int & s_ticket, const int dbg )
{
  const int xcc = xcc_id();
  if ( xcc >= n_xcd ) { if ( threadIdx.x == 0 ) atomicExch( &ws->error, 3 ); return; }
  for ( ;; ) {
    const int t = take_ticket( ws, xcc, &s_ticket, threadIdx.x );
    const int mine = xcc < n_groups ? ( n_groups - xcc + n_xcd - 1 ) / n_xcd : 0;     // groups of this XCD: xcc, xcc + n_xcd, ...
    if ( t >= mine * mbh_max ) return;
    loopfilter_strip_row( list, ( t % mine ) * n_xcd + xcc, t / mine, mbh_max, mbw_max, ws, bnd, S, dbg, xcc );    // ROW-major: see take_ticket
  }
}

// grid.x = n_xcd * ceil(n_groups / n_xcd) * mbh_max workgroups
__global__ __launch_bounds__( kLanes ) __attribute__( ( amdgpu_waves_per_eu( 2 ) ) ) void k_loopfilter_rows4( const aa_frame_list list, const int n_groups, const int mbh_max, const int mbw_max, aa_sync_ws * ws, uint8_t * bnd,
                                                                 const int n_xcd, const int dbg )
{
  __builtin_amdgcn_s_setprio( 3 );      // reconstruction shares the SIMDs with seconds-long entropy-decode waves: win the issue arbitration

  __shared__ LfStripLds S;
  __shared__ int s_ticket;
  loopfilter_rows4_body( list, n_groups, mbh_max, mbw_max, ws, bnd, n_xcd, S, s_ticket, dbg );
}

// Which XCDs do workgroups of this device land on?  out[x] = number of workgroups of the launch that ran on XCD x.
__global__ void k_probe_xcds( int * out )
{
  if ( threadIdx.x == 0 ) atomicAdd( &out[xcc_id() & 15], 1 );
}

// How many of the context's HIP streams really run side by side?  One single-thread kernel per stream: each counts itself in and
// waits until all n have arrived or `timeout` (100 MHz ticks) has passed, then says how many it saw.  Streams that share a
// hardware queue run their kernels one after the other, so the kernels of the first round see only as many arrivals as there
// are queues: the smallest figure is the concurrency (GPU_MAX_HW_QUEUES: 4 by default, this runtime wants 16).
__global__ void k_probe_concurrency( uint32_t * counter, uint32_t * seen, uint32_t n, unsigned long long timeout, int slot )
{
  if ( threadIdx.x || blockIdx.x ) return;
  __hip_atomic_fetch_add( counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
  const unsigned long long t0 = wall_clock64();
  uint32_t m;
  while ( ( m = __hip_atomic_load( counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) ) < n && wall_clock64() - t0 < timeout ) __builtin_amdgcn_s_sleep( 32 );
  seen[slot] = m;
}

// After a row-pipelined launch: every queue must have handed out all of its rows.  A queue whose XCD received no workgroup
// (placement is not promised by HIP) would otherwise leave its units silently unprocessed.
__global__ void k_check_tickets( aa_sync_ws * ws, const int n_groups, const int mbh_max, const int n_xcd )
{
  if ( threadIdx.x != 0 || blockIdx.x != 0 ) return;
  for ( int x = 0; x < n_xcd; x++ ) {
    const int mine = x < n_groups ? ( n_groups - x + n_xcd - 1 ) / n_xcd : 0;
    const int t = __hip_atomic_load( &ws->ticket[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
    if ( t < mine * mbh_max && atomicCAS( &ws->error, 0, 5 ) == 0 ) { ws->where[0] = x; ws->where[1] = t; ws->where[2] = mine * mbh_max; }
  }
}

// Raster planes of n frame jobs, decided on the host when the frames are handed to reconstruction: out, last, golden, altref.
__global__ void k_bind_rasters( const aa_raster_binding * b, int n )
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= n ) return;
  aa_dev_frame * f = b[i].job;
  for ( int p = 0; p < 3; p++ ) f->cur[p] = b[i].cur[p];
  for ( int r = 1; r < 4; r++ ) for ( int p = 0; p < 3; p++ ) f->ref[r][p] = b[i].ref[r - 1][p];
}

// Delivery of shown frames (aa_download_batch_async): raster i (three padded planes, contiguous in its pool piece) -> piece i of one
// staging area, so that ONE copy takes a whole frame index of a batch to the host instead of three per stream.  blockIdx.y =
// raster, 16 bytes per thread and iteration, whole-line accesses.
__global__ __launch_bounds__( 256 ) void k_gather_rasters( const aa_gather_job * jobs, uint8_t * staging, size_t stride )
{
  const aa_gather_job J = jobs[blockIdx.y];
  const uint4 * src = reinterpret_cast<const uint4 *>( J.src );
  uint4 * dst = reinterpret_cast<uint4 *>( staging + stride * blockIdx.y );
  const size_t n16 = J.bytes >> 4;
  for ( size_t i = size_t( blockIdx.x ) * 256 + threadIdx.x; i < n16; i += size_t( gridDim.x ) * 256 ) dst[i] = src[i];
  if ( blockIdx.x == 0 && threadIdx.x < ( J.bytes & 15 ) ) staging[stride * blockIdx.y + ( n16 << 4 ) + threadIdx.x] = J.src[( n16 << 4 ) + threadIdx.x];
}

} // namespace

// ---- SSIM windows of the encoder's loop-filter search (util/ssim.cc:57-71 -> libx264 pixel_ssim_wxh) -------
// One thread per 8x8 window (windows step by 4): the sums over its 64 pixel pairs, then x264's ssim_end1 in single precision
// -- integer arithmetic up to the four factors, two multiplications and one correctly rounded division, nothing a contraction
// could fuse -- so the value is the one the plain-C algorithm produces (oracle/ssim_x264.c).  The host adds the windows up in
// x264's order (floats, four at a time, row by row): the order is part of the result.
__global__ __launch_bounds__( 256 ) void k_ssim_windows( const uint8_t * a, const uint8_t * b, int width, int w4, int h4, float * out )
{
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;       // window (x, y): pixels [4x, 4x+8) x [4y, 4y+8)
  if ( x >= w4 - 1 || y >= h4 - 1 ) return;
  uint32_t s1 = 0, s2 = 0, ss = 0, s12 = 0;
  for ( int r = 0; r < 8; r++ ) {
    const size_t off = static_cast<size_t>( 4 * y + r ) * width + 4 * x;
    const uint32_t * ra = reinterpret_cast<const uint32_t *>( a + off ), * rb = reinterpret_cast<const uint32_t *>( b + off );
    const uint32_t wa[2] = { ra[0], ra[1] }, wb[2] = { rb[0], rb[1] };
    for ( int k = 0; k < 8; k++ ) {
      const uint32_t p = ( wa[k >> 2] >> ( 8 * ( k & 3 ) ) ) & 255u, q = ( wb[k >> 2] >> ( 8 * ( k & 3 ) ) ) & 255u;
      s1 += p; s2 += q; ss += p * p + q * q; s12 += p * q;
    }
  }
  const int i1 = static_cast<int>( s1 ), i2 = static_cast<int>( s2 ), iss = static_cast<int>( ss ), i12 = static_cast<int>( s12 );
  const int c1 = 416, c2 = 235963;               // (int)(.01*.01*255*255*64 + .5), (int)(.03*.03*255*255*64*63 + .5)
  const int vars = iss * 64 - i1 * i1 - i2 * i2, covar = i12 * 64 - i1 * i2;
  const float num = __fmul_rn( static_cast<float>( 2 * i1 * i2 + c1 ), static_cast<float>( 2 * covar + c2 ) );
  const float den = __fmul_rn( static_cast<float>( i1 * i1 + i2 * i2 + c1 ), static_cast<float>( vars + c2 ) );
  out[static_cast<size_t>( y ) * ( w4 - 1 ) + x] = __fdiv_rn( num, den );
}

// Macroblock records of a loop-filter candidate: the frame's records with the per-macroblock level replaced by the candidate's
// (by segment; zero mode / reference adjustments: encoder.cc:464-470, loopfilter.cc:59-79).  levels = four bytes, one per segment.
__global__ __launch_bounds__( 256 ) void k_lf_relevel( const aa_mb_info * src, aa_mb_info * dst, unsigned nmb, uint32_t levels )
{
  const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
  if ( i >= nmb ) return;
  const uint4 * s = reinterpret_cast<const uint4 *>( src + i );
  uint4 * d = reinterpret_cast<uint4 *>( dst + i );
  uint4 head = s[0];                                              // y_mode uv_mode ref_frame segment_id | flags lf_level ...
  const uint32_t seg = ( head.x >> 24 ) & 3u;
  head.y = ( head.y & 0xFFFF00FFu ) | ( ( ( levels >> ( 8 * seg ) ) & 255u ) << 8 );
  d[0] = head;
  for ( int k = 1; k < 5; k++ ) d[k] = s[k];
}
static_assert( sizeof( aa_mb_info ) == 80, "k_lf_relevel copies five 16-byte pieces" );

int launch_lf_relevel( const aa_mb_info * src, aa_mb_info * dst, unsigned nmb, uint32_t levels, void * stream )
{
  hipLaunchKernelGGL( k_lf_relevel, dim3( ( nmb + 255 ) / 256 ), dim3( 256 ), 0, static_cast<hipStream_t>( stream ), src, dst, nmb, levels );
  return static_cast<int>( hipGetLastError() );
}

int launch_ssim_windows( const uint8_t * a, const uint8_t * b, int width, int height, float * out, void * stream )
{
  const int w4 = width >> 2, h4 = height >> 2;
  if ( w4 < 2 || h4 < 2 ) return static_cast<int>( hipErrorInvalidValue );
  hipLaunchKernelGGL( k_ssim_windows, dim3( ( w4 - 1 + 255 ) / 256, h4 - 1 ), dim3( 256 ), 0, static_cast<hipStream_t>( stream ), a, b, width, w4, h4, out );
  return static_cast<int>( hipGetLastError() );
}

int launch_gather_rasters( const aa_gather_job * jobs, int n, uint8_t * staging, size_t stride, size_t max_bytes, void * stream )
{
  const unsigned per = static_cast<unsigned>( std::min<size_t>( 64, std::max<size_t>( 1, max_bytes / ( 256 * 16 * 4 ) ) ) );
  for ( int base = 0; base < n; base += 32768 ) {
    const int cnt = std::min( 32768, n - base );
    hipLaunchKernelGGL( k_gather_rasters, dim3( per, cnt ), dim3( 256 ), 0, static_cast<hipStream_t>( stream ), jobs + base, staging + stride * base, stride );
    if ( hipError_t e = hipGetLastError() ) return static_cast<int>( e );
  }
  return 0;
}

int launch_bind_rasters( const aa_raster_binding * b, int n, void * stream )
{
  hipLaunchKernelGGL( k_bind_rasters, dim3( ( n + 63 ) / 64 ), dim3( 64 ), 0, static_cast<hipStream_t>( stream ), b, n );
  return static_cast<int>( hipGetLastError() );
}

int launch_recon_inter( const aa_frame_list & list, int n, unsigned max_mbs, bool split_only, void * stream )
{
  const unsigned blocks = ( ( max_mbs + 7u ) >> 3 ) * 8u;
  hipLaunchKernelGGL( k_recon_inter, dim3( blocks, n ), dim3( kLanes ), 0, static_cast<hipStream_t>( stream ), list, max_mbs, split_only ? 1 : 0 );
  return static_cast<int>( hipGetLastError() );
}
int launch_recon_inter4( const aa_frame_list & list, int n, unsigned max_mbs, void * stream )
{
  const unsigned quads = ( max_mbs + 3u ) >> 2;
  const unsigned blocks = ( ( quads + 7u ) >> 3 ) * 8u;
  hipLaunchKernelGGL( k_recon_inter4, dim3( blocks, n ), dim3( kLanes ), 0, static_cast<hipStream_t>( stream ), list, quads );
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
// Measurement hook (tools/row_kernel_probe.py): ALFALFA_AMD_LF_DEBUG=<bits> switches parts of the loop-filter row kernel
// OFF (1 stores, 2 own-row loads, 4 filter arithmetic, 8 loads of the rows above, 16 waiting).  Output is then invalid.
// Bit 5 and bits 8-20: fault injection and the waits' bounds for the test of the expired-wait path (lf_fault_injected).
static int lf_debug_bits()
{
  static const int bits = [] { const char * e = std::getenv( "ALFALFA_AMD_LF_DEBUG" ); return e ? std::atoi( e ) : 0; }();
  return bits;
}
// Test hook: ALFALFA_AMD_TEST_LDS_PAD=<bytes> adds dynamic LDS to the row-pipelined launches to force PARTIAL residency
// (the ordering protocol must not depend on every workgroup being resident).
static unsigned test_lds_pad()
{
  static const unsigned pad = [] { const char * e = std::getenv( "ALFALFA_AMD_TEST_LDS_PAD" ); return e ? static_cast<unsigned>( std::atoi( e ) ) : 0u; }();
  return pad;
}
int launch_loopfilter_rows4( const aa_frame_list & list, int n_groups, int mbh_max, int mbw_max, aa_sync_ws * ws, uint8_t * boundary, int n_xcd, void * stream )
{
  hipLaunchKernelGGL( k_loopfilter_rows4, dim3( n_xcd * ( ( n_groups + n_xcd - 1 ) / n_xcd ) * mbh_max ), dim3( kLanes ), test_lds_pad(), static_cast<hipStream_t>( stream ), list, n_groups, mbh_max, mbw_max, ws, boundary, n_xcd, lf_debug_bits() );
  hipLaunchKernelGGL( k_check_tickets, dim3( 1 ), dim3( 64 ), 0, static_cast<hipStream_t>( stream ), ws, n_groups, mbh_max, n_xcd );
  return static_cast<int>( hipGetLastError() );
}
int launch_recon_intra4( const aa_frame_list & list, int n_groups, int mbh_max, aa_sync_ws * ws, int n_xcd, void * stream )
{
  hipLaunchKernelGGL( k_recon_intra4, dim3( n_xcd * ( ( n_groups + n_xcd - 1 ) / n_xcd ) * mbh_max ), dim3( kLanes ), test_lds_pad(), static_cast<hipStream_t>( stream ), list, n_groups, mbh_max, ws, n_xcd );
  hipLaunchKernelGGL( k_check_tickets, dim3( 1 ), dim3( 64 ), 0, static_cast<hipStream_t>( stream ), ws, n_groups, mbh_max, n_xcd );
  return static_cast<int>( hipGetLastError() );
}
int launch_probe_concurrency( uint32_t * counter, uint32_t * seen, uint32_t n, unsigned long long timeout_ticks, int slot, void * stream )
{
  hipLaunchKernelGGL( k_probe_concurrency, dim3( 1 ), dim3( 64 ), 0, static_cast<hipStream_t>( stream ), counter, seen, n, timeout_ticks, slot );
  return static_cast<int>( hipGetLastError() );
}
int launch_probe_xcds( int * out16, int blocks, void * stream )
{
  hipLaunchKernelGGL( k_probe_xcds, dim3( blocks ), dim3( kLanes ), 0, static_cast<hipStream_t>( stream ), out16 );
  return static_cast<int>( hipGetLastError() );
}

} // namespace aa
