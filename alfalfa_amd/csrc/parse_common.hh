// Macroblock-header parse shared by the host parser (parser.cpp) and the device parser (parse_kernels.hip).
//
// One implementation of what the reference does in the Macroblock constructor and decode_prediction_modes
// (macroblock.cc:43-111,342-456), the motion-vector census (scorer.hh, macroblock.cc:143-195,301-312), MotionVector
// reading (macroblock.cc:198-229,283-287), split-MV sub-block prediction (macroblock.cc:231-281) and the per-macroblock
// loop-filter level (frame.cc:144-166, macroblock.cc:611-623, loopfilter.cc:59-79), templated on the boolean decoder so
// that the same statements run on a host core (64-bit window, bool_reader.hh) and in a GPU lane (32-bit window below).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "../../include/alfalfa_amd.h"
#include "vp8_tables.h"

#if defined( __HIPCC__ )
#define AA_HD __host__ __device__
#else
#define AA_HD
#endif

namespace aa {

enum MbMode : uint8_t { DC_PRED, V_PRED, H_PRED, TM_PRED, B_PRED, NEARESTMV, NEARMV, ZEROMV, NEWMV, SPLITMV };
enum BMode : uint8_t { B_DC_PRED, B_TM_PRED, B_VE_PRED, B_HE_PRED, B_LD_PRED, B_RD_PRED, B_VR_PRED, B_VL_PRED,
                       B_HD_PRED, B_HU_PRED, LEFT4X4, ABOVE4X4, ZERO4X4, NEW4X4 };
enum RefFrame : uint8_t { CURRENT_FRAME, LAST_FRAME, GOLDEN_FRAME, ALTREF_FRAME };
enum BlockType { Y_AFTER_Y2 = 0, Y2 = 1, UV = 2, Y_WITHOUT_Y2 = 3 };   // block.hh:46

// RFC 6386 trees as in modemv_data.cc:162-281 (leaves stored as -value, inner nodes as positive even indices)
constexpr int8_t kKfYModeTree[8] = { -B_PRED, 2, 4, 6, -DC_PRED, -V_PRED, -H_PRED, -TM_PRED };
constexpr int8_t kYModeTree[8] = { -DC_PRED, 2, 4, 6, -V_PRED, -H_PRED, -TM_PRED, -B_PRED };
constexpr int8_t kUvModeTree[6] = { -DC_PRED, 2, -V_PRED, 4, -H_PRED, -TM_PRED };
constexpr int8_t kBModeTree[18] = { -B_DC_PRED, 2, -B_TM_PRED, 4, -B_VE_PRED, 6, 8, 12, -B_HE_PRED, 10,
                                    -B_RD_PRED, -B_VR_PRED, -B_LD_PRED, 14, -B_VL_PRED, 16, -B_HD_PRED, -B_HU_PRED };
constexpr int8_t kSmallMvTree[14] = { 2, 8, 4, 6, -0, -1, -2, -3, 10, 12, -4, -5, -6, -7 };
constexpr int8_t kMvRefTree[8] = { -ZEROMV, 2, -NEARESTMV, 4, -NEARMV, 6, -NEWMV, -SPLITMV };
constexpr int8_t kSubMvRefTree[6] = { -LEFT4X4, 2, -ABOVE4X4, 4, -ZERO4X4, -NEW4X4 };
constexpr int8_t kSplitMvTree[6] = { -3, 2, -2, 4, -0, -1 };
constexpr int8_t kSegmentIdTree[6] = { 2, 4, -0, -1, -2, -3 };

constexpr uint8_t kZigzag[16] = { 0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15 };
constexpr uint8_t kBand[17] = { 0, 1, 2, 3, 6, 4, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7, 0 };

// mv_partitions (modemv_data.cc:245-276): partition index of each raster-order sub-block
constexpr uint8_t kSplitLayout[4][16] = { { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 1, 1, 1, 1 },
                                          { 0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 1, 1, 0, 0, 1, 1 },
                                          { 0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3 },
                                          { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 } };
constexpr uint8_t kSplitFirst[4][16] = { { 0, 8 }, { 0, 2 }, { 0, 2, 8, 10 }, { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 } };
constexpr uint8_t kSplitCount[4] = { 2, 2, 4, 16 };

// Everything the macroblock loop of one frame needs that is decided by the frame header and the persistent DecoderState
// (decoder_state.hh:72-167).  Written by the host header pre-pass (Parser::parse_header); read by the host macroblock loop
// or, unchanged, by the device parse kernels (one record per frame in HBM).  Plain data, no pointers.
//   CoeffProbs   the token probabilities: what the token lanes read (first: 16-byte aligned in a ParseJob)
//   HeaderParams everything else: what the macroblock-HEADER parse reads -- 200 bytes, which the header kernel copies into LDS per
//                lane (a field read from the job in HBM is a dependent global load in front of nearly every bool)
struct CoeffProbs { uint8_t coeff_probs[4][8][3][11]; };
struct HeaderParams {
  uint32_t first_off, first_size;        // first partition: byte range inside the frame (uncompressed_chunk.cc:117-130)
  uint32_t bd_bitpos;                    // boolean decoder hand-over at the first macroblock header (see BoolState)
  uint8_t bd_range, bd_active;
  uint8_t key, nparts;
  uint32_t part_off[8], part_size[8];    // DCT partitions (uncompressed_chunk.cc:132-155)
  uint16_t mbw, mbh;
  uint8_t seg_enabled, seg_update_map, seg_tree_probs[3];
  uint8_t skip_enabled, prob_skip, prob_inter, prob_last, prob_golden;
  uint8_t sign_bias_golden, sign_bias_alt;
  uint8_t loop_filter_level, fadj_enabled;
  int8_t fadj_ref[4], fadj_mode[4];
  int16_t seg_level[4];                  // loop-filter level per segment before the per-macroblock adjustments (Q3: unclamped)
  uint8_t y_mode_probs[4], uv_mode_probs[3], mv_probs[2][19];
  uint8_t pad[3];
};
struct FrameParams : CoeffProbs, HeaderParams {};
static_assert( sizeof( FrameParams ) % 4 == 0 && sizeof( HeaderParams ) % 4 == 0 && sizeof( FrameParams ) == sizeof( CoeffProbs ) + sizeof( HeaderParams ),
               "FrameParams / HeaderParams are copied as words" );

// Every constant table the macroblock-header parse reads with a data-dependent index, as ONE blob: the header kernel copies it into
// LDS once per workgroup (from the kernel's constant data each tree node and each probability was a dependent global load: two of
// them in front of every bool); the host parser reads the static instance.
struct HeaderTables {
  int8_t kf_y_mode_tree[8], y_mode_tree[8], uv_mode_tree[6], b_mode_tree[18], small_mv_tree[14], mv_ref_tree[8], sub_mv_ref_tree[6], split_mv_tree[6], segment_id_tree[6];
  uint8_t kf_y_mode_probs[4], kf_uv_mode_probs[3], b_mode_probs[9], mv_counts_to_probs[24], split_mv_probs[3], submv_ref_probs[15];
  uint8_t split_layout[4][16], split_first[4][16], split_count[4];
  uint8_t kf_b_mode_probs[900];          // [above mode][left mode][node]
  uint8_t pad[2];
};
static_assert( sizeof( HeaderTables ) % 4 == 0, "HeaderTables is copied as words" );
constexpr HeaderTables make_header_tables()
{
  HeaderTables t {};
  for ( int i = 0; i < 8; i++ ) { t.kf_y_mode_tree[i] = kKfYModeTree[i]; t.y_mode_tree[i] = kYModeTree[i]; t.mv_ref_tree[i] = kMvRefTree[i]; }
  for ( int i = 0; i < 6; i++ ) { t.uv_mode_tree[i] = kUvModeTree[i]; t.sub_mv_ref_tree[i] = kSubMvRefTree[i]; t.split_mv_tree[i] = kSplitMvTree[i]; t.segment_id_tree[i] = kSegmentIdTree[i]; }
  for ( int i = 0; i < 18; i++ ) t.b_mode_tree[i] = kBModeTree[i];
  for ( int i = 0; i < 14; i++ ) t.small_mv_tree[i] = kSmallMvTree[i];
  for ( int i = 0; i < 4; i++ ) t.kf_y_mode_probs[i] = k_kf_y_mode_probs[i];
  for ( int i = 0; i < 3; i++ ) { t.kf_uv_mode_probs[i] = k_kf_uv_mode_probs[i]; t.split_mv_probs[i] = k_split_mv_probs[i]; }
  for ( int i = 0; i < 9; i++ ) t.b_mode_probs[i] = k_b_mode_probs[i];
  for ( int i = 0; i < 24; i++ ) t.mv_counts_to_probs[i] = k_mv_counts_to_probs[i];
  for ( int i = 0; i < 15; i++ ) t.submv_ref_probs[i] = k_submv_ref_probs[i];
  for ( int i = 0; i < 4; i++ ) { t.split_count[i] = kSplitCount[i]; for ( int j = 0; j < 16; j++ ) { t.split_layout[i][j] = kSplitLayout[i][j]; t.split_first[i][j] = kSplitFirst[i][j]; } }
  for ( int i = 0; i < 900; i++ ) t.kf_b_mode_probs[i] = k_kf_b_mode_probs[i];
  return t;
}
constexpr HeaderTables kHeaderTables = make_header_tables();



// Arithmetic-decoder state in a window-independent form.  The window of a VP8 boolean decoder is "8 active bits that
// have been through the subtractions" followed by raw stream bits that nothing has touched yet (a subtraction of
// split << (W-8) never borrows from below).  So (range, active byte, index of the first raw bit not yet shifted into the
// active byte) says everything, whatever the window width of the reader that continues.
struct BoolState { uint32_t bitpos; uint8_t range, active; };

// Boolean decoder with a 32-bit window over a byte range; same arithmetic as the reference's BoolDecoder
// (bool_decoder.hh:45-120).  Used by the GPU lanes of the macroblock-header kernel (and, on the host, by the tests that
// replay the device algorithm).  Bytes past the end read as zero.
class BoolReader32
{
  const uint8_t * base_ = nullptr;
  uint32_t pos_ = 0, end_ = 0;   // next byte to append, end of the partition (offsets from base_)
  uint32_t value_ = 0;           // window, most significant bits first
  int count_ = 0;                // valid bits below the 8 being compared
  uint32_t range_ = 255;
  uint32_t cache_ = 0;           // the bytes at pos_, pos_ + 1, ... of the aligned word they were read with (lowest byte first)
  uint32_t cache_n_ = 0;         // how many of them: the stream is read a word at a time -- on the GPU every read is a trip to the L2

  // the byte at pos_ (zero past the end), pos_ advanced.  The word read is the ALIGNED one the byte lies in: it never reaches into a
  // page the byte itself is not on, whatever the alignment of the partition and wherever the buffer ends.  Memory is touched only
  // while pos_ is INSIDE the partition: a truncated or crafted frame keeps the decoder consuming zeros for every macroblock header
  // that follows (bool_decoder.hh:56-65), and a word read per four of those would walk hundreds of KB past the partition -- off the
  // end of the pinned arena on the host-lane path.
  AA_HD inline uint32_t next_byte()
  {
    if ( cache_n_ == 0 ) {
      if ( pos_ < end_ ) {
        const uintptr_t a = reinterpret_cast<uintptr_t>( base_ + pos_ );
        const uint32_t skip = static_cast<uint32_t>( a & 3u );
        cache_ = *reinterpret_cast<const uint32_t *>( a - skip ) >> ( 8u * skip );
        cache_n_ = 4u - skip;
      } else { cache_ = 0; cache_n_ = 4u; }
    }
    const uint32_t byte = pos_ < end_ ? ( cache_ & 0xFFu ) : 0u;
    cache_ >>= 8; cache_n_--; pos_++;
    return byte;
  }

public:
  AA_HD BoolReader32() {}
  AA_HD void resume( const uint8_t * base, uint32_t size, const BoolState & st )
  {
    base_ = base; end_ = size; range_ = st.range;
    // raw bits from st.bitpos up to the next byte boundary + 16 go below the active byte; from there on whole bytes
    const uint32_t byte = st.bitpos >> 3, r = st.bitpos & 7;
    pos_ = byte; cache_n_ = 0;
    uint32_t b = 0;
    for ( int k = 0; k < 3; k++ ) b = ( b << 8 ) | next_byte();
    value_ = ( static_cast<uint32_t>( st.active ) << 24 ) | ( ( b << r ) & 0xFFFFFFu );
    count_ = 24 - static_cast<int>( r );
  }
  AA_HD void reset( const uint8_t * base, uint32_t size )     // fresh partition: the first byte is the active byte
  {
    BoolState st; st.bitpos = 8; st.range = 255; st.active = size ? base[0] : 0;
    resume( base, size, st );
  }
  AA_HD inline int get( const uint32_t prob )
  {
    const uint32_t split = 1 + ( ( ( range_ - 1 ) * prob ) >> 8 );
    if ( count_ < 0 ) {                                         // one byte always fits: 8 + count_ < 8
      value_ |= next_byte() << ( 16 - count_ );
      count_ += 8;
    }
    const uint32_t bigsplit = split << 24;
    int bit;
    uint32_t range;
    if ( value_ >= bigsplit ) { range = range_ - split; value_ -= bigsplit; bit = 1; }
    else { range = split; bit = 0; }
    const int shift = __builtin_clz( range ) - 24;
    range_ = range << shift;
    value_ <<= shift;
    count_ -= shift;
    return bit;
  }
  AA_HD inline int flag() { return get( 128 ); }
  AA_HD inline int literal( int bits ) { int v = 0; while ( bits-- ) v = ( v << 1 ) | get( 128 ); return v; }
  AA_HD inline int tree( const int8_t * nodes, const uint8_t * probs )
  {
    int i = 0;
    while ( ( i = nodes[i + get( probs[i >> 1] )] ) > 0 ) {}
    return -i;
  }
};

struct Mv {
  int16_t x = 0, y = 0;
  AA_HD bool zero() const { return x == 0 && y == 0; }
  AA_HD bool operator==( const Mv & o ) const { return x == o.x && y == o.y; }
};

AA_HD inline int clamp_int( int v, int lo, int hi ) { return v < lo ? lo : ( v > hi ? hi : v ); }

// MotionVector::read_component, macroblock.cc:198-229
template <class BD>
AA_HD inline int16_t read_mv_component( BD & bd, const uint8_t * p, const HeaderTables & T )
{
  enum { MV_IS_SHORT, SIGN, SHORT, BITS = SHORT + 8 - 1, MV_LONG_BITS = 10 };
  int x = 0;
  if ( bd.get( p[MV_IS_SHORT] ) ) {
    for ( int i = 0; i < 3; i++ ) x += bd.get( p[BITS + i] ) << i;
    for ( int i = MV_LONG_BITS - 1; i > 3; i-- ) x += bd.get( p[BITS + i] ) << i;
    if ( !( x & 0xFFF0 ) || bd.get( p[BITS + 3] ) ) x += 8;
  } else {
    x = bd.tree( T.small_mv_tree, p + SHORT );
  }
  x <<= 1;
  if ( x && bd.get( p[SIGN] ) ) x = -x;
  return static_cast<int16_t>( x );
}

template <class BD>
AA_HD inline Mv read_mv( BD & bd, const HeaderParams & fp, const HeaderTables & T )
{
  Mv m;
  m.y = read_mv_component( bd, fp.mv_probs[0], T );   // row first (macroblock.cc:283-287)
  m.x = read_mv_component( bd, fp.mv_probs[1], T );
  return m;
}

// Scorer::clamp, macroblock.cc:183-195
AA_HD inline Mv clamp_mv( Mv m, unsigned col, unsigned row, unsigned mbw, unsigned mbh )
{
  const int to_left = clamp_int( -( static_cast<int>( col * 16 ) << 3 ) - 128, -32768, 32767 );
  const int to_right = clamp_int( ( static_cast<int>( ( mbw - 1 - col ) * 16 ) << 3 ) + 128, -32768, 32767 );
  const int to_top = clamp_int( -( static_cast<int>( row * 16 ) << 3 ) - 128, -32768, 32767 );
  const int to_bottom = clamp_int( ( static_cast<int>( ( mbh - 1 - row ) * 16 ) << 3 ) + 128, -32768, 32767 );
  m.x = static_cast<int16_t>( clamp_int( m.x, to_left, to_right ) );
  m.y = static_cast<int16_t>( clamp_int( m.y, to_top, to_bottom ) );
  return m;
}

// Final loop-filter level of one macroblock: frame.cc:144-166, macroblock.cc:611-623, loopfilter.cc:59-79
AA_HD inline uint8_t mb_lf_level( const HeaderParams & fp, unsigned segment_id, unsigned ref_frame, unsigned y_mode )
{
  if ( !fp.loop_filter_level ) return 0;
  int level = fp.seg_level[segment_id];
  if ( fp.fadj_enabled ) {
    level += fp.fadj_ref[ref_frame];
    if ( ref_frame == CURRENT_FRAME ) level += ( y_mode == B_PRED ) ? fp.fadj_mode[0] : 0;
    else if ( y_mode == ZEROMV ) level += fp.fadj_mode[1];
    else if ( y_mode == SPLITMV ) level += fp.fadj_mode[3];
    else level += fp.fadj_mode[2];
  }
  return static_cast<uint8_t>( level <= 0 ? 0 : ( level > 63 ? 63 : level ) );
}

// One macroblock of the segment-map pass that follows a device parse (frames of a stream in order): a frame that updates
// the map publishes its ids, a frame that does not inherits them (and only now learns its loop-filter levels).
AA_HD inline void segment_fixup( const HeaderParams & fp, aa_mb_info & mb, uint8_t & map )
{
  if ( fp.seg_update_map ) map = mb.segment_id;
  else {
    mb.segment_id = map;
    mb.lf_level = mb_lf_level( fp, map, mb.ref_frame, mb.y_mode );
  }
}

AA_HD inline bool mv_flipped( const HeaderParams & fp, unsigned ref_frame )    // motion_vectors_flipped_, macroblock.cc:464-465
{
  return ( ref_frame == GOLDEN_FRAME && fp.sign_bias_golden ) || ( ref_frame == ALTREF_FRAME && fp.sign_bias_alt );
}

// Header of macroblock (col,row): segment id, skip flag, reference frame, prediction modes, motion vectors.  Writes the
// whole record mbs[mi] except nz_mask / coeff_index / the HAS_NONZERO and LF_SKIP_INNER flags, which belong to the token
// parse.  `segmap`: the persistent segment map (mb_width*mb_height), or null when the caller resolves inherited segment ids
// afterwards (device parser: frames of one stream are parsed concurrently; see k_segment_fixup).
// Returns the record's flags (INTER | HAS_Y2 | SKIP).
template <class BD>
AA_HD inline uint8_t parse_mb_header( BD & bd, const HeaderParams & fp, const HeaderTables & T, aa_mb_info * mbs, unsigned mi, unsigned col, unsigned row,
                                      uint8_t * segmap )
{
  const unsigned mbw = fp.mbw, mbh = fp.mbh;
  const bool key = fp.key;
  aa_mb_info & mb = mbs[mi];
  {
    uint32_t * w = reinterpret_cast<uint32_t *>( &mb );
    for ( unsigned k = 0; k < sizeof( aa_mb_info ) / 4; k++ ) w[k] = 0;
  }

  // Macroblock ctor: segment id, skip flag, inter/intra + reference (macroblock.cc:43-71, 458-465)
  unsigned segment_id = 0;
  if ( fp.seg_enabled ) {
    if ( fp.seg_update_map ) {
      segment_id = static_cast<unsigned>( bd.tree( T.segment_id_tree, fp.seg_tree_probs ) );
      if ( segmap ) segmap[mi] = static_cast<uint8_t>( segment_id );
    } else if ( segmap ) segment_id = segmap[mi];
  }
  mb.segment_id = static_cast<uint8_t>( segment_id );
  const bool skip = fp.skip_enabled ? bd.get( fp.prob_skip ) : false;
  bool inter = false;
  if ( !key ) {
    inter = bd.get( fp.prob_inter );
    if ( inter ) {
      mb.ref_frame = LAST_FRAME;
      if ( bd.get( fp.prob_last ) ) mb.ref_frame = bd.get( fp.prob_golden ) ? ALTREF_FRAME : GOLDEN_FRAME;
    }
  }

  if ( !inter ) {
    // ---- intra modes: macroblock.cc:84-111 (key) / 354-376 (inter frame) ----
    mb.y_mode = static_cast<uint8_t>( key ? bd.tree( T.kf_y_mode_tree, T.kf_y_mode_probs ) : bd.tree( T.y_mode_tree, fp.y_mode_probs ) );
    if ( mb.y_mode == B_PRED ) {
      // (the sixteen modes in registers until all are known: read back out of the record they were a store -> load round trip
      // through the L2 per sub-block on the GPU)
      uint8_t bm[16];
      uint8_t above4[4] = { B_DC_PRED, B_DC_PRED, B_DC_PRED, B_DC_PRED }, left4[4] = { B_DC_PRED, B_DC_PRED, B_DC_PRED, B_DC_PRED };
      if ( key ) {
        if ( row > 0 ) for ( int k = 0; k < 4; k++ ) above4[k] = mbs[mi - mbw].u.b_mode[12 + k];
        if ( col > 0 ) for ( int k = 0; k < 4; k++ ) left4[k] = mbs[mi - 1].u.b_mode[4 * k + 3];
      }
#if defined( __HIP_DEVICE_COMPILE__ )
#pragma unroll
#endif
      for ( int b = 0; b < 16; b++ ) {
        if ( key ) {
          const int above_mode = b >= 4 ? bm[b - 4] : above4[b];
          const int left_mode = ( b & 3 ) ? bm[b - 1] : left4[b >> 2];
          bm[b] = static_cast<uint8_t>( bd.tree( T.b_mode_tree, T.kf_b_mode_probs + ( above_mode * 10 + left_mode ) * 9 ) );
        } else {
          bm[b] = static_cast<uint8_t>( bd.tree( T.b_mode_tree, T.b_mode_probs ) );
        }
      }
      for ( int b = 0; b < 16; b++ ) mb.u.b_mode[b] = bm[b];
    } else {
      constexpr uint8_t kImplied[4] = { B_DC_PRED, B_VE_PRED, B_HE_PRED, B_TM_PRED };   // macroblock.hh:134-143
      const uint8_t m = kImplied[mb.y_mode];
      for ( int b = 0; b < 16; b++ ) mb.u.b_mode[b] = m;
    }
    mb.uv_mode = static_cast<uint8_t>( key ? bd.tree( T.uv_mode_tree, T.kf_uv_mode_probs ) : bd.tree( T.uv_mode_tree, fp.uv_mode_probs ) );
  } else {
    // ---- inter modes: census (scorer.hh, macroblock.cc:143-181,301-312) then mode / MVs (:377-455) ----
    mb.flags |= AA_MB_INTER;
    uint8_t score[4] = { 0, 0, 0, 0 };
    Mv cand[4];
    int idx = 0, split_score = 0;
    const bool my_flip = mv_flipped( fp, mb.ref_frame );
    auto consider = [&]( unsigned ni, int weight ) {
      const aa_mb_info & nb = mbs[ni];
      if ( !( nb.flags & AA_MB_INTER ) ) return;
      Mv mv; mv.x = nb.u.mv[15][0]; mv.y = nb.u.mv[15][1];
      if ( mv_flipped( fp, nb.ref_frame ) != my_flip ) { mv.x = static_cast<int16_t>( -mv.x ); mv.y = static_cast<int16_t>( -mv.y ); }
      if ( mv.zero() ) score[0] += weight;
      else {
        if ( !( mv == cand[idx] ) ) cand[++idx] = mv;
        score[idx] += weight;
      }
      if ( nb.y_mode == SPLITMV ) split_score += weight;
    };
    if ( row > 0 ) consider( mi - mbw, 2 );
    if ( col > 0 ) consider( mi - 1, 2 );
    if ( row > 0 && col > 0 ) consider( mi - mbw - 1, 1 );
    if ( score[3] && cand[idx] == cand[1] ) score[1] += score[3];                       // Q8
    if ( score[2] > score[1] ) {
      const uint8_t ts = score[1]; score[1] = score[2]; score[2] = ts;
      const Mv tm = cand[1]; cand[1] = cand[2]; cand[2] = tm;
    }
    if ( score[1] >= score[0] ) cand[0] = cand[1];
    const uint8_t mode_probs[4] = { T.mv_counts_to_probs[score[0] * 4 + 0], T.mv_counts_to_probs[score[1] * 4 + 1],
                                    T.mv_counts_to_probs[score[2] * 4 + 2], T.mv_counts_to_probs[split_score * 4 + 3] };
    mb.y_mode = static_cast<uint8_t>( bd.tree( T.mv_ref_tree, mode_probs ) );
    Mv base;
    switch ( mb.y_mode ) {
    case NEARESTMV: base = clamp_mv( cand[1], col, row, mbw, mbh ); break;
    case NEARMV: base = clamp_mv( cand[2], col, row, mbw, mbh ); break;
    case ZEROMV: break;
    case NEWMV: {
      const Mv delta = read_mv( bd, fp, T );
      const Mv best = clamp_mv( cand[0], col, row, mbw, mbh );
      base.x = static_cast<int16_t>( delta.x + best.x ); base.y = static_cast<int16_t>( delta.y + best.y );
      break; }
    default: {   // SPLITMV (the tree has no other leaf)
      mb.split_partition = static_cast<uint8_t>( bd.tree( T.split_mv_tree, T.split_mv_probs ) );
      const Mv best = clamp_mv( cand[0], col, row, mbw, mbh );
      const uint8_t * layout = T.split_layout[mb.split_partition];
      for ( int part = 0; part < T.split_count[mb.split_partition]; part++ ) {
        const int b = T.split_first[mb.split_partition][part];
        // YBlock::read_subblock_inter_prediction, macroblock.cc:231-281
        Mv lmv, amv;
        if ( b & 3 ) { lmv.x = mb.u.mv[b - 1][0]; lmv.y = mb.u.mv[b - 1][1]; }
        else if ( col > 0 && ( mbs[mi - 1].flags & AA_MB_INTER ) ) { lmv.x = mbs[mi - 1].u.mv[b + 3][0]; lmv.y = mbs[mi - 1].u.mv[b + 3][1]; }
        if ( b >= 4 ) { amv.x = mb.u.mv[b - 4][0]; amv.y = mb.u.mv[b - 4][1]; }
        else if ( row > 0 && ( mbs[mi - mbw].flags & AA_MB_INTER ) ) { amv.x = mbs[mi - mbw].u.mv[b + 12][0]; amv.y = mbs[mi - mbw].u.mv[b + 12][1]; }
        int ctx = 0;
        if ( lmv == amv ) ctx = lmv.zero() ? 4 : 3;
        else if ( amv.zero() ) ctx = 2;
        else if ( lmv.zero() ) ctx = 1;
        Mv m;
        switch ( bd.tree( T.sub_mv_ref_tree, T.submv_ref_probs + ctx * 3 ) ) {
        case LEFT4X4: m = lmv; break;
        case ABOVE4X4: m = amv; break;
        case ZERO4X4: break;
        default: { const Mv d = read_mv( bd, fp, T ); m.x = static_cast<int16_t>( d.x + best.x ); m.y = static_cast<int16_t>( d.y + best.y ); break; }   // NEW4X4
        }
        for ( int k = 0; k < 16; k++ ) if ( layout[k] == part ) { mb.u.mv[k][0] = m.x; mb.u.mv[k][1] = m.y; }
      }
      break; }
    }
    if ( mb.y_mode != SPLITMV ) for ( int k = 0; k < 16; k++ ) { mb.u.mv[k][0] = base.x; mb.u.mv[k][1] = base.y; }
  }

  if ( !( mb.y_mode == B_PRED || mb.y_mode == SPLITMV ) ) mb.flags |= AA_MB_HAS_Y2;
  if ( skip ) mb.flags |= AA_MB_SKIP;
  mb.lf_level = mb_lf_level( fp, segment_id, mb.ref_frame, mb.y_mode );
  return mb.flags;
}

} // namespace aa
