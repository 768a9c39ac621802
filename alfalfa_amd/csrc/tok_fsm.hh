// Token parse (Block::parse_tokens tokens.cc:50-135, Macroblock::parse_tokens macroblock.cc:475-502, driven per row as in
// frame.cc:121-137) as a flat state machine: ONE boolean decode per step, every lane of a wave at its own place in its own
// frame.  A GPU lane cannot afford the natural loop nest (block loop inside macroblock loop inside row loop): lanes of a wave
// would wait for the longest block / macroblock of the 64 at every level.  Here a step is "read one probability, decode one
// bool, move to the next node"; block ends, macroblock ends, row ends and the end of the frame are transitions of the same
// machine, so lanes never wait for each other.
//
// The same code is the device kernel's body (parse_kernels.hip) and, compiled for the host, what tests/cpp/fsm_sim.cc replays
// lane by lane against the host parser -- a lane touches nothing but its own state, so one lane at a time is exact.
//
// Memory of a lane ("LDS" = the lane's slice of the workgroup's LDS on the GPU, a plain buffer on the host):
//   LDS   this frame's token probabilities + the fixed extra-bit probabilities (one byte read per step)
//         a 256-byte ring of the current partition's bytes and a 256-entry ring of macroblock header flags; both are
//         topped up from HBM every kPeriod steps, for all lanes at once, with loads issued one period ahead -- the step
//         itself never waits for HBM
//         the above-row non-zero flags (9 bits per macroblock column), the Y2 block under construction, saved decoder
//         states of the other DCT partitions
//   HBM   the frame's compressed bytes, flags[mi] from the header kernel (in), coefficient blocks + nz_mask / coeff_index /
//         flags of every macroblock record (out, fire-and-forget stores)
#pragma once
#include <initializer_list>

#include "parse_common.hh"

namespace aa {

struct alignas( 16 ) V16 { uint32_t x, y, z, w; };   // one 16-byte memory transaction

// On the GPU the pointers a lane follows come out of a job record in memory, so the compiler cannot tell which address space
// they point into and would emit FLAT accesses -- whose completion is counted on the LDS counter too, so that every LDS
// read of the step would wait for the coefficient stores.  They are HBM pointers: say so.
#if defined( __HIP_DEVICE_COMPILE__ )
#define AA_GLOBAL __attribute__( ( address_space( 1 ) ) )
#else
#define AA_GLOBAL
#endif

struct FrameSummary {           // written by the device parser, read by the host once the parse event has fired
  uint32_t num_coeff_blocks;
  uint32_t num_intra_mbs;
  uint32_t has_split;
  uint32_t steps;               // boolean decodes + macroblock events of the token lane (diagnostics)
};

// One frame to parse on the device.  Built by the host header pre-pass, resident in HBM.
struct alignas( 16 ) ParseJob {
  FrameParams fp;
  const uint8_t * data;         // compressed frame, 16-byte aligned, readable up to data_padded
  uint32_t size, data_padded;   // data_padded: multiple of 16, >= size
  uint32_t nmb, flags_padded;   // flags_padded: multiple of 16, >= nmb
  aa_mb_info * mbs;
  int16_t * coeffs;             // 25 * nmb + 1 blocks of 16
  unsigned long long * intra_rows;
  uint8_t * mbflags;            // [flags_padded]: INTER | HAS_Y2 | SKIP of every macroblock, header kernel -> token kernel
  FrameSummary * summary;
};

namespace tok {

constexpr uint32_t kPeriod = 64;          // steps between ring top-ups (a step consumes at most one stream byte / one flag)
constexpr uint32_t kRing = 256;           // bytes per ring
constexpr uint32_t kChunks = 4;           // 16-byte chunks fetched per top-up (= kPeriod bytes)

// lane LDS layout (byte offsets)
constexpr uint32_t kProbs = 0;            // [4][8][3][11] token probabilities
constexpr uint32_t kXtab = 1056;          // extra-bit probabilities of the six categories, then the sign's 128
constexpr uint32_t kSignP = kXtab + 26;
constexpr uint32_t kStream = 1088;        // stream ring
constexpr uint32_t kMeta = kStream + kRing;
constexpr uint32_t kY2 = kMeta + kRing;   // 16 x int16
constexpr uint32_t kPart = kY2 + 32;      // 8 saved partition decoders x 16 bytes
constexpr uint32_t kAbove = kPart + 128;  // uint16 per macroblock column
AA_HD constexpr uint32_t lane_lds_bytes( uint32_t mbw ) { return ( kAbove + 2 * mbw + 15 ) & ~15u; }

// dct_cat probabilities (tokens.cc:36-48) laid out back to back: cat1 @0, cat2 @1, cat3 @3, cat4 @6, cat5 @10, cat6 @15
constexpr uint8_t kXtabInit[27] = { 159, 165, 145, 173, 148, 140, 176, 155, 140, 135, 180, 157, 141, 134, 130,
                                    254, 254, 243, 230, 196, 177, 153, 140, 133, 130, 129, 128 };

enum : uint32_t { ST_SIGN = 11, ST_EXTRA = 12, ST_MB = 13, ST_DONE = 14 };
enum : uint32_t { NX_SIGN = 11, NX_EXTRA = 12, NX_ZERO = 13, NX_EOB = 14 };

// The token tree (tokens.cc:73-124) as two nibble tables indexed by node: where a 0 / a 1 leads, and the argument of a
// leaf (the magnitude for DCT_1..4, the category for dct_cat1..6).
constexpr uint64_t nib( std::initializer_list<unsigned> v ) { uint64_t r = 0; unsigned i = 0; for ( unsigned x : v ) r |= static_cast<uint64_t>( x ) << ( 4 * i++ ); return r; }
constexpr uint64_t tri( std::initializer_list<unsigned> v ) { uint64_t r = 0; unsigned i = 0; for ( unsigned x : v ) r |= static_cast<uint64_t>( x ) << ( 3 * i++ ); return r; }
//                                   node:    0        1        2        3  4        5        6  7         8  9         10
constexpr uint64_t kNext0 = nib( { NX_EOB, NX_ZERO, NX_SIGN, 4, NX_SIGN, NX_SIGN, 7, NX_EXTRA, 9, NX_EXTRA, NX_EXTRA } );
constexpr uint64_t kNext1 = nib( { 1,      2,       3,       6, 5,       NX_SIGN, 8, NX_EXTRA, 10, NX_EXTRA, NX_EXTRA } );
constexpr uint64_t kArg0 = nib( { 0, 0, 1, 0, 2, 3, 0, 0, 0, 2, 4 } );
constexpr uint64_t kArg1 = nib( { 0, 0, 0, 0, 0, 4, 0, 1, 0, 3, 5 } );
constexpr uint64_t kXLen = nib( { 1, 2, 3, 4, 5, 11 } );
constexpr uint64_t kXOff = nib( { 0, 1, 3, 6, 10, 15 } );
constexpr uint64_t kXBase = 5ull | ( 7ull << 8 ) | ( 11ull << 16 ) | ( 19ull << 24 ) | ( 35ull << 32 ) | ( 67ull << 40 );
constexpr uint64_t kZigzagNib = nib( { 0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15 } );
constexpr uint64_t kBandTri = tri( { 0, 1, 2, 3, 6, 4, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7, 0 } );   // coefficient band of position 0..16

AA_HD inline uint32_t band_of( uint32_t idx ) { return static_cast<uint32_t>( kBandTri >> ( idx * 3 ) ) & 7u; }

// What a lane needs of its ParseJob at every step, held in registers (the job itself stays in HBM and is only consulted on
// the rare paths: partition switches, the end of the frame).
struct Frame {
  const AA_GLOBAL ParseJob * job;
  const AA_GLOBAL uint8_t * data;
  const AA_GLOBAL uint8_t * mbflags;
  AA_GLOBAL aa_mb_info * mbs;
  AA_GLOBAL int16_t * coeffs;
  uint32_t data_padded, flags_padded, nmb, mbw, nparts;
  uint32_t max_steps;           // no frame of this size can take more steps: a lane that gets there stops (never a hung GPU)
};
AA_HD inline Frame frame_of( const ParseJob * job )
{
  Frame F;
  F.job = (const AA_GLOBAL ParseJob *) job;
  F.data = (const AA_GLOBAL uint8_t *) job->data; F.mbflags = (const AA_GLOBAL uint8_t *) job->mbflags;
  F.mbs = (AA_GLOBAL aa_mb_info *) job->mbs; F.coeffs = (AA_GLOBAL int16_t *) job->coeffs;
  // per macroblock at most 25 blocks x 16 tokens x (11 tree nodes + 11 extra bits + sign), plus one event step
  const uint64_t bound = static_cast<uint64_t>( job->nmb ) * ( 25u * 16u * 23u + 2u ) + 64u;
  F.max_steps = bound > 0xFFFFFFF0ull ? 0xFFFFFFF0u : static_cast<uint32_t>( bound );
  F.data_padded = job->data_padded; F.flags_padded = job->flags_padded; F.nmb = job->nmb; F.mbw = job->fp.mbw; F.nparts = job->fp.nparts;
  return F;
}

struct Chunk16 { uint32_t w[4]; };

AA_HD inline Chunk16 load16( const AA_GLOBAL uint8_t * p )          // 16-byte aligned
{
  const AA_GLOBAL V16 * q = (const AA_GLOBAL V16 *) p;
  const V16 v = *q;
  Chunk16 c; c.w[0] = v.x; c.w[1] = v.y; c.w[2] = v.z; c.w[3] = v.w;
  return c;
}
AA_HD inline void lds_store16( uint8_t * lds, uint32_t off, const Chunk16 & c )
{
  V16 v; v.x = c.w[0]; v.y = c.w[1]; v.z = c.w[2]; v.w = c.w[3];
  *reinterpret_cast<V16 *>( lds + off ) = v;
}

struct Lane {
  // boolean decoder of the current partition: 32-bit window, `count` valid bits below the 8 being compared
  uint32_t value, range;
  int32_t count;
  uint32_t rpos, rend;            // next stream byte to shift in / end of the partition (offsets into the frame)
  uint32_t wpos;                  // stream ring holds [wpos - kRing, wpos)
  uint32_t mwpos;                 // flag ring holds macroblocks [mwpos - kRing, mwpos)
  uint32_t pend_wpos, pend_mwpos; // what the chunks in flight are for (kNoPend: nothing in flight)
  Chunk16 pend[kChunks], mpend[kChunks];
  // position
  uint32_t mi, col, row, part;
  // macroblock in progress
  uint32_t flags, left_nz, above_nz, nz_mask, y2_nz, mb_first;
  uint32_t coeff_blocks;
  // block in progress
  uint32_t blk, idx, typeoff, rowoff, nonzero, ctx_next;
  uint32_t st, paddr;
  uint32_t mag, xrem, xbase;
  uint32_t steps;
};
constexpr uint32_t kNoPend = 0xFFFFFFFFu;

AA_HD inline void zero_slot( const Frame & J, uint32_t block )
{
  V16 z; z.x = z.y = z.z = z.w = 0;
  AA_GLOBAL V16 * p = (AA_GLOBAL V16 *) ( J.coeffs + static_cast<size_t>( block ) * 16 );
  p[0] = z; p[1] = z;
}

// ---- stream ring ----------------------------------------------------------------------------------------------------
AA_HD inline uint32_t ring_byte( const uint8_t * lds, uint32_t pos ) { return lds[kStream + ( pos & ( kRing - 1 ) )]; }

// synchronous (re)fill of the whole stream ring around rpos: start of a partition
AA_HD inline void prime_stream( Lane & L, uint8_t * lds, const Frame & J )
{
  const uint32_t base = L.rpos & ~15u;
  for ( uint32_t k = 0; k < kRing / 16; k++ ) {
    const uint32_t at = base + 16 * k;
    Chunk16 c; c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0;
    if ( at < J.data_padded ) c = load16( J.data + at );
    lds_store16( lds, kStream + ( at & ( kRing - 1 ) ), c );
  }
  L.wpos = base + kRing;
  L.pend_wpos = kNoPend;
}

// decoder of partition `p` at its first bit (BoolDecoder ctor, bool_decoder.hh:45-54): state words for the save area
AA_HD inline void start_partition( Lane & L, uint8_t * lds, const Frame & J, uint32_t p )
{
  L.part = p;
  L.rpos = J.job->fp.part_off[p];
  L.rend = J.job->fp.part_off[p] + J.job->fp.part_size[p];
  prime_stream( L, lds, J );
  uint32_t v = 0;
  for ( int k = 0; k < 4; k++ ) { v = ( v << 8 ) | ( L.rpos < L.rend ? ring_byte( lds, L.rpos ) : 0u ); L.rpos++; }
  L.value = v; L.count = 24; L.range = 255;
}

AA_HD inline void save_partition( const Lane & L, uint8_t * lds )
{
  uint32_t * s = reinterpret_cast<uint32_t *>( lds + kPart + 16 * L.part );
  s[0] = L.value; s[1] = L.range | ( static_cast<uint32_t>( L.count ) << 8 ); s[2] = L.rpos; s[3] = 1;
}

AA_HD inline void switch_partition( Lane & L, uint8_t * lds, const Frame & J, uint32_t p )
{
  save_partition( L, lds );
  const uint32_t * s = reinterpret_cast<const uint32_t *>( lds + kPart + 16 * p );
  if ( !s[3] ) { start_partition( L, lds, J, p ); return; }
  L.part = p;
  L.value = s[0]; L.range = s[1] & 255u; L.count = static_cast<int32_t>( s[1] >> 8 ); L.rpos = s[2];
  L.rend = J.job->fp.part_off[p] + J.job->fp.part_size[p];
  prime_stream( L, lds, J );
}

// ---- every kPeriod steps, all lanes together: land the chunks requested a period ago, request the next ---------------
AA_HD inline void top_up( Lane & L, uint8_t * lds, const Frame & J )
{
  if ( L.st == ST_DONE ) return;
  if ( L.pend_wpos == L.wpos ) {
    for ( uint32_t k = 0; k < kChunks; k++ ) lds_store16( lds, kStream + ( ( L.wpos + 16 * k ) & ( kRing - 1 ) ), L.pend[k] );
    L.wpos += 16 * kChunks;
  }
  if ( L.pend_mwpos == L.mwpos ) {
    for ( uint32_t k = 0; k < kChunks; k++ ) lds_store16( lds, kMeta + ( ( L.mwpos + 16 * k ) & ( kRing - 1 ) ), L.mpend[k] );
    L.mwpos += 16 * kChunks;
  }
  L.pend_wpos = L.pend_mwpos = kNoPend;
  // a request may be written a period from now iff the ring then still has room: lead <= kRing - 16*kChunks now
  if ( L.wpos - L.rpos <= kRing - 16 * kChunks ) {
    for ( uint32_t k = 0; k < kChunks; k++ ) {
      const uint32_t at = L.wpos + 16 * k;
      Chunk16 c; c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0;
      if ( at < J.data_padded ) c = load16( J.data + at );
      L.pend[k] = c;
    }
    L.pend_wpos = L.wpos;
  }
  if ( L.mwpos - L.mi <= kRing - 16 * kChunks ) {
    for ( uint32_t k = 0; k < kChunks; k++ ) {
      const uint32_t at = L.mwpos + 16 * k;
      Chunk16 c; c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0;
      if ( at < J.flags_padded ) c = load16( J.mbflags + at );
      L.mpend[k] = c;
    }
    L.pend_mwpos = L.mwpos;
  }
}

// ---- block / macroblock transitions ----------------------------------------------------------------------------------
// blk: parse order within the macroblock: 0 = Y2, 1..16 = Y, 17..20 = U, 21..24 = V (macroblock.cc:480-500)
AA_HD inline void nz_bits_of( uint32_t blk, uint32_t & a, uint32_t & l )
{
  if ( blk == 0 ) { a = l = 8; }
  else if ( blk <= 16 ) { const uint32_t b = blk - 1; a = b & 3; l = b >> 2; }
  else { const uint32_t k = blk - 17, pl = k >> 2; a = 4 + 2 * pl + ( k & 1 ); l = 4 + 2 * pl + ( ( k >> 1 ) & 1 ); }
}

AA_HD inline void setup_block( Lane & L )
{
  const uint32_t has_y2 = L.flags & AA_MB_HAS_Y2;
  uint32_t type, first = 0;
  if ( L.blk == 0 ) type = Y2;
  else if ( L.blk <= 16 ) { type = has_y2 ? Y_AFTER_Y2 : Y_WITHOUT_Y2; first = has_y2 ? 1 : 0; }
  else type = UV;
  uint32_t a, l;
  nz_bits_of( L.blk, a, l );
  const uint32_t ctx = ( ( L.above_nz >> a ) & 1 ) + ( ( L.left_nz >> l ) & 1 );
  L.typeoff = kProbs + type * 264;
  L.idx = first;
  L.rowoff = L.typeoff + band_of( first ) * 33 + ctx * 11;
  L.st = 0; L.paddr = L.rowoff;
  L.nonzero = 0;
}

AA_HD inline void store_mb( const Frame & J, uint32_t mi, uint32_t nz_mask, uint32_t coeff_index, uint32_t flags )
{
  AA_GLOBAL aa_mb_info * mb = J.mbs + mi;
  mb->nz_mask = nz_mask;
  mb->coeff_index = coeff_index;
  mb->flags = static_cast<uint8_t>( flags );
}

AA_HD inline void end_macroblock( Lane & L, uint8_t * lds, const Frame & J )
{
  if ( L.y2_nz ) {                                  // Y2 is parsed first but stored after the macroblock's other blocks
    const V16 * src = reinterpret_cast<const V16 *>( lds + kY2 );
    AA_GLOBAL V16 * dst = (AA_GLOBAL V16 *) ( J.coeffs + static_cast<size_t>( L.coeff_blocks ) * 16 );
    dst[0] = src[0]; dst[1] = src[1];
    V16 z; z.x = z.y = z.z = z.w = 0;
    V16 * y2 = reinterpret_cast<V16 *>( lds + kY2 );
    y2[0] = z; y2[1] = z;
    L.nz_mask |= 1u << 24;
    L.coeff_blocks++;
    zero_slot( J, L.coeff_blocks );
  }
  reinterpret_cast<uint16_t *>( lds + kAbove )[L.col] = static_cast<uint16_t>( L.above_nz );
  const uint32_t has_y2 = L.flags & AA_MB_HAS_Y2;
  uint32_t flags = L.flags;
  if ( L.nz_mask ) flags |= AA_MB_HAS_NONZERO;
  else if ( has_y2 ) flags |= AA_MB_LF_SKIP_INNER;
  store_mb( J, L.mi, L.nz_mask, L.mb_first, flags );
  L.mi++; L.col++;
  L.st = ST_MB;
}

AA_HD inline void end_block( Lane & L, uint8_t * lds, const Frame & J )
{
  uint32_t a, l;
  nz_bits_of( L.blk, a, l );
  const uint32_t nz = L.nonzero;
  L.above_nz = ( L.above_nz & ~( 1u << a ) ) | ( nz << a );
  L.left_nz = ( L.left_nz & ~( 1u << l ) ) | ( nz << l );
  if ( nz ) {
    if ( L.blk == 0 ) L.y2_nz = 1;
    else { L.nz_mask |= 1u << ( L.blk - 1 ); L.coeff_blocks++; zero_slot( J, L.coeff_blocks ); }
  }
  L.blk++;
  if ( L.blk == 25 ) end_macroblock( L, lds, J );
  else setup_block( L );
}

// st == ST_MB: take the next macroblock (a skipped one is finished on the spot)
AA_HD inline void begin_macroblock( Lane & L, uint8_t * lds, const Frame & J )
{
  if ( L.mi == J.nmb ) {
    AA_GLOBAL FrameSummary * sum = (AA_GLOBAL FrameSummary *) J.job->summary;
    sum->num_coeff_blocks = L.coeff_blocks;
    sum->steps = L.steps;
    L.st = ST_DONE;
    return;
  }
  if ( L.col == J.mbw ) {
    L.col = 0; L.row++; L.left_nz = 0;
    if ( J.nparts > 1 ) switch_partition( L, lds, J, L.row % J.nparts );
  }
  const uint32_t flags = lds[kMeta + ( L.mi & ( kRing - 1 ) )];
  uint32_t above = reinterpret_cast<const uint16_t *>( lds + kAbove )[L.col];
  const uint32_t has_y2 = flags & AA_MB_HAS_Y2;
  L.mb_first = L.coeff_blocks;
  if ( flags & AA_MB_SKIP ) {
    const uint32_t keep = has_y2 ? 0u : 0x100u;      // a non-coded Y2 leaves its chain untouched (frame.cc:255-269)
    above &= keep; L.left_nz &= keep;
    reinterpret_cast<uint16_t *>( lds + kAbove )[L.col] = static_cast<uint16_t>( above );
    store_mb( J, L.mi, 0, L.mb_first, flags | ( has_y2 ? AA_MB_LF_SKIP_INNER : 0u ) );
    L.mi++; L.col++;
    return;
  }
  L.flags = flags; L.above_nz = above; L.nz_mask = 0; L.y2_nz = 0;
  L.blk = has_y2 ? 0 : 1;
  setup_block( L );
}

// ---- one step -------------------------------------------------------------------------------------------------------
AA_HD inline void step( Lane & L, uint8_t * lds, const Frame & J )
{
  if ( L.st == ST_DONE ) return;
  if ( ++L.steps > J.max_steps ) {             // cannot happen for any input; if it does the frame is reported, not hung on
    AA_GLOBAL FrameSummary * sum = (AA_GLOBAL FrameSummary *) J.job->summary;
    sum->num_coeff_blocks = L.coeff_blocks; sum->steps = 0xFFFFFFFFu;
    L.st = ST_DONE;
    return;
  }
  if ( L.st == ST_MB ) { begin_macroblock( L, lds, J ); if ( L.st >= ST_MB ) return; }

  // the two LDS reads of a step: the probability of this node and the next stream byte
  const uint32_t prob = lds[L.paddr];
  const uint32_t raw = ring_byte( lds, L.rpos );
  // top the window up by one byte whenever one fits: a decode shifts out at most 7 bits, so the 8 bits being compared are
  // always real (count >= 0) and the refill is never on the critical path
  if ( L.count <= 16 ) {
    const uint32_t byte = L.rpos < L.rend ? raw : 0u;      // bytes past the end of a partition read as zero (bool_decoder.hh:56-65)
    L.value |= byte << ( 16 - L.count );
    L.count += 8;
    L.rpos++;
  }
  // BoolDecoder::get (bool_decoder.hh:67-107)
  const uint32_t split = 1 + ( ( ( L.range - 1 ) * prob ) >> 8 );
  const uint32_t bigsplit = split << 24;
  const uint32_t bit = L.value >= bigsplit ? 1u : 0u;
  uint32_t range = bit ? L.range - split : split;
  if ( bit ) L.value -= bigsplit;
  const int shift = __builtin_clz( range ) - 24;
  L.range = range << shift;
  L.value <<= shift;
  L.count -= shift;

  bool block_done = false;
  if ( L.st <= 10 ) {
    const uint32_t sh = L.st * 4;
    const uint32_t nx = static_cast<uint32_t>( ( bit ? kNext1 : kNext0 ) >> sh ) & 15u;
    const uint32_t arg = static_cast<uint32_t>( ( bit ? kArg1 : kArg0 ) >> sh ) & 15u;
    if ( nx <= 10 ) { L.st = nx; L.paddr = L.rowoff + nx; }
    else if ( nx == NX_SIGN ) { L.mag = arg; L.ctx_next = arg == 1 ? 1 : 2; L.st = ST_SIGN; L.paddr = kSignP; }
    else if ( nx == NX_EXTRA ) {
      L.xrem = static_cast<uint32_t>( kXLen >> ( arg * 4 ) ) & 15u;
      L.xbase = static_cast<uint32_t>( kXBase >> ( arg * 8 ) ) & 255u;
      L.paddr = kXtab + ( static_cast<uint32_t>( kXOff >> ( arg * 4 ) ) & 15u );
      L.mag = 0; L.ctx_next = 2; L.st = ST_EXTRA;
    } else if ( nx == NX_ZERO ) {                          // ZERO token: no EOB check at the next position
      L.idx++;
      if ( L.idx == 16 ) block_done = true;
      else { L.rowoff = L.typeoff + band_of( L.idx ) * 33; L.st = 1; L.paddr = L.rowoff + 1; }
    } else block_done = true;                              // EOB
  } else if ( L.st == ST_EXTRA ) {
    L.mag = ( L.mag << 1 ) + bit;
    L.paddr++;
    if ( --L.xrem == 0 ) { L.mag += L.xbase; L.st = ST_SIGN; L.paddr = kSignP; }
  } else {                                                 // sign: the token is complete
    const int16_t v = static_cast<int16_t>( bit ? -static_cast<int32_t>( L.mag ) : static_cast<int32_t>( L.mag ) );
    const uint32_t zz = static_cast<uint32_t>( kZigzagNib >> ( L.idx * 4 ) ) & 15u;
    if ( L.blk == 0 ) reinterpret_cast<int16_t *>( lds + kY2 )[zz] = v;
    else J.coeffs[static_cast<size_t>( L.coeff_blocks ) * 16 + zz] = v;
    L.nonzero = 1;
    L.idx++;
    if ( L.idx == 16 ) block_done = true;
    else { L.rowoff = L.typeoff + band_of( L.idx ) * 33 + L.ctx_next * 11; L.st = 0; L.paddr = L.rowoff; }
  }
  if ( block_done ) end_block( L, lds, J );
}

// ---- a lane's life ---------------------------------------------------------------------------------------------------
AA_HD inline void begin_frame( Lane & L, uint8_t * lds, const Frame & J )
{
  // lane LDS: probabilities, constants, zeroed above-row flags / Y2 block / partition save area
  const AA_GLOBAL uint32_t * src = (const AA_GLOBAL uint32_t *) &J.job->fp.coeff_probs[0][0][0][0];
  uint32_t * dst = reinterpret_cast<uint32_t *>( lds + kProbs );
  for ( uint32_t k = 0; k < 1056 / 4; k++ ) dst[k] = src[k];
  for ( uint32_t k = 0; k < 27; k++ ) lds[kXtab + k] = kXtabInit[k];
  for ( uint32_t k = 0; k < ( 32 + 128 ) / 4; k++ ) reinterpret_cast<uint32_t *>( lds + kY2 )[k] = 0;
  for ( uint32_t k = 0; k < J.mbw; k++ ) reinterpret_cast<uint16_t *>( lds + kAbove )[k] = 0;
  L.mi = 0; L.col = 0; L.row = 0; L.left_nz = 0; L.coeff_blocks = 0; L.steps = 0;
  L.flags = L.above_nz = L.nz_mask = L.y2_nz = L.mb_first = 0;
  L.blk = L.idx = L.typeoff = L.rowoff = L.nonzero = L.ctx_next = 0;
  L.paddr = kSignP; L.mag = L.xrem = L.xbase = 0;
  zero_slot( J, 0 );
  start_partition( L, lds, J, 0 );
  // flag ring: macroblocks [0, kRing)
  for ( uint32_t k = 0; k < kRing / 16; k++ ) {
    Chunk16 c; c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0;
    if ( 16 * k < J.flags_padded ) c = load16( J.mbflags + 16 * k );
    lds_store16( lds, kMeta + 16 * k, c );
  }
  L.mwpos = kRing;
  L.pend_mwpos = kNoPend;
  L.st = ST_MB;
}

} // namespace tok
} // namespace aa
