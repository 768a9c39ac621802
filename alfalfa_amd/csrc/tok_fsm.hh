// Token parse (Block::parse_tokens tokens.cc:50-135, Macroblock::parse_tokens macroblock.cc:475-502, driven per row as in
// frame.cc:121-137) as a flat, TABLE-DRIVEN state machine: ONE boolean decode per step, every lane of a wave at its own
// place in its own frame.
//
// Why this shape.  A GPU lane cannot afford the natural loop nest (token tree inside block loop inside macroblock loop):
// the lanes of a wave would wait for the longest block / macroblock among them at every level, and -- measured on MI355X
// with a first, branchy version of this file -- a wave executes the UNION of every path any of its lanes takes: ~600
// instructions per bool, 2 us per step.  So the step is written to be the same short instruction sequence for every lane:
//   * the token tree, the extra-bit chains of the six DCT categories and the sign are 38 NODES; a node's record (8 bytes in
//     LDS, two halves selected by the decoded bit) says where to go next, which probability that node reads and what to do
//     (set a literal magnitude, shift a bit in, emit a coefficient, count a zero, end the block);
//   * what the record asks for is applied with selects, not branches; only the coefficient store, the block boundary and
//     the (rare) macroblock boundary are predicated regions.
// Block ends, macroblock ends, row ends and the end of the frame are transitions of the same machine, so lanes never wait
// for each other.
//
// The same code is the device kernel's body (parse_kernels.hip) and, compiled for the host, what tests/cpp/fsm_sim.cc replays
// lane by lane against the host parser -- a lane touches nothing but its own state, so one lane at a time is exact.
//
// Memory of a lane ("LDS" = the lane's slice of the workgroup's LDS on the GPU, a plain buffer on the host):
//   LDS   this frame's token probabilities + the fixed extra-bit probabilities (one byte read per step)
//         a 256-byte ring of the current partition's bytes and a 256-entry ring of macroblock header flags; both are
//         topped up from HBM every kPeriod steps, for all lanes at once, with loads issued one period ahead -- the step
//         itself never waits for HBM
//         the above-row non-zero flags (9 bits per macroblock column), saved decoder states of the other DCT partitions
//   HBM   the frame's compressed bytes, flags[mi] from the header kernel (in), coefficient blocks + nz_mask / coeff_index /
//         flags of every macroblock record (out, fire-and-forget stores)
// Shared by the lanes of a workgroup (LDS, read-only): the node records and the per-block constants.
#pragma once
#include <initializer_list>

#include "parse_common.hh"

namespace aa {

struct alignas( 16 ) V16 { uint32_t x, y, z, w; };   // one 16-byte memory transaction
struct alignas( 8 ) V8 { uint32_t x, y; };

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
  uint32_t steps;               // steps of the token lane (diagnostics); 0xFFFFFFFF: the lane hit its step bound
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
constexpr uint32_t kSignX = 26;           // index of the sign's probability in that table
constexpr uint32_t kStream = 1280;        // stream ring
constexpr uint32_t kMeta = kStream + kRing;
constexpr uint32_t kPart = kMeta + kRing; // 8 saved partition decoders x 16 bytes
constexpr uint32_t kAbove = kPart + 128;  // uint16 per macroblock column
AA_HD constexpr uint32_t lane_lds_bytes( uint32_t mbw ) { return ( kAbove + 2 * mbw + 255 ) & ~255u; }

// dct_cat probabilities (tokens.cc:36-48) laid out back to back: cat1 @0, cat2 @1, cat3 @3, cat4 @6, cat5 @10, cat6 @15, sign @26
constexpr uint8_t kXtabInit[27] = { 159, 165, 145, 173, 148, 140, 176, 155, 140, 135, 180, 157, 141, 134, 130,
                                    254, 254, 243, 230, 196, 177, 153, 140, 133, 130, 129, 128 };

// ---- nodes ----------------------------------------------------------------------------------------------------------
// 0..10   the token tree (tokens.cc:73-124); node k reads probability k of the current (type, band, context) row
// 11..36  extra bits: cat1 = 11, cat2 = 12-13, cat3 = 14-16, cat4 = 17-20, cat5 = 21-25, cat6 = 26-36 (node e reads kXtab[e-11])
// 37      sign
// 38..40  not decoding: just completed a macroblock / at a macroblock boundary / done with the frame
enum : uint32_t { N_SIGN = 37, N_MBDONE = 38, N_MB = 39, N_DONE = 40, kNodes = 38 };
enum : uint32_t { A_NONE = 0, A_SETMAG = 1, A_XBIT = 2, A_EMIT = 3, A_ZERO = 4, A_EOB = 5 };
// half of a node record = what a decoded 0 / 1 at that node means:
//   [0,6) next node  [6,14) probability index of the next node  [14] that index is relative to the current row (else to kXtab)
//   [15,18) literal magnitude  [18,25) constant added to the magnitude when it is emitted  [25,27) context the token leaves
//   [29,32) action
constexpr uint32_t half( uint32_t next, uint32_t poff, uint32_t rowrel, uint32_t act, uint32_t k = 0, uint32_t addv = 0, uint32_t c = 0 )
{
  return next | ( poff << 6 ) | ( rowrel << 14 ) | ( k << 15 ) | ( addv << 18 ) | ( c << 25 ) | ( act << 29 );
}
constexpr uint32_t tree( uint32_t k ) { return half( k, k, 1, A_NONE ); }                       // on to tree node k
constexpr uint32_t lit( uint32_t mag, uint32_t ctx ) { return half( N_SIGN, kSignX, 0, A_SETMAG, mag, 0, ctx ); }   // DCT_1..4
// dct_catN: value = base + N extra bits, base = 2^N + 3 for cat1..5 (start the shift register at 1, add 3), 67 for cat6
// (start at 0, add 67)
constexpr uint32_t cat( uint32_t first_node, uint32_t start, uint32_t addv ) { return half( first_node, first_node - 11, 0, A_SETMAG, start, addv, 2 ); }
constexpr uint32_t xbit( uint32_t e, bool last ) { return last ? half( N_SIGN, kSignX, 0, A_XBIT ) : half( e + 1, e + 1 - 11, 0, A_XBIT ); }
struct NodeTable { V8 n[kNodes]; };
constexpr NodeTable make_nodes()
{
  NodeTable t {};
  t.n[0] = { half( 0, 0, 1, A_EOB ), tree( 1 ) };
  t.n[1] = { half( 1, 1, 1, A_ZERO ), tree( 2 ) };          // a ZERO token is followed by node 1 of the next position (no EOB check)
  t.n[2] = { lit( 1, 1 ), tree( 3 ) };
  t.n[3] = { tree( 4 ), tree( 6 ) };
  t.n[4] = { lit( 2, 2 ), tree( 5 ) };
  t.n[5] = { lit( 3, 2 ), lit( 4, 2 ) };
  t.n[6] = { tree( 7 ), tree( 8 ) };
  t.n[7] = { cat( 11, 1, 3 ), cat( 12, 1, 3 ) };
  t.n[8] = { tree( 9 ), tree( 10 ) };
  t.n[9] = { cat( 14, 1, 3 ), cat( 17, 1, 3 ) };
  t.n[10] = { cat( 21, 1, 3 ), cat( 26, 0, 67 ) };
  for ( uint32_t e = 11; e <= 36; e++ ) {
    const bool last = e == 11 || e == 13 || e == 16 || e == 20 || e == 25 || e == 36;
    t.n[e] = { xbit( e, last ), xbit( e, last ) };
  }
  t.n[N_SIGN] = { half( 0, 0, 1, A_EMIT ), half( 0, 0, 1, A_EMIT ) };   // then the EOB check of the next position
  return t;
}
constexpr NodeTable kNodeTable = make_nodes();

// ---- blocks ---------------------------------------------------------------------------------------------------------
// parse order within a macroblock (macroblock.cc:480-500): 0 = Y2, 1..16 = Y, 17..20 = U, 21..24 = V.  Per block: where its
// "above" / "left" non-zero flags live in Lane::ctxbits (above: bits 0-8 = 4 Y columns, 2 U, 2 V, Y2; left: bits 16-24),
// the kind of probabilities it reads, its bit in nz_mask.
struct BlockTable { V8 b[26]; };
constexpr BlockTable make_blocks()
{
  BlockTable t {};
  for ( uint32_t blk = 0; blk < 25; blk++ ) {
    uint32_t a = 8, l = 8, sel = 2, bit = 24;                       // Y2
    if ( blk >= 1 && blk <= 16 ) { const uint32_t b = blk - 1; a = b & 3; l = b >> 2; sel = 0; bit = b; }
    else if ( blk >= 17 ) { const uint32_t k = blk - 17, pl = k >> 2; a = 4 + 2 * pl + ( k & 1 ); l = 4 + 2 * pl + ( ( k >> 1 ) & 1 ); sel = 1; bit = blk - 1; }
    t.b[blk] = { a | ( ( 16 + l ) << 8 ) | ( sel << 16 ) | ( bit << 24 ), ( 1u << a ) | ( 1u << ( 16 + l ) ) };
  }
  t.b[25] = { 0, 0 };
  return t;
}
constexpr BlockTable kBlockTable = make_blocks();

struct Tables { const V8 * nodes; const V8 * blocks; };       // where a workgroup keeps the two tables (LDS on the GPU)
constexpr uint32_t kTablesBytes = ( sizeof( NodeTable ) + sizeof( BlockTable ) + 15 ) & ~15u;

constexpr uint64_t nib( std::initializer_list<unsigned> v ) { uint64_t r = 0; unsigned i = 0; for ( unsigned x : v ) r |= static_cast<uint64_t>( x ) << ( 4 * i++ ); return r; }
constexpr uint64_t tri( std::initializer_list<unsigned> v ) { uint64_t r = 0; unsigned i = 0; for ( unsigned x : v ) r |= static_cast<uint64_t>( x ) << ( 3 * i++ ); return r; }
constexpr uint64_t kZigzagNib = nib( { 0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15 } );
constexpr uint64_t kBandTri = tri( { 0, 1, 2, 3, 6, 4, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7, 0 } );   // coefficient band of position 0..16

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
  // per macroblock at most 25 blocks x 16 tokens x (11 tree nodes + 11 extra bits + sign), plus boundary steps
  const uint64_t bound = static_cast<uint64_t>( job->nmb ) * ( 25u * 16u * 23u + 4u ) + 4096u;
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
// bytes at offsets >= end (absolute offsets at .. at+15) cleared: partitions end anywhere, and past the end a boolean
// decoder reads zeros (bool_decoder.hh:56-65) -- done once per chunk so that the step does not have to ask
AA_HD inline Chunk16 mask_past_end( Chunk16 c, uint32_t at, uint32_t end )
{
  for ( uint32_t k = 0; k < 4; k++ ) {
    const uint32_t a = at + 4 * k;
    const uint32_t keep = a >= end ? 0u : ( end - a >= 4 ? 0xFFFFFFFFu : ( ( 1u << ( 8 * ( end - a ) ) ) - 1u ) );
    c.w[k] &= keep;
  }
  return c;
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
  // token in progress
  uint32_t node;                  // node about to be decoded (N_MB / N_DONE: not decoding)
  uint32_t paddr;                 // LDS offset of its probability
  uint32_t rowoff, typeoff;       // LDS offsets of the current probability row / of this block type's probabilities
  uint32_t idx, mag, tinfo, nonzero;   // tinfo: constant to add at emission | context left behind << 7
  // block in progress
  uint32_t blk, nzsel, blkbit;
  // macroblock in progress
  uint32_t ctxbits;               // non-zero flags: above (this column) bits 0-8, left bits 16-24
  uint32_t flags, nz_mask, mb_first, coeff_blocks, ytypeoff, yfirst;
  // position
  uint32_t mi, col, row, part;
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
// synchronous (re)fill of the whole stream ring around rpos: start of a partition
AA_HD inline void prime_stream( Lane & L, uint8_t * lds, const Frame & J )
{
  const uint32_t base = L.rpos & ~15u;
  for ( uint32_t k = 0; k < kRing / 16; k++ ) {
    const uint32_t at = base + 16 * k;
    Chunk16 c; c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0;
    if ( at < J.data_padded ) c = load16( J.data + at );
    lds_store16( lds, kStream + ( at & ( kRing - 1 ) ), mask_past_end( c, at, L.rend ) );
  }
  L.wpos = base + kRing;
  L.pend_wpos = kNoPend;
}

// decoder of partition `p` at its first bit (BoolDecoder ctor, bool_decoder.hh:45-54)
AA_HD inline void start_partition( Lane & L, uint8_t * lds, const Frame & J, uint32_t p )
{
  L.part = p;
  L.rpos = J.job->fp.part_off[p];
  L.rend = J.job->fp.part_off[p] + J.job->fp.part_size[p];
  prime_stream( L, lds, J );
  uint32_t v = 0;
  for ( int k = 0; k < 4; k++ ) { v = ( v << 8 ) | lds[kStream + ( L.rpos & ( kRing - 1 ) )]; L.rpos++; }
  L.value = v; L.count = 24; L.range = 255;
}

AA_HD inline void switch_partition( Lane & L, uint8_t * lds, const Frame & J, uint32_t p )
{
  uint32_t * s = reinterpret_cast<uint32_t *>( lds + kPart + 16 * L.part );
  s[0] = L.value; s[1] = L.range | ( static_cast<uint32_t>( L.count ) << 8 ); s[2] = L.rpos; s[3] = 1;
  const uint32_t * t = reinterpret_cast<const uint32_t *>( lds + kPart + 16 * p );
  if ( !t[3] ) { start_partition( L, lds, J, p ); return; }
  L.part = p;
  L.value = t[0]; L.range = t[1] & 255u; L.count = static_cast<int32_t>( t[1] >> 8 ); L.rpos = t[2];
  L.rend = J.job->fp.part_off[p] + J.job->fp.part_size[p];
  prime_stream( L, lds, J );
}

// ---- every kPeriod steps, all lanes together: land the chunks requested a period ago, request the next ---------------
AA_HD inline void top_up( Lane & L, uint8_t * lds, const Frame & J )
{
  if ( L.node == N_DONE ) return;
  if ( L.pend_wpos == L.wpos ) {
    for ( uint32_t k = 0; k < kChunks; k++ )
      lds_store16( lds, kStream + ( ( L.wpos + 16 * k ) & ( kRing - 1 ) ), mask_past_end( L.pend[k], L.wpos + 16 * k, L.rend ) );
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
#if defined( __HIP_DEVICE_COMPILE__ )
#define AA_ANY( x ) ( __any( x ) != 0 )        // wave-uniform: does any lane ...
#else
#define AA_ANY( x ) ( x )
#endif
#if defined( __clang__ )
#define AA_SELECT( c ) __builtin_unpredictable( c )   // keep `c ? a : b` a select: lanes disagree, there is nothing to predict
#else
#define AA_SELECT( c ) ( c )
#endif

AA_HD inline void store_mb( const Frame & J, uint32_t mi, uint32_t nz_mask, uint32_t coeff_index, uint32_t flags )
{
  AA_GLOBAL aa_mb_info * mb = J.mbs + mi;
  mb->nz_mask = nz_mask;
  mb->coeff_index = coeff_index;
  mb->flags = static_cast<uint8_t>( flags );
}

// Make block `L.blk` (its BlockTable entry in `e`) the current one: contexts from the non-zero flags, first probability row.
AA_HD inline void setup_block( Lane & L, const V8 e )
{
  const uint32_t a = e.x & 255u, l = ( e.x >> 8 ) & 255u, sel = ( e.x >> 16 ) & 255u;
  L.nzsel = e.y;
  L.blkbit = 1u << ( e.x >> 24 );
  const uint32_t ctx = ( ( L.ctxbits >> a ) & 1u ) + ( ( L.ctxbits >> l ) & 1u );
  L.typeoff = sel == 0 ? L.ytypeoff : ( sel == 1 ? kProbs + UV * 264u : kProbs + Y2 * 264u );
  L.idx = sel == 0 ? L.yfirst : 0u;                         // Y blocks after a Y2 start at position 1 (tokens.cc:61)
  L.rowoff = L.typeoff + L.idx * 33u + ctx * 11u;           // band of position 0 / 1 is 0 / 1
  L.node = 0; L.paddr = L.rowoff;
  L.nonzero = 0;
}

// The slow path, for lanes at a macroblock boundary (node N_MBDONE: the step just completed one; N_MB: waiting for flags):
// take macroblocks until one has tokens (skipped ones are settled on the spot), the flag ring runs dry (try again later)
// or the frame ends.  Everything rare lives here: row ends, partition switches, the end of the frame.
AA_HD inline void macroblock_boundary( Lane & L, uint8_t * lds, const Tables & T, const Frame & J )
{
  uint16_t * const above = reinterpret_cast<uint16_t *>( lds + kAbove );
  if ( L.node == N_MBDONE ) { L.mi++; L.col++; L.node = N_MB; }
  if ( ++L.steps > J.max_steps ) {             // cannot happen for any input; if it does the frame is reported, not hung on
    AA_GLOBAL FrameSummary * sum = (AA_GLOBAL FrameSummary *) J.job->summary;
    sum->num_coeff_blocks = L.coeff_blocks; sum->steps = 0xFFFFFFFFu;
    L.node = N_DONE;
    return;
  }
  for ( ;; ) {
    if ( L.mi == J.nmb ) {
      AA_GLOBAL FrameSummary * sum = (AA_GLOBAL FrameSummary *) J.job->summary;
      sum->num_coeff_blocks = L.coeff_blocks;
      sum->steps = L.steps;
      L.node = N_DONE;
      return;
    }
    if ( L.mi >= L.mwpos ) return;                          // flags not here yet (only a long run of skipped macroblocks gets ahead of the ring)
    if ( L.col == J.mbw ) {
      L.col = 0; L.row++; L.ctxbits = 0;
      if ( J.nparts > 1 ) switch_partition( L, lds, J, L.row % J.nparts );
    }
    const uint32_t flags = lds[kMeta + ( L.mi & ( kRing - 1 ) )];
    const uint32_t has_y2 = flags & AA_MB_HAS_Y2;
    L.ctxbits = ( L.ctxbits & 0x01FF0000u ) | above[L.col];
    L.mb_first = L.coeff_blocks;
    if ( !( flags & AA_MB_SKIP ) ) {
      L.flags = flags; L.nz_mask = 0;
      L.ytypeoff = kProbs + ( has_y2 ? Y_AFTER_Y2 : Y_WITHOUT_Y2 ) * 264u;
      L.yfirst = has_y2 ? 1u : 0u;
      L.blk = has_y2 ? 0u : 1u;
      setup_block( L, T.blocks[L.blk] );
      return;
    }
    L.ctxbits &= has_y2 ? 0u : 0x01000100u;                 // a non-coded Y2 leaves its chain untouched (frame.cc:255-269)
    above[L.col] = static_cast<uint16_t>( L.ctxbits );
    store_mb( J, L.mi, 0, L.mb_first, flags | ( has_y2 ? AA_MB_LF_SKIP_INNER : 0u ) );
    L.mi++; L.col++;
  }
}

AA_HD inline bool at_boundary( const Lane & L ) { return L.node == N_MBDONE || L.node == N_MB; }

// ---- one step: decode one bool (lanes with a node to decode; the others sit it out) -------------------------------------
// Straight-line code: everything the bit can mean is computed and selected, the only predicated regions are stores.
AA_HD inline void step( Lane & L, uint8_t * lds, const Tables & T, const Frame & J )
{
  if ( L.node >= N_MBDONE ) return;
  L.steps++;

  // the LDS reads of a step; all addresses were known at the end of the previous one
  const uint32_t prob = lds[L.paddr];
  const uint32_t raw = lds[kStream + ( L.rpos & ( kRing - 1 ) )];
  const V8 rec = T.nodes[L.node];
  const V8 nextblk = T.blocks[L.blk + 1];

  // top the window up by one byte whenever one fits: a decode shifts out at most 7 bits, so the 8 bits being compared are
  // always real (count >= 0) and the refill is never on the critical path
  // (mask arithmetic rather than a conditional: the compiler would otherwise branch around the ring read and wait for it there)
  const uint32_t room = static_cast<uint32_t>( ( L.count - 17 ) >> 31 );        // all ones iff count <= 16
  L.value |= ( raw << ( ( 16 - L.count ) & 31 ) ) & room;
  L.count += static_cast<int32_t>( 8u & room );
  L.rpos -= room;

  // BoolDecoder::get (bool_decoder.hh:67-107)
  const uint32_t split = ( ( L.range - 1 ) * prob + 256u ) >> 8;        // = 1 + (((range - 1) * prob) >> 8)
  const uint32_t bigsplit = split << 24;
  const bool bit = L.value >= bigsplit;
  const uint32_t range = bit ? L.range - split : split;
  const uint32_t value = bit ? L.value - bigsplit : L.value;
  const int shift = __builtin_clz( range ) - 24;
  L.range = range << shift;
  L.value = value << shift;
  L.count -= shift;

  // what the node says this bit means
  const uint32_t h = bit ? rec.y : rec.x;
  const uint32_t act = h >> 29;
  const bool setm = act == A_SETMAG, emit = act == A_EMIT, zero = act == A_ZERO;
  const uint32_t shifted = 2 * L.mag + ( bit ? 1u : 0u );
  const uint32_t kept = AA_SELECT( act == A_XBIT ) ? shifted : L.mag;
  const uint32_t mag = AA_SELECT( setm ) ? ( h >> 15 ) & 7u : kept;
  const uint32_t tinfo = AA_SELECT( setm ) ? ( h >> 18 ) & 0x1FFu : L.tinfo;
  L.mag = mag; L.tinfo = tinfo;
  if ( emit ) {                                 // the sign: the token is complete (tokens.cc:126-133)
    const int32_t m = static_cast<int32_t>( mag + ( tinfo & 127u ) );
    const uint32_t zz = static_cast<uint32_t>( kZigzagNib >> ( L.idx * 4 ) ) & 15u;
    J.coeffs[static_cast<size_t>( L.coeff_blocks ) * 16 + zz] = static_cast<int16_t>( bit ? -m : m );
  }
  const uint32_t nonzero = L.nonzero | ( emit ? 1u : 0u );
  const bool adv = emit || zero;                // on to the next coefficient position
  const uint32_t idx = L.idx + ( adv ? 1u : 0u );
  const uint32_t ctx = zero ? 0u : tinfo >> 7;
  const uint32_t band = static_cast<uint32_t>( kBandTri >> ( idx * 3 ) ) & 7u;
  const uint32_t rowoff = adv ? L.typeoff + band * 33u + ctx * 11u : L.rowoff;
  const uint32_t node = h & 63u;
  const uint32_t paddr = ( ( h >> 14 ) & 1u ? rowoff : kXtab ) + ( ( h >> 6 ) & 255u );
  const bool bend = act == A_EOB || ( adv && idx == 16 );

  if ( !AA_ANY( bend ) ) {                      // (wave-uniform) nobody ends a block in this step
    L.nonzero = nonzero; L.idx = idx; L.rowoff = rowoff; L.node = node; L.paddr = paddr;
    return;
  }
  // ---- end of a block (applied to the lanes with `bend`) ----
  const uint32_t ctxbits = nonzero ? L.ctxbits | L.nzsel : L.ctxbits & ~L.nzsel;
  const bool commit = bend && nonzero;
  const uint32_t coeff_blocks = L.coeff_blocks + ( commit ? 1u : 0u );
  if ( commit ) zero_slot( J, coeff_blocks );
  const uint32_t nz_mask = commit ? L.nz_mask | L.blkbit : L.nz_mask;
  const uint32_t blk = L.blk + 1;
  const bool mbdone = bend && blk == 25;
  if ( mbdone ) {                               // the macroblock is complete: its record, its column's flags
    reinterpret_cast<uint16_t *>( lds + kAbove )[L.col] = static_cast<uint16_t>( ctxbits );
    uint32_t flags = L.flags;
    flags |= nz_mask ? AA_MB_HAS_NONZERO : ( ( flags & AA_MB_HAS_Y2 ) ? AA_MB_LF_SKIP_INNER : 0u );
    store_mb( J, L.mi, nz_mask, L.mb_first, flags );
  }
  // the block after it (never a Y2)
  const uint32_t a = nextblk.x & 255u, l = ( nextblk.x >> 8 ) & 255u, uv = ( nextblk.x >> 16 ) & 255u;
  const uint32_t nctx = ( ( ctxbits >> a ) & 1u ) + ( ( ctxbits >> l ) & 1u );
  const uint32_t ntypeoff = uv ? kProbs + UV * 264u : L.ytypeoff;
  const uint32_t nidx = uv ? 0u : L.yfirst;
  const uint32_t nrowoff = ntypeoff + nidx * 33u + nctx * 11u;
  L.coeff_blocks = coeff_blocks; L.nz_mask = nz_mask;
  L.ctxbits = bend ? ctxbits : L.ctxbits;
  L.blk = bend ? blk : L.blk;
  L.nzsel = bend ? nextblk.y : L.nzsel;
  L.blkbit = bend ? 1u << ( nextblk.x >> 24 ) : L.blkbit;
  L.typeoff = bend ? ntypeoff : L.typeoff;
  L.nonzero = bend ? 0u : nonzero;
  L.idx = bend ? nidx : idx;
  L.rowoff = bend ? nrowoff : rowoff;
  L.node = bend ? ( mbdone ? static_cast<uint32_t>( N_MBDONE ) : 0u ) : node;
  L.paddr = bend ? nrowoff : paddr;
}

// One period of a wave: kPeriod steps, leaving the hot loop whenever a lane has reached a macroblock boundary.
AA_HD inline void run_period( Lane & L, uint8_t * lds, const Tables & T, const Frame & J )
{
  uint32_t it = 0;
  while ( it < kPeriod ) {
    if ( AA_ANY( at_boundary( L ) ) ) {
      if ( at_boundary( L ) ) macroblock_boundary( L, lds, T, J );
      it++;                                                 // (a lane waiting for flags must not spin the period away)
      if ( !AA_ANY( L.node < N_MBDONE ) ) break;            // nobody has anything to decode
    }
    do { step( L, lds, T, J ); it++; } while ( it < kPeriod && !AA_ANY( at_boundary( L ) ) );
  }
}

// ---- a lane's life ---------------------------------------------------------------------------------------------------
AA_HD inline void begin_frame( Lane & L, uint8_t * lds, const Frame & J )
{
  // lane LDS: probabilities, constants, zeroed above-row flags / partition save area
  const AA_GLOBAL uint32_t * src = (const AA_GLOBAL uint32_t *) &J.job->fp.coeff_probs[0][0][0][0];
  uint32_t * dst = reinterpret_cast<uint32_t *>( lds + kProbs );
  for ( uint32_t k = 0; k < 1056 / 4; k++ ) dst[k] = src[k];
  for ( uint32_t k = 0; k < 27; k++ ) lds[kXtab + k] = kXtabInit[k];
  for ( uint32_t k = 0; k < 128 / 4; k++ ) reinterpret_cast<uint32_t *>( lds + kPart )[k] = 0;
  for ( uint32_t k = 0; k < J.mbw; k++ ) reinterpret_cast<uint16_t *>( lds + kAbove )[k] = 0;
  L.mi = 0; L.col = 0; L.row = 0; L.ctxbits = 0; L.coeff_blocks = 0; L.steps = 0;
  L.flags = L.nz_mask = L.mb_first = L.ytypeoff = L.yfirst = 0;
  L.blk = L.idx = L.typeoff = L.rowoff = L.nonzero = L.tinfo = L.nzsel = L.blkbit = 0;
  L.paddr = kXtab; L.mag = 0;
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
  L.node = N_MB;
}

} // namespace tok
} // namespace aa
