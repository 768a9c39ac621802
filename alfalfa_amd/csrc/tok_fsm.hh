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
//
// Where the coefficient blocks go.  A frame stores only its non-zero 4x4 blocks, and how many those are is known when the
// parse is over -- so a lane does not get a worst-case sized piece (25 blocks per macroblock: 6.5 MB per 1080p frame, of
// which video uses a fraction) but draws CHUNKS of kChunkBlocks blocks from a pool shared by the whole GPU (CoeffPool: a
// ring of free chunk numbers) whenever the chunk it is filling cannot take another macroblock.  All chunks are pieces of ONE
// heap, so a macroblock's coeff_index is simply the index of its first block in that heap and the reconstruction kernels
// address it as before (heap base + 16 * index).  The chunks a frame took are listed in the frame's records (chunk_list) and
// go back to the ring, on the device, when the frame's records are released.
//
// Packed coefficients (the second storage format, template parameter PK of the functions below; ALFALFA_AMD_PACKED=1 selects
// it for a context).  Video leaves 2-4 of a stored block's 16 coefficients non-zero, so a dense block is mostly zeros: 611
// bytes per 1080p macroblock, which is what bounds how many parsed frames a GPU can hold ahead of reconstruction.  Packed, a
// stored block is ONE MASK WORD (bit k: zigzag position k holds a coefficient) followed by those coefficients as 16-bit
// values in zigzag order; a macroblock's blocks follow each other in parse order as before, a macroblock never straddles a
// chunk, and where it starts is recorded per macroblock (ParseJob::packed_pos).  The lane writes one 16-bit value per
// coefficient and one mask word per block -- fewer stores and fewer instructions than the dense path, which zeroes 32 bytes
// ahead of every block.  Reconstruction reads dense blocks: a frame's words are expanded into a transient dense array when
// the frame is handed to reconstruction (coeff_pack.hh, k_expand_coeffs), macroblock by macroblock, coeff_index being set
// then.
//
// One lane per DCT partition (template parameter MP of the functions below: "capable of"; a frame uses it when its job says
// so, ParseJob::mp_stride != 0).  Row r of a frame is coded in partition r % P (frame.cc:119-137), so a frame with P = 2, 4 or 8
// partitions is P independent bit streams -- tied together only by the above-row non-zero flags a macroblock's contexts
// need.  P lanes OF ONE WAVE take one partition each: lane p decodes rows p, p + P, ... as if they were a frame of their own
// (its own sequence of macroblock flags: the header kernel lays them out partition by partition), all lanes read and write ONE
// above-row array (in the slice of the lane that has the frame's last row -- it finishes last), and a lane starts macroblock
// (r, c) only when the lane of row r - 1 has completed (r - 1, c): a progress word per lane in LDS, checked at the macroblock
// boundary.  Rows then follow each other like a wavefront, one macroblock apart at best, and a key frame's chain is P times
// shorter.  Chunks are drawn per lane, listed in the frame's one chunk list through a counter in LDS; the lane that finishes
// last reports the frame.  A frame whose wave has fewer than P idle lanes when it is drawn runs on one lane as before.
//
// Who runs a frame.  Lanes are WORKERS (k_token_workers): a lane that has finished its frame takes the next ParseJob from a
// queue in HBM (TokQueue) at once, so a wave does not wait for its longest lane and a launch not for its longest wave.
#pragma once
#include <initializer_list>

#include "parse_common.hh"

#ifndef AA_STEP_UNROLL
#define AA_STEP_UNROLL 1              /* build parameter (A/B runs): the kBendEvery steps of a group unrolled */
#endif
#ifndef AA_STEP_SCHED_BARRIER
#define AA_STEP_SCHED_BARRIER 1       /* build parameter (A/B runs): a scheduling barrier between the load-independent and the load-dependent half of a step */
#endif
#ifndef AA_SLICE_SKEW
#define AA_SLICE_SKEW 0               /* build parameter (A/B runs) */
#endif
#ifndef AA_STEP_STORE_ALWAYS
#define AA_STEP_STORE_ALWAYS 2        /* build parameter (A/B runs): 0 = the coefficient store of a step under `if ( emit )`, 1 = always at blk, 2 = at blk or at the sink */
#endif
#ifndef AA_STEP_PRELOAD
#define AA_STEP_PRELOAD 1             /* build parameter (A/B runs): the step's LDS reads are asked for at the end of the previous step (tok::preload) */
#endif

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

// Written by the device parser into pinned host memory the GPU has mapped; the host polls `done` (the token lane's last
// store, behind a system-scope release: everything the frame's parse wrote to HBM is visible to kernels launched after the
// host has seen it).
struct FrameSummary {
  uint32_t num_coeff_blocks;
  uint32_t num_intra_mbs;       // (header kernel)
  uint32_t has_split;           // (header kernel)
  uint32_t steps;               // steps of the token lane (an upper bound on the bools it decoded)
  uint32_t num_chunks;          // coefficient chunks the frame took (= chunk_list[0])
  uint32_t status;              // TOK_OK ...
  uint32_t done;                // 1: the token lane is through with this frame
  uint32_t packed_words;        // packed storage (see "Packed coefficients"): 16-bit words the frame's coefficients take; 0: stored dense
};
enum : uint32_t { TOK_OK = 0, TOK_STEP_BOUND = 1, TOK_NO_MEMORY = 2,
                  TOK_HOST_FAILED = 3 };    // (written by a host lane -- runtime.cpp -- that could not place the frame's records)

// ---- atomics: agent scope on the GPU (coherent across the XCDs' L2s); the host simulation runs one lane at a time ----
#if defined( __HIP_DEVICE_COMPILE__ )
#define AA_AT_ADD( p, v ) __hip_atomic_fetch_add( ( p ), ( v ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT )
#define AA_AT_ADD_ACQ( p, v ) __hip_atomic_fetch_add( ( p ), ( v ), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT )
#define AA_AT_ADD_REL( p, v ) __hip_atomic_fetch_add( ( p ), ( v ), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT )
#define AA_AT_LOAD( p ) __hip_atomic_load( ( p ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT )
#define AA_AT_STORE( p, v ) __hip_atomic_store( ( p ), ( v ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT )
#define AA_NOW() wall_clock64()                        /* constant 100 MHz */
#else
template <class T, class V> AA_HD inline T aa_host_add( T * p, V v ) { const T old = *p; *p = static_cast<T>( old + static_cast<T>( v ) ); return old; }
#define AA_AT_ADD( p, v ) aa::aa_host_add( ( p ), ( v ) )
#define AA_AT_ADD_ACQ( p, v ) aa::aa_host_add( ( p ), ( v ) )
#define AA_AT_ADD_REL( p, v ) aa::aa_host_add( ( p ), ( v ) )
#define AA_AT_LOAD( p ) ( *( p ) )
#define AA_AT_STORE( p, v ) ( *( p ) = ( v ) )
AA_HD inline unsigned long long aa_host_clock() { static unsigned long long t = 0; return t += 1000; }
#define AA_NOW() aa::aa_host_clock()                  /* the simulation's clock: 10 us per look */
#endif

// Free coefficient chunks of the GPU: a ring of chunk numbers + a semaphore.  Consumers are token lanes (pool_take), producers
// the kernels that return the chunks of released frames or add the chunks of newly mapped heap (pool_push).  The ring has at
// least as many entries as there are chunks, so a producer never overwrites an entry that has not been handed out.
struct CoeffPool {
  int32_t avail;                // published entries nobody has claimed yet (claimed first, then a ticket is drawn)
  uint32_t head;                // next ticket
  uint32_t reserve, publish;    // producers: entries reserved / entries visible
  uint32_t mask;                // ring entries - 1 (a power of two)
  uint32_t starving;            // lanes that found the pool empty (events; the host grows the heap when it moves)
  uint32_t pad[2];
};
constexpr uint32_t kChunkBlocks = 2048;               // 64 KB
constexpr uint32_t kNoChunk = 0xFFFFFFFFu;
constexpr uint32_t kMbBlocks = 26;                    // what must be left in a chunk when a macroblock starts: 25 blocks + the pre-zeroed next one
// Packed storage: a chunk is 32768 16-bit words, a macroblock takes at most 25 x (mask word + 16 values)
constexpr uint32_t kChunkWords = kChunkBlocks * 16u;
constexpr uint32_t kMbMaskSlots = 25u;                // a macroblock's words start with one mask slot per block (coeff_pack.hh)
constexpr uint32_t kMbWords = kMbMaskSlots + 25u * 16u;
// entries of a frame's chunk list ([0] = count): a chunk that is left behind holds at least 80 macroblocks' worth of dense
// blocks ((2048 - 26 + 1) / 25) or 77 of packed words, whatever the content -- one bound for both formats
// (lanes > 1: one lane per partition -- every lane leaves a partly filled chunk behind)
AA_HD constexpr uint32_t chunk_list_entries( uint32_t nmb, uint32_t lanes = 1 ) { return 2u + ( nmb + 75u ) / 76u + 2u * ( lanes - 1u ); }

// Jobs waiting for a token lane: slots[ticket & mask] = the ParseJob; tickets below `publish` are ready, `head` is the next
// one to take.  Producers (k_enqueue_jobs) reserve a range, fill it, publish in order.  The host keeps fewer jobs in flight
// than the ring has slots.
struct TokQueue {
  uint32_t head, reserve, publish, mask;
};

// What every worker lane of a GPU shares (kernel arguments: uniform)
struct Heap {
  AA_GLOBAL int16_t * base;     // the coefficient heap: block i at base + 16 * i
  AA_GLOBAL CoeffPool * pool;
  AA_GLOBAL uint32_t * ring;
};

// One frame to parse on the device.  Built by the host header pre-pass, resident in HBM.
struct alignas( 16 ) ParseJob {
  FrameParams fp;
  const uint8_t * data;         // compressed frame, 16-byte aligned, readable up to data_padded
  uint32_t size, data_padded;   // data_padded: multiple of 16, >= size
  uint32_t nmb, flags_padded;   // flags_padded: multiple of 16, >= nmb
  aa_mb_info * mbs;
  uint32_t * chunk_list;        // [chunk_list_entries( nmb )]: [0] = how many coefficient chunks the frame took, then their numbers
  uint32_t mp_stride;           // != 0: one lane per partition may be used; behind the flags in raster order (mbflags[0, flags_padded)) there is
  uint32_t mp_pad;              //       then a second copy laid out partition by partition: partition p's rows back to back at
                                //       mbflags + flags_padded + p * mp_stride (mp_stride: a multiple of 16; mp_flag_index)
  uint32_t * packed_pos;        // [nmb], packed storage only: where a macroblock's words start = ordinal of the chunk in chunk_list << 15 | word in the chunk
  unsigned long long * intra_rows;
  uint8_t * mbflags;            // [flags_padded]: INTER | HAS_Y2 | SKIP of every macroblock, header kernel -> token kernel
  FrameSummary * summary;
};

// where the header pass puts the second copy of the flags byte of macroblock (col, row), if the job asks for one (mp_stride != 0)
AA_HD inline uint32_t mp_flag_index( const ParseJob & J, uint32_t row, uint32_t col )
{
  const uint32_t P = J.fp.nparts;
  return J.flags_padded + ( row % P ) * J.mp_stride + ( row / P ) * J.fp.mbw + col;
}
// bytes of a job's flags: raster copy + (mp) one padded run per partition; stride for a frame of mbw x mbh with P partitions
AA_HD constexpr uint32_t mp_flag_stride( uint32_t mbw, uint32_t mbh, uint32_t P ) { return ( ( ( mbh + P - 1u ) / P ) * mbw + 15u ) & ~15u; }

namespace tok {

// Ring sizes.  Invariant of the stream ring: lead = wpos - rpos >= kPeriod at every top-up (a period consumes at most kPeriod
// bytes).  A top-up asks for kPeriod more bytes iff lead <= kRing - kPeriod (they land a period later, when the bytes they
// replace have been read); otherwise lead > kRing - kPeriod >= 2 * kPeriod leaves >= kPeriod after the period.  So
// kRing >= 3 * kPeriod is enough -- 128 bytes of LDS per lane with a period of 32 steps (it was 256 / 64: the slice is what
// bounds the number of chains a CU holds).
constexpr uint32_t kPeriod = 32;          // steps between ring top-ups (a step consumes at most one stream byte)
constexpr uint32_t kRing = 128;           // bytes in the stream ring
constexpr uint32_t kChunks = 2;           // 16-byte chunks fetched per top-up (= kPeriod bytes)
constexpr uint32_t kMetaRing = 32;        // macroblock flags in the flag ring
constexpr uint32_t kMetaChunks = 1;       // 16-flag chunks fetched per top-up (only runs of skipped macroblocks use more: they wait)
static_assert( kRing >= 3 * kPeriod && 16 * kChunks == kPeriod && ( kRing & ( kRing - 1 ) ) == 0, "stream ring invariant" );

// Workgroup LDS ("smem"; one flat buffer on the host): the node and block tables and the constant probabilities at offset
// 0 -- so that a node record's address is a plain number a record can carry --, then one 128-byte STREAM RING per lane (a region of
// its own, 128-byte aligned: the step forms a ring address with one and-or), then one slice per lane.  The number of chains a CU
// holds is what LDS is left (160 KB / (ring + slice)), so a slice carries nothing that could be shared or left out.
constexpr uint32_t kNodeTabOff = 0;       // node records, 8 bytes each: 47 nodes (addresses < 512) ...
constexpr uint32_t kBandTabOff = 384;     // band33[position 0..17]: 33 * coefficient band of a position (a multiple of 32: Lane::ia)
constexpr uint32_t kXtab = 408;           // extra-bit probabilities of the six categories, then the sign's 128 (absolute address)
constexpr uint32_t kSignX = 26;           // index of the sign's probability in that table
constexpr uint32_t kBlockTabOff = 440;    // block entries, 8 bytes each (26)
constexpr uint32_t kIdleTabOff = 648;     // ... and 5 IDLE records (addresses >= 512: bit 9 says "this lane is not decoding")
constexpr uint32_t kTablesBytes = 768;    // first stream ring (a multiple of 128)
AA_HD constexpr uint32_t ring_addr( uint32_t lane ) { return kTablesBytes + lane * 128u; }
// a lane's slice: behind the rings of all `lanes` lanes of the workgroup; lane_bytes = lane_lds_bytes() (ring included)
AA_HD constexpr uint32_t slice_addr( uint32_t lane, uint32_t lanes, uint32_t lane_bytes ) { return kTablesBytes + lanes * 128u + lane * ( lane_bytes - 128u ); }
// offsets relative to a lane's slice:
constexpr uint32_t kMeta = 0;             // flag ring (16-byte aligned: filled 16 bytes at a time)
// Token probabilities: THREE of the frame's four 264-byte type planes ([8][3][11] each), not the whole [4][8][3][11] table.  The
// order of block types inside a macroblock is fixed (macroblock.cc:475-502): Y2, 16 x Y_AFTER_Y2, 8 x UV -- or, for a macroblock
// without a Y2 block (B_PRED, SPLITMV), 16 x Y_WITHOUT_Y2, 8 x UV.  A macroblock therefore reads ONE of the two Y planes, and which
// one is known from its flags byte when it starts: the slice keeps the Y2 and UV planes for the whole frame and one Y plane
// (Lane::ykind), replaced from the job's table in HBM at the macroblock boundary where the kind changes (the slow path; inter
// frames are nearly all-Y2, the reference encoder's key frames all-B_PRED; a frame that mixes the kinds pays one 264-byte read
// per change).  264 bytes less per lane = 5 more chains per wave at 1080p.
constexpr uint32_t kPlaneBytes = 264;
constexpr uint32_t kPlaneY = kMeta + kMetaRing;     // Y_AFTER_Y2 or Y_WITHOUT_Y2
constexpr uint32_t kPlaneUV = kPlaneY + kPlaneBytes;
constexpr uint32_t kPlaneY2 = kPlaneUV + kPlaneBytes;
constexpr uint32_t kAbove = kPlaneY2 + kPlaneBytes; // the above-row non-zero flags, 9 bits per macroblock column:
//   a lane of its own (SH = false): one BYTE per column (4 Y, 2 U, 2 V) followed by one BIT per column (Y2) -- 135 bytes at 1080p
//   lanes of one frame sharing the array (one lane per partition, SH = true): uint16 per column -- two lanes may be in columns of
//   the same byte of a bit array in the same instruction, and a read-modify-write would lose one of the bits
static_assert( kPlaneY % 8 == 0 && kPlaneUV % 8 == 0 && kPlaneY2 % 8 == 0 && kPlaneY2 / 8 < 128, "plane offsets travel as bytes / 8 in the block table" );
AA_HD constexpr uint32_t above_bytes( uint32_t mbw, bool shared ) { return shared ? 2u * mbw : mbw + ( mbw + 7u ) / 8u; }
// then, only for frames with more than one token partition: 8 saved partition decoders x 16 bytes
AA_HD constexpr uint32_t part_off( uint32_t mbw, bool shared ) { return ( kAbove + above_bytes( mbw, shared ) + 15 ) & ~15u; }
// (AA_SLICE_SKEW: a slice of an ODD number of 16-byte units -- consecutive lanes' slices then start 4 * odd banks apart (8 different bank
// phases) instead of 0 / 16 banks: lanes that read the same probability of their frames hit the same bank 4 ways instead of 15)
AA_HD constexpr uint32_t slice_skew( uint32_t slice ) { return AA_SLICE_SKEW && ( slice / 16u ) % 2u == 0u ? 16u : 0u; }
AA_HD constexpr uint32_t lane_lds_bytes( uint32_t mbw, bool multi_partition, bool shared = false )     // ring + slice
{
  return kRing + part_off( mbw, shared ) + ( multi_partition ? 128u : 0u ) + slice_skew( part_off( mbw, shared ) + ( multi_partition ? 128u : 0u ) );
}
// (The flags were tried in HBM -- 240 bytes of LDS per lane at 1080p would buy 15 % more chains per CU --, the lane keeping the
// eight columns it passes in registers.  Measured on MI355X, round 3: every use of those registers costs the wave an
// s_waitcnt vmcnt(0), i.e. a drain of ALL its outstanding coefficient stores at every macroblock boundary of every lane;
// steps went from 0.30 to 1.8 us with 26 lanes per wave.  The step must never wait for HBM: the flags stay in LDS.)

// dct_cat probabilities (tokens.cc:36-48) laid out back to back: cat1 @0, cat2 @1, cat3 @3, cat4 @6, cat5 @10, cat6 @15, sign @26
constexpr uint8_t kXtabInit[27] = { 159, 165, 145, 173, 148, 140, 176, 155, 140, 135, 180, 157, 141, 134, 130,
                                    254, 254, 243, 230, 196, 177, 153, 140, 133, 130, 129, 128 };

// ---- nodes ----------------------------------------------------------------------------------------------------------
// 0..10   the token tree (tokens.cc:73-124); node k reads probability k of the current (type, band, context) row
// 11..36  extra bits: cat1 = 11, cat2 = 12-13, cat3 = 14-16, cat4 = 17-20, cat5 = 21-25, cat6 = 26-36 (node e reads kXtab[e-11])
// 37..46  the sign, one node per token kind: the record knows the magnitude to add (DCT_1..4: the literal; dct_catN: its
//         base, the extra bits having been shifted into Lane::mag) and the context the token leaves behind
// A lane's state is the ADDRESS of its node's record (8 * node).  A lane that is not decoding -- waiting for the block-end pass,
// at a macroblock boundary, without a frame -- holds the address of an IDLE record instead, one per reason (R_*), and RUNS THE
// STEP LIKE EVERYBODY ELSE (round 6; until then the step was a predicated region, paid for with a compare, an exec-mask save /
// restore and two taken branches per step): an idle record leads back to itself and asks for nothing, and a lane whose record
// address has bit 9 set decodes with probability 256 -- split = range, the bit is 0, range, value and shift do not move
// (bool_decoder.hh:82-107 with split == range) -- so the step leaves an idle lane exactly as it found it, but for the stream
// byte it may still shift into its window (which is what the next real step would have done first).
enum : uint32_t { kNodes = 47, kIdleRecs = 5,
                  R_MBDONE = kIdleTabOff, R_MB = kIdleTabOff + 8, R_DONE = kIdleTabOff + 16,
                  R_PARK = kIdleTabOff + 24,      // one lane per partition: through, but its slice holds what lanes still running share
                  R_BEND = kIdleTabOff + 32 };    // the block in progress has ended: the lane waits for the wave's next block-end pass (tok::block_end)
static_assert( kIdleTabOff >= 512 && R_BEND + 8 <= kTablesBytes && kNodes * 8 <= kBandTabOff, "node addresses below 512, idle records from 512 on" );
// half of a node record = what a decoded 0 / 1 at that node means:
//   [0,10) address of the next node's record (an EOB leads straight to R_BEND)   [10,15) index of the next node's probability
//   [15] shift the bit into the magnitude   [16] on to the next coefficient position   [17] emit the coefficient
//   [19,24) 11 * context the token leaves behind   [24,31) magnitude to add at emission   [31] ... probability in the current row (else in kXtab)
constexpr uint32_t H_ROWREL = 1u << 31, H_XS = 1u << 15, H_ADV = 1u << 16, H_EMIT = 1u << 17;
constexpr uint32_t kZeroX = 27;           // kXtab[27] = 0: what a record that reads no probability points at
constexpr uint32_t half_at( uint32_t next_addr, uint32_t pk, uint32_t flags, uint32_t ctx = 0, uint32_t addv = 0 )
{
  return next_addr | ( pk << 10 ) | flags | ( ( ctx * 11 ) << 19 ) | ( addv << 24 );
}
constexpr uint32_t half( uint32_t next, uint32_t pk, uint32_t flags, uint32_t ctx = 0, uint32_t addv = 0 ) { return half_at( next * 8, pk, flags, ctx, addv ); }
constexpr uint32_t tree( uint32_t k ) { return half( k, k, H_ROWREL ); }                       // on to tree node k
constexpr uint32_t sign_node( uint32_t addv )   // DCT_1..4 -> 37..40, dct_cat1..6 (bases 5,7,11,19,35,67) -> 41..46
{
  return addv <= 4 ? 36 + addv : ( addv == 5 ? 41 : addv == 7 ? 42 : addv == 11 ? 43 : addv == 19 ? 44 : addv == 35 ? 45 : 46 );
}
constexpr uint32_t to_sign( uint32_t addv, uint32_t flags = 0 ) { return half( sign_node( addv ), kSignX, flags ); }
constexpr uint32_t to_extra( uint32_t e ) { return half( e, e - 11, 0 ); }
struct NodeTable { V8 n[kNodes]; };
constexpr NodeTable make_nodes()
{
  NodeTable t {};
  t.n[0] = { half_at( R_BEND, kZeroX, 0 ), tree( 1 ) };     // EOB: the block has ended
  t.n[1] = { half( 1, 1, H_ROWREL | H_ADV ), tree( 2 ) };   // a ZERO token: node 1 of the next position (no EOB check), context 0
  t.n[2] = { to_sign( 1 ), tree( 3 ) };
  t.n[3] = { tree( 4 ), tree( 6 ) };
  t.n[4] = { to_sign( 2 ), tree( 5 ) };
  t.n[5] = { to_sign( 3 ), to_sign( 4 ) };
  t.n[6] = { tree( 7 ), tree( 8 ) };
  t.n[7] = { to_extra( 11 ), to_extra( 12 ) };
  t.n[8] = { tree( 9 ), tree( 10 ) };
  t.n[9] = { to_extra( 14 ), to_extra( 17 ) };
  t.n[10] = { to_extra( 21 ), to_extra( 26 ) };
  const uint32_t first[6] = { 11, 12, 14, 17, 21, 26 }, last[6] = { 11, 13, 16, 20, 25, 36 }, base[6] = { 5, 7, 11, 19, 35, 67 };
  for ( uint32_t c = 0; c < 6; c++ )
    for ( uint32_t e = first[c]; e <= last[c]; e++ ) {
      const uint32_t h = e == last[c] ? to_sign( base[c], H_XS ) : ( to_extra( e + 1 ) | H_XS );
      t.n[e] = { h, h };
    }
  const uint32_t addvs[10] = { 1, 2, 3, 4, 5, 7, 11, 19, 35, 67 };
  for ( uint32_t k = 0; k < 10; k++ ) {                     // the sign: emit, then the EOB check of the next position
    const uint32_t h = half( 0, 0, H_ROWREL | H_ADV | H_EMIT, addvs[k] == 1 ? 1 : 2, addvs[k] );
    t.n[37 + k] = { h, h };
  }
  return t;
}
constexpr NodeTable kNodeTable = make_nodes();
// idle record k (address kIdleTabOff + 8 k): both halves lead back to it, read no probability, ask for nothing
constexpr uint32_t idle_half( uint32_t k ) { return half_at( kIdleTabOff + 8 * k, kZeroX, 0 ); }

// ---- blocks ---------------------------------------------------------------------------------------------------------
// parse order within a macroblock (macroblock.cc:480-500): 0 = Y2, 1..16 = Y, 17..20 = U, 21..24 = V.  Per block: where its
// "above" / "left" non-zero flags live in Lane::ctxbits (above: bits 0-8 = 4 Y columns, 2 U, 2 V, Y2; left: bits 16-24),
// the plane of probabilities it reads (byte 2: offset of the plane in the lane's slice / 8, + kBlkIsY for a Y block, whose first
// position depends on the macroblock having a Y2), its bit in nz_mask.  Word 1 = the two flag bits as a mask: the block's context
// is the number of bits of ctxbits under it.
constexpr uint32_t kBlkIsY = 128;
struct BlockTable { V8 b[26]; };
constexpr BlockTable make_blocks()
{
  BlockTable t {};
  for ( uint32_t blk = 0; blk < 25; blk++ ) {
    uint32_t a = 8, l = 8, sel = kPlaneY2 / 8, bit = 24;            // Y2
    if ( blk >= 1 && blk <= 16 ) { const uint32_t b = blk - 1; a = b & 3; l = b >> 2; sel = kPlaneY / 8 + kBlkIsY; bit = b; }
    else if ( blk >= 17 ) { const uint32_t k = blk - 17, pl = k >> 2; a = 4 + 2 * pl + ( k & 1 ); l = 4 + 2 * pl + ( ( k >> 1 ) & 1 ); sel = kPlaneUV / 8; bit = blk - 1; }
    t.b[blk] = { a | ( ( 16 + l ) << 8 ) | ( sel << 16 ) | ( bit << 24 ), ( 1u << a ) | ( 1u << ( 16 + l ) ) };
  }
  t.b[25] = { 0, 0 };
  return t;
}
constexpr BlockTable kBlockTable = make_blocks();
static_assert( kBandTabOff % 32 == 0 && kBandTabOff + 18 <= kXtab && kXtab + 28 <= kBlockTabOff && kBlockTabOff % 8 == 0 && kBlockTabOff + sizeof( BlockTable ) <= kIdleTabOff
               && kTablesBytes % 128 == 0, "tables overlap" );

constexpr uint64_t nib( std::initializer_list<unsigned> v ) { uint64_t r = 0; unsigned i = 0; for ( unsigned x : v ) r |= static_cast<uint64_t>( x ) << ( 4 * i++ ); return r; }
constexpr uint64_t kZigzagNib = nib( { 0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15 } );
constexpr uint64_t kBandNib = nib( { 0, 1, 2, 3, 6, 4, 5, 6, 6, 6, 6, 6, 6, 6, 6, 7 } );   // coefficient band of position 0..15 (16: 0)

// What a lane needs of its ParseJob at every step, held in registers (the job itself stays in HBM and is only consulted on
// the rare paths: partition switches, the end of the frame).
struct Frame {
  const AA_GLOBAL ParseJob * job;
  const AA_GLOBAL uint8_t * data;
  const AA_GLOBAL uint8_t * mbflags;
  AA_GLOBAL aa_mb_info * mbs;
  AA_GLOBAL uint32_t * chunk_list;
  AA_GLOBAL uint32_t * packed_pos;
  uint32_t data_padded, flags_padded, nmb, mbw, nparts;
  uint32_t max_steps;           // no frame of this size can take more steps: a lane that gets there stops (never a hung GPU)
  // one lane per partition (mp_P > 1; then nmb / mbflags / flags_padded are this lane's rows only):
  uint32_t mp_P, mp_p;          // lanes of the frame, this lane's partition
  uint32_t mp_owner;            // LDS offset of the slice that holds what the frame's lanes share (above-row flags, MpShared)
};
AA_HD inline Frame frame_of( const ParseJob * job )
{
  Frame F;
  F.job = (const AA_GLOBAL ParseJob *) job;
  F.data = (const AA_GLOBAL uint8_t *) job->data; F.mbflags = (const AA_GLOBAL uint8_t *) job->mbflags;
  F.mbs = (AA_GLOBAL aa_mb_info *) job->mbs; F.chunk_list = (AA_GLOBAL uint32_t *) job->chunk_list;
  F.packed_pos = (AA_GLOBAL uint32_t *) job->packed_pos;
  // per macroblock at most 25 blocks x (16 tokens x (11 tree nodes + 11 extra bits + sign) + the wait for the block-end pass),
  // plus boundary steps
  const uint64_t bound = static_cast<uint64_t>( job->nmb ) * ( 25u * ( 16u * 23u + 8u ) + 8u ) + 4096u;
  F.max_steps = bound > 0xFFFFFFF0ull ? 0xFFFFFFF0u : static_cast<uint32_t>( bound );
  F.data_padded = job->data_padded; F.flags_padded = job->flags_padded; F.nmb = job->nmb; F.mbw = job->fp.mbw; F.nparts = job->fp.nparts;
  F.mp_P = 1; F.mp_p = 0; F.mp_owner = 0;
  return F;
}
// ... as lane `p` of the frame's P = nparts lanes sees it: its rows p, p + P, ... are its frame
AA_HD inline Frame frame_of_partition( const ParseJob * job, uint32_t p, uint32_t owner_base )
{
  Frame F = frame_of( job );
  const uint32_t P = job->fp.nparts, mbh = job->fp.mbh;
  const uint32_t rows = mbh > p ? ( mbh - p + P - 1u ) / P : 0u;
  F.mbflags = (const AA_GLOBAL uint8_t *) job->mbflags + job->flags_padded + p * job->mp_stride;
  F.nmb = rows * F.mbw;
  F.flags_padded = job->mp_stride;
  F.mp_P = P; F.mp_p = p; F.mp_owner = owner_base;
  return F;
}
// the partition whose lane finishes last (it has the frame's last row): its slice holds what the lanes share
AA_HD inline uint32_t mp_owner_partition( const ParseJob * job ) { return ( job->fp.mbh - 1u ) % job->fp.nparts; }

// How a wave deals its idle lanes out to the tickets it has just drawn (one lane per partition): ticket t = 0 .. got - 1 is a
// frame of parts_of( t ) partitions.  In ticket order, a frame gets a lane per partition if the wave has that many idle lanes to
// spare beyond one per ticket, else one lane as ever; lanes go out in rank order (rank = position among the wave's idle lanes).
// -> what the idle lane of rank `rank` does: any = false: nothing (it stays idle); else partition `part` of ticket `ticket`,
// whose n lanes are the ranks start .. start + n - 1.  Every lane of the wave evaluates this with the same n_idle / got / parts.
struct MpDeal { uint32_t ticket, part, n, start; bool any; };
template <class PartsOf>
AA_HD inline MpDeal mp_deal( uint32_t n_idle, uint32_t got, uint32_t rank, PartsOf parts_of )
{
  MpDeal d { 0, 0, 1, 0, false };
  uint32_t spare = n_idle - got, start = 0;
  for ( uint32_t t = 0; t < got; t++ ) {
    const uint32_t P = parts_of( t );
    const uint32_t n = ( P > 1u && P - 1u <= spare ) ? P : 1u;
    spare -= n - 1u;
    if ( rank >= start && rank < start + n ) { d.ticket = t; d.part = rank - start; d.n = n; d.start = start; d.any = true; }
    start += n;
  }
  return d;
}

// What the lanes of a frame share (one lane per partition), in the owner's slice where a single lane keeps its saved partition
// decoders (part_off): all zero at the start except `left`.
struct MpShared {
  uint32_t prog[8];             // [p]: macroblocks lane p has completed (in ITS sequence: whole rows * mbw + columns of the current one)
  uint32_t nchunks;             // chunks in the frame's list so far
  uint32_t left;                // lanes that have not finished
  uint32_t blocks, words, steps;// sums over the lanes that have finished
  uint32_t status;              // != TOK_OK: a lane gave up (no memory, step bound): the others stop at their next macroblock boundary
};
static_assert( sizeof( MpShared ) <= 128, "MpShared lives in the partition save area" );

struct Chunk16 { uint32_t w[4]; };

AA_HD inline Chunk16 load16( const AA_GLOBAL uint8_t * p )          // 16-byte aligned
{
  const AA_GLOBAL V16 * q = (const AA_GLOBAL V16 *) p;
  const V16 v = *q;
  Chunk16 c; c.w[0] = v.x; c.w[1] = v.y; c.w[2] = v.z; c.w[3] = v.w;
  return c;
}
AA_HD inline void lds_store16( uint8_t * smem, uint32_t off, const Chunk16 & c )
{
  V16 v; v.x = c.w[0]; v.y = c.w[1]; v.z = c.w[2]; v.w = c.w[3];
  *reinterpret_cast<V16 *>( smem + off ) = v;
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

// All LDS addresses in a Lane are offsets into the workgroup's smem (the lane's slice starts at `base`).
struct Lane {
  uint32_t base;                  // offset of this lane's slice
  uint32_t sbase;                 // ... of its stream ring (128-byte aligned)
  // boolean decoder of the current partition: 32-bit window; sh = 16 - (valid bits below the 8 being compared)
  uint32_t value, range;
  int32_t sh;
  uint32_t rpos, rend;            // next stream byte to shift in / end of the partition (offsets into the frame)
  uint32_t wpos;                  // stream ring holds [wpos - kRing, wpos)
  uint32_t mwpos;                 // flag ring holds macroblocks [mwpos - kMetaRing, mwpos)
  uint32_t pend_wpos, pend_mwpos; // what the chunks in flight are for (kNoPend: nothing in flight)
  Chunk16 pend[kChunks], mpend[kMetaChunks];
  // token in progress
  uint32_t rec;                   // address of the record of the node about to be decoded (>= R_MBDONE: an idle record -- not decoding)
#if AA_STEP_PRELOAD
  // what the next step reads from LDS, asked for ahead of time (tok::preload): the probability at paddr, the stream byte at rpos, the
  // node record at rec, the band of the position after ia.  Whoever changes one of those four outside the step asks again.
  uint32_t pre_prob, pre_raw, pre_band;
  V8 pre_rec;
#endif
  uint32_t paddr;                 // address of its probability
  uint32_t rowaddr, typeaddr;     // addresses of the current probability row / of this block type's probabilities
  uint32_t ia;                    // kBandTabOff + coefficient position (a multiple of 32 + position: shifts and `& 16` see the position)
  uint32_t mag, nonzero;
  // block in progress
  uint32_t blkaddr;               // address of the BlockTable entry of the block AFTER the current one
  uint32_t nzsel, blkbit;
  uint32_t blkslot;               // number of the block in progress in nz_mask's numbering (packed storage: its mask slot)
  // macroblock in progress
  uint32_t ctxbits;               // non-zero flags: above (this column) bits 0-8, left bits 16-24
  uint32_t flags, nz_mask, mb_first, coeff_blocks, yfirst;
  uint32_t ykind;                 // which Y plane the slice holds (Y_AFTER_Y2 / Y_WITHOUT_Y2; kNoPlane: none yet)
  AA_GLOBAL int16_t * blk;        // the coefficient block being filled = Heap::base + 16 * blk_index (zeroed in advance)
  uint32_t blk_index;             // its index in the heap
  uint32_t blk_left;              // blocks left in the chunk being filled, the current one included (0: no chunk yet)
  uint32_t nchunks;               // chunks taken so far
  unsigned long long mem_since;   // waiting for a chunk since (0: not waiting)
  // packed storage only (then blk = where the next coefficient value goes, blk_index = first block of the chunk being filled):
  AA_GLOBAL int16_t * sink;       // (packed) where blk points while the lane has no chunk: a word nobody reads (see step: the store is unconditional)
  AA_GLOBAL int16_t * hdr;        // the macroblock's first word = mask slot 0 (a block that ends non-zero writes its slot); blk = the next value
  uint32_t zzmask;                // zigzag positions of the block in progress that hold a coefficient
  uint32_t words;                 // words used in the chunks left behind
  uint32_t mi_real;               // one lane per partition: index of the current macroblock's record (mi counts this lane's macroblocks)
  uint32_t chunk_ord;             // ordinal, in the frame's chunk list, of the chunk being filled
  // position
  uint32_t mi, col, row, part;
  uint32_t steps;
};
constexpr uint32_t kNoPend = 0xFFFFFFFFu;
constexpr uint32_t kNoPlane = 0xFFFFFFFFu;

AA_HD inline void zero_slot( AA_GLOBAL int16_t * block )
{
  V16 z; z.x = z.y = z.z = z.w = 0;
  AA_GLOBAL V16 * p = (AA_GLOBAL V16 *) block;
  p[0] = z; p[1] = z;
}

// ---- coefficient chunks ---------------------------------------------------------------------------------------------
// -> a free chunk's number, or kNoChunk when the pool is empty right now
AA_HD inline uint32_t pool_take( const Heap & H )
{
  // claim first (the semaphore counts published entries), then draw the ticket: a ticket is only ever drawn for an entry
  // that is there.  No acquire fence: the entry is read with an agent-scope atomic load (it comes from the coherence point,
  // where the producer's atomic store landed before it raised the semaphore), and the chunk itself is only ever WRITTEN by
  // this lane -- an acquire here would invalidate the CU's vector cache, and the XCD's L2 lines, once per 64 KB per lane.
  const int32_t old = AA_AT_ADD( &H.pool->avail, -1 );
  if ( old <= 0 ) { AA_AT_ADD( &H.pool->avail, 1 ); return kNoChunk; }
  const uint32_t t = AA_AT_ADD( &H.pool->head, 1u );
  return AA_AT_LOAD( &H.ring[t & H.pool->mask] );
}
// producers (one thread per call): n chunk numbers ids[0..n) (or first, first+1, ... when ids is null) go back to the ring
AA_HD inline void pool_push( const Heap & H, const AA_GLOBAL uint32_t * ids, uint32_t first, uint32_t n )
{
  if ( !n ) return;
  const uint32_t b = AA_AT_ADD( &H.pool->reserve, n );
  for ( uint32_t i = 0; i < n; i++ ) AA_AT_STORE( &H.ring[( b + i ) & H.pool->mask], ids ? ids[i] : first + i );
  while ( AA_AT_LOAD( &H.pool->publish ) != b ) {}          // in order (whoever reserved before us is running: it reserved)
  AA_AT_STORE( &H.pool->publish, b + n );
  AA_AT_ADD_REL( &H.pool->avail, static_cast<int32_t>( n ) );
}

// ---- stream ring ----------------------------------------------------------------------------------------------------
// synchronous (re)fill of the whole stream ring around rpos: start of a partition
AA_HD inline void prime_stream( Lane & L, uint8_t * smem, const Frame & J )
{
  const uint32_t base = L.rpos & ~15u;
  for ( uint32_t k = 0; k < kRing / 16; k++ ) {
    const uint32_t at = base + 16 * k;
    Chunk16 c; c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0;
    if ( at < J.data_padded ) c = load16( J.data + at );
    lds_store16( smem, L.sbase + ( at & ( kRing - 1 ) ), mask_past_end( c, at, L.rend ) );
  }
  L.wpos = base + kRing;
  L.pend_wpos = kNoPend;
}

// decoder of partition `p` at its first bit (BoolDecoder ctor, bool_decoder.hh:45-54)
AA_HD inline void start_partition( Lane & L, uint8_t * smem, const Frame & J, uint32_t p )
{
  L.part = p;
  L.rpos = J.job->fp.part_off[p];
  L.rend = J.job->fp.part_off[p] + J.job->fp.part_size[p];
  prime_stream( L, smem, J );
  uint32_t v = 0;
  for ( int k = 0; k < 4; k++ ) { v = ( v << 8 ) | smem[L.sbase + ( L.rpos & ( kRing - 1 ) )]; L.rpos++; }
  L.value = v; L.sh = -8; L.range = 255;
}

template <bool SH = false>
AA_HD inline void switch_partition( Lane & L, uint8_t * smem, const Frame & J, uint32_t p )
{
  const uint32_t save = L.base + part_off( J.mbw, SH );
  uint32_t * s = reinterpret_cast<uint32_t *>( smem + save + 16 * L.part );
  s[0] = L.value; s[1] = L.range | ( static_cast<uint32_t>( L.sh + 64 ) << 8 ); s[2] = L.rpos; s[3] = 1;
  const uint32_t * t = reinterpret_cast<const uint32_t *>( smem + save + 16 * p );
  if ( !t[3] ) { start_partition( L, smem, J, p ); return; }
  L.part = p;
  L.value = t[0]; L.range = t[1] & 255u; L.sh = static_cast<int32_t>( t[1] >> 8 ) - 64; L.rpos = t[2];
  L.rend = J.job->fp.part_off[p] + J.job->fp.part_size[p];
  prime_stream( L, smem, J );
}

// ---- every kPeriod steps, all lanes together: land the chunks requested a period ago, request the next ---------------
template <bool MP = false>
AA_HD inline void top_up( Lane & L, uint8_t * smem, const Frame & J )
{
  if ( L.rec == R_DONE || ( MP && L.rec == R_PARK ) ) return;
  if ( L.pend_wpos == L.wpos ) {
    for ( uint32_t k = 0; k < kChunks; k++ )
      lds_store16( smem, L.sbase + ( ( L.wpos + 16 * k ) & ( kRing - 1 ) ), mask_past_end( L.pend[k], L.wpos + 16 * k, L.rend ) );
    L.wpos += 16 * kChunks;
  }
  if ( L.pend_mwpos == L.mwpos ) {
    for ( uint32_t k = 0; k < kMetaChunks; k++ ) lds_store16( smem, L.base + kMeta + ( ( L.mwpos + 16 * k ) & ( kMetaRing - 1 ) ), L.mpend[k] );
    L.mwpos += 16 * kMetaChunks;
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
  if ( L.mwpos - L.mi <= kMetaRing - 16 * kMetaChunks ) {
    for ( uint32_t k = 0; k < kMetaChunks; k++ ) {
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
#define AA_ANY( x ) ( __builtin_amdgcn_ballot_w64( x ) != 0 )        // wave-uniform: does any lane ...
// (lanes of one wave add to the same LDS word in the same instruction: an LDS atomic, not a read-modify-write)
#define AA_LDS_ADD( smem, off, v ) __hip_atomic_fetch_add( aa::tok::lds_at<uint32_t>( ( smem ), ( off ) ), ( v ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP )
// (a word another lane of the wave writes: read it from LDS every time, never out of a register)
#define AA_LDS_LOAD( smem, off ) __hip_atomic_load( aa::tok::lds_at<uint32_t>( ( smem ), ( off ) ), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP )
#define AA_MUL24( a, b ) __umul24( ( a ), ( b ) )
#define AA_POPC( v ) static_cast<uint32_t>( __builtin_popcount( v ) )
#define AA_UBFE( v, off, width ) __builtin_amdgcn_ubfe( ( v ), ( off ), ( width ) )
#define AA_BFI( m, a, b ) ( ( ( m ) & ( a ) ) | ( ~( m ) & ( b ) ) )       /* v_bfi_b32 */
#define AA_SBFE( v, off, width ) __builtin_amdgcn_sbfe( static_cast<int>( v ), ( off ), ( width ) )      /* sign-extended field: a 1-bit field gives 0 / -1 */
#define AA_CLZ( v ) __builtin_clz( v )                                      /* v_ffbh_u32: 32 for 0 on the GPU (idle lanes' garbage) */
// The workgroup's dynamic LDS starts at LDS address 0 (the kernel has no static LDS), so an offset into smem IS the LDS
// address: form the pointer from the number and spare the hot loop one "add the base symbol" per access.
template <class T> __device__ inline __attribute__( ( address_space( 3 ) ) ) T * lds_at( uint8_t *, uint32_t off )
{
  return (__attribute__( ( address_space( 3 ) ) ) T *) static_cast<uintptr_t>( off );
}
#else
#define AA_ANY( x ) ( x )
#define AA_LDS_ADD( smem, off, v ) aa::aa_host_add( aa::tok::lds_at<uint32_t>( ( smem ), ( off ) ), ( v ) )
#define AA_LDS_LOAD( smem, off ) ( *aa::tok::lds_at<uint32_t>( ( smem ), ( off ) ) )
#define AA_MUL24( a, b ) ( ( a ) * ( b ) )
#define AA_POPC( v ) static_cast<uint32_t>( __builtin_popcount( v ) )
#define AA_UBFE( v, off, width ) ( ( ( v ) >> ( off ) ) & ( ( 1u << ( width ) ) - 1u ) )
#define AA_BFI( m, a, b ) ( ( ( m ) & ( a ) ) | ( ~( m ) & ( b ) ) )
#define AA_SBFE( v, off, width ) ( static_cast<int32_t>( static_cast<uint32_t>( v ) << ( 32 - ( off ) - ( width ) ) ) >> ( 32 - ( width ) ) )
#define AA_CLZ( v ) ( ( v ) ? __builtin_clz( v ) : 32 )
template <class T> inline T * lds_at( uint8_t * smem, uint32_t off ) { return reinterpret_cast<T *>( smem + off ); }
#endif

AA_HD inline void store_mb( const Frame & J, uint32_t mi, uint32_t nz_mask, uint32_t coeff_index, uint32_t flags )
{
  AA_GLOBAL aa_mb_info * mb = J.mbs + mi;
  mb->nz_mask = nz_mask;
  mb->coeff_index = coeff_index;
  mb->flags = static_cast<uint8_t>( flags );
}

// packed storage: where the macroblock's words start -- as chunk ordinal + word in the chunk into packed_pos (the host's view: a
// frame's chunks copied back), and as a 40-bit offset in 16-bit words from the heap's base into the record itself (coeff_index = the
// low 32 bits, `reserved` = bits 32-39: what the reconstruction kernels follow -- one load, no chunk list)
AA_HD inline void store_mb_packed( const Frame & J, uint32_t mi, uint32_t nz_mask, uint32_t pos, uint32_t flags, unsigned long long word_off = 0 )
{
  AA_GLOBAL aa_mb_info * mb = J.mbs + mi;
  mb->nz_mask = nz_mask;
  mb->coeff_index = static_cast<uint32_t>( word_off );
  mb->flags = static_cast<uint8_t>( flags );
  mb->reserved = static_cast<uint8_t>( word_off >> 32 );
  J.packed_pos[mi] = pos;
}

// The four LDS reads of a step (addresses all known when the previous step ends), asked for AHEAD of time -- at the end of the previous
// step, in front of its coefficient store and bookkeeping, so that the round trip runs beside ~15 instructions instead of in front of
// the step (a lone wave per SIMD has nobody else to hide it behind).  AA_STEP_PRELOAD=0: the step reads at its top (A/B builds).
AA_HD inline void preload( Lane & L, uint8_t * smem )
{
#if AA_STEP_PRELOAD
  L.pre_prob = *lds_at<const uint8_t>( smem, L.paddr );
  L.pre_raw = *lds_at<const uint8_t>( smem, L.sbase | ( L.rpos & ( kRing - 1 ) ) );
  L.pre_rec = *lds_at<const V8>( smem, L.rec );
  L.pre_band = *lds_at<const uint8_t>( smem, L.ia + 1u );
#else
  (void) L; (void) smem;
#endif
}

// p + n words, n = 0 / 1, for a pointer into a coefficient chunk.  On the GPU a 32-bit add: a chunk is 64 KB and 64-KB aligned (the
// runtime checks the heap's base), so the low half of the address never carries into the high half -- a 64-bit add is two
// instructions on a path where every instruction is four cycles of every bool.
AA_HD inline AA_GLOBAL int16_t * bump_words( AA_GLOBAL int16_t * p, uint32_t n )
{
#if defined( __HIP_DEVICE_COMPILE__ )
  const unsigned long long a = reinterpret_cast<unsigned long long>( p );
  return reinterpret_cast<AA_GLOBAL int16_t *>( ( a & 0xFFFFFFFF00000000ull ) | static_cast<uint32_t>( static_cast<uint32_t>( a ) + 2u * n ) );
#else
  return p + n;
#endif
}

// Make block `blk` of the macroblock the current one: contexts from the non-zero flags, first probability row.
AA_HD inline void setup_block( Lane & L, const uint8_t * smem, uint32_t blk )
{
  const V8 e = *reinterpret_cast<const V8 *>( smem + kBlockTabOff + 8 * blk );
  const uint32_t sel = ( e.x >> 16 ) & 255u;
  L.blkaddr = kBlockTabOff + 8 * ( blk + 1 );
  L.nzsel = e.y;
  L.blkslot = e.x >> 24;
  L.blkbit = 1u << L.blkslot;
  const uint32_t ctx = AA_POPC( L.ctxbits & e.y );          // "above" flag + "left" flag
  L.typeaddr = L.base + 8u * ( sel & ( kBlkIsY - 1u ) );
  const uint32_t idx = ( sel & kBlkIsY ) ? L.yfirst : 0u;   // Y blocks after a Y2 start at position 1 (tokens.cc:61)
  L.ia = kBandTabOff + idx;
  L.rowaddr = L.typeaddr + idx * 33u + ctx * 11u;           // band of position 0 / 1 is 0 / 1
  L.rec = 0; L.paddr = L.rowaddr;
  L.nonzero = 0; L.mag = 0;
}

// The slow path, for lanes at a macroblock boundary (R_MBDONE: the step just completed one; R_MB: waiting for flags):
// take macroblocks until one has tokens (skipped ones are settled on the spot), the flag ring runs dry (try again later)
// or the frame ends.  Everything rare lives here: row ends, partition switches, the end of the frame.
// The lane is through with its frame: counts to the host, the chunk list closed, then -- behind a release that makes every
// record and coefficient this lane stored visible to the whole device -- the `done` word the host polls.
template <bool PK>
AA_HD inline void finish_frame( Lane & L, const Frame & J, const Heap & H, uint32_t status )
{
  J.chunk_list[0] = L.nchunks;
  AA_GLOBAL FrameSummary * sum = (AA_GLOBAL FrameSummary *) J.job->summary;
  sum->num_coeff_blocks = L.coeff_blocks;
  if constexpr ( PK ) sum->packed_words = L.words + ( L.nchunks ? static_cast<uint32_t>( L.blk - ( H.base + static_cast<size_t>( L.blk_index ) * 16 ) ) : 0u );
  else sum->packed_words = 0;
  sum->steps = L.steps;
  sum->num_chunks = L.nchunks;
  sum->status = status;
#if defined( __HIP_DEVICE_COMPILE__ )
  __builtin_amdgcn_fence( __ATOMIC_RELEASE, "" );               // system scope: HBM stores written back, host stores ordered
  asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );            // (the compiler may drop the wait behind the write-back)
  __hip_atomic_store( &sum->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
#else
  sum->done = 1u;
#endif
  L.rec = R_DONE;
  L.blk = L.sink;                 // (the chunk is the frame's from here on: this lane's steps must not touch it any more)
}

// One lane per partition: this lane is through with its rows (or gives up: status != TOK_OK, which the other lanes of the frame
// see at their next macroblock boundary).  Its counts go to the frame's sums; the lane that finishes LAST reports the frame.
template <bool PK>
AA_HD inline void finish_partition( Lane & L, uint8_t * smem, const Frame & J, const Heap & H, uint32_t status )
{
  const uint32_t sh = J.mp_owner + part_off( J.mbw, true );
  if ( status != TOK_OK ) *lds_at<uint32_t>( smem, sh + offsetof( MpShared, status ) ) = status;
  uint32_t words = 0;
  if constexpr ( PK ) words = L.words + ( L.nchunks ? static_cast<uint32_t>( L.blk - ( H.base + static_cast<size_t>( L.blk_index ) * 16 ) ) : 0u );
  AA_LDS_ADD( smem, sh + offsetof( MpShared, blocks ), L.coeff_blocks );
  AA_LDS_ADD( smem, sh + offsetof( MpShared, words ), words );
  AA_LDS_ADD( smem, sh + offsetof( MpShared, steps ), L.steps );
  L.blk = L.sink;                 // (as in finish_frame)
  const uint32_t left = AA_LDS_ADD( smem, sh + offsetof( MpShared, left ), 0xFFFFFFFFu );
  // (the owner finishes last when all goes well -- it has the last row; when lanes give up it may not: then it stays, parked,
  // until the others are through with what its slice holds)
  L.rec = ( left != 1u && L.base == J.mp_owner ) ? static_cast<uint32_t>( R_PARK ) : static_cast<uint32_t>( R_DONE );
  if ( left != 1u ) return;
  // the last one: every lane's sums are in (their adds were issued before their decrement)
  const MpShared fin = *lds_at<const MpShared>( smem, sh );
  J.chunk_list[0] = fin.nchunks;
  AA_GLOBAL FrameSummary * sum = (AA_GLOBAL FrameSummary *) J.job->summary;
  sum->num_coeff_blocks = fin.blocks;
  sum->packed_words = PK ? fin.words : 0u;
  sum->steps = fin.steps;
  sum->num_chunks = fin.nchunks;
  sum->status = fin.status;
#if defined( __HIP_DEVICE_COMPILE__ )
  __builtin_amdgcn_fence( __ATOMIC_RELEASE, "" );               // (this lane's release covers its wave-mates' stores too: the
  asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );            //  counters are the wave's)
  __hip_atomic_store( &sum->done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
#else
  sum->done = 1u;
#endif
}

// ---- the above-row non-zero flags of column `col` (9 bits: ctxbits bits 0-8); `abase` = address of the array -----------------
template <bool SH>
AA_HD inline uint32_t above_load( uint8_t * smem, uint32_t abase, uint32_t mbw, uint32_t col )
{
  if constexpr ( SH ) return *lds_at<const uint16_t>( smem, abase + 2u * col );
  else {
    const uint32_t yuv = *lds_at<const uint8_t>( smem, abase + col ), y2 = *lds_at<const uint8_t>( smem, abase + mbw + ( col >> 3 ) );
    return yuv | ( ( ( y2 >> ( col & 7u ) ) & 1u ) << 8 );
  }
}
template <bool SH>
AA_HD inline void above_store( uint8_t * smem, uint32_t abase, uint32_t mbw, uint32_t col, uint32_t bits )
{
  if constexpr ( SH ) *lds_at<uint16_t>( smem, abase + 2u * col ) = static_cast<uint16_t>( bits );
  else {
    *lds_at<uint8_t>( smem, abase + col ) = static_cast<uint8_t>( bits );
    const uint32_t at = abase + mbw + ( col >> 3 ), m = 1u << ( col & 7u );      // (this lane's own array: nobody else touches the byte)
    const uint32_t old = *lds_at<const uint8_t>( smem, at );
    *lds_at<uint8_t>( smem, at ) = static_cast<uint8_t>( ( bits & 0x100u ) ? old | m : old & ~m );
  }
}

// plane `type` of the frame's token probabilities -> the slice's plane at `off` (66 words; the job's table is in HBM)
AA_HD inline void load_plane( uint8_t * smem, uint32_t at, const Frame & J, uint32_t type )
{
  const AA_GLOBAL uint32_t * src = (const AA_GLOBAL uint32_t *) &J.job->fp.coeff_probs[type][0][0][0];
  auto dst = lds_at<uint32_t>( smem, at );
  for ( uint32_t k = 0; k < kPlaneBytes / 4; k++ ) dst[k] = src[k];
}

constexpr unsigned long long kMemWaitTicks = 200000000ull;      // 2 s of the 100 MHz clock: then the frame is handed back (TOK_NO_MEMORY)

// The pass in PHASES (round 6): what a macroblock boundary decides -- wait, a coded macroblock to set up, the frame (or partition)
// through, a partition switch -- is found by a loop that walks runs of skipped macroblocks and touches only the lane's position
// (mi, col, row, the non-zero flags); what follows from the verdict is straight-line code behind the loop.  With the set-up inside
// the loop (every exit a `return`) the compiler carried every field of the lane any exit writes through the loop's merge points:
// ~85 register copies per pass, each an issue slot of the whole wave.
template <bool PK, bool MP = false>
AA_HD inline void macroblock_boundary_body( Lane & L, uint8_t * smem, const Frame & J, const Heap & H )
{
  const bool mp = MP && J.mp_P > 1;                         // this frame has one lane per partition
  const uint32_t above = ( mp ? J.mp_owner : L.base ) + kAbove;
  const uint32_t shared = J.mp_owner + part_off( J.mbw, true );   // (mp) MpShared
  if constexpr ( MP ) if ( L.rec == R_PARK ) {
    if ( AA_LDS_LOAD( smem, shared + offsetof( MpShared, left ) ) == 0u ) L.rec = R_DONE;
    return;
  }
  if ( L.rec == R_MBDONE ) {
    above_store<MP>( smem, above, J.mbw, L.col, L.ctxbits );           // the column's flags as the macroblock leaves them
    L.mi++; L.col++; L.rec = R_MB;
    if constexpr ( MP ) { L.mi_real++; if ( mp ) *lds_at<uint32_t>( smem, shared + 4 * J.mp_p ) = L.mi; }       // completed: the row below may follow
  }
  enum : uint32_t { V_WAIT = 0, V_CODED = 1, V_FINISH = 2, V_SWITCH = 3 };
  uint32_t verdict = V_WAIT, flags = 0, status = TOK_OK;
  if ( L.steps > J.max_steps ) { verdict = V_FINISH; status = TOK_STEP_BOUND; }      // cannot happen for any input; if it does the frame is reported, not hung on
  else for ( ;; ) {
    if constexpr ( MP ) if ( mp && AA_LDS_LOAD( smem, shared + offsetof( MpShared, status ) ) != TOK_OK ) { verdict = V_FINISH; break; }   // another lane of the frame gave up (its status stands): so does this one
    if ( L.mi == J.nmb ) { verdict = V_FINISH; break; }
    if ( L.mi >= L.mwpos ) break;                           // flags not here yet (only a long run of skipped macroblocks gets ahead of the ring)
    if ( L.col == J.mbw ) {
      L.col = 0; L.row++; L.ctxbits = 0;
      if constexpr ( MP ) if ( mp ) L.mi_real += ( J.mp_P - 1u ) * J.mbw;      // this lane's next row is P rows further down
      if ( !mp && J.nparts > 1 ) { verdict = V_SWITCH; break; }               // (the pass after this one goes on from column 0)
    }
    if constexpr ( MP ) if ( mp && ( J.mp_p | L.row ) != 0 ) {
      // (r, c) needs the flags (r - 1, c) left behind: row r - 1 is the previous partition's -- its L.row-th row, or, for
      // partition 0, the last partition's (L.row - 1)-th
      const uint32_t q = J.mp_p ? J.mp_p - 1u : J.mp_P - 1u, k = J.mp_p ? L.row : L.row - 1u;
      if ( AA_LDS_LOAD( smem, shared + 4 * q ) < k * J.mbw + L.col + 1u ) break;      // not there yet: ask again at the next pass
    }
    flags = smem[L.base + kMeta + ( L.mi & ( kMetaRing - 1 ) )];
    L.ctxbits = ( L.ctxbits & 0x01FF0000u ) | above_load<MP>( smem, above, J.mbw, L.col );
    if ( !( flags & AA_MB_SKIP ) ) { verdict = V_CODED; break; }
    const uint32_t has_y2 = flags & AA_MB_HAS_Y2;
    L.ctxbits &= has_y2 ? 0u : 0x01000100u;                 // a non-coded Y2 leaves its chain untouched (frame.cc:255-269)
    above_store<MP>( smem, above, J.mbw, L.col, L.ctxbits );
    const uint32_t mi_rec = MP ? L.mi_real : L.mi;
    if constexpr ( PK ) store_mb_packed( J, mi_rec, 0, 0, flags | ( has_y2 ? AA_MB_LF_SKIP_INNER : 0u ) );
    else store_mb( J, mi_rec, 0, L.blk_index, flags | ( has_y2 ? AA_MB_LF_SKIP_INNER : 0u ) );
    L.mi++; L.col++;
    if constexpr ( MP ) { L.mi_real++; if ( mp ) *lds_at<uint32_t>( smem, shared + 4 * J.mp_p ) = L.mi; }
  }
  if ( verdict == V_WAIT ) return;
  if ( verdict == V_SWITCH ) { switch_partition<MP>( L, smem, J, L.row % J.nparts ); return; }
  if ( verdict == V_FINISH ) {
    if ( mp ) finish_partition<PK>( L, smem, J, H, status );
    else finish_frame<PK>( L, J, H, status );
    return;
  }
  // ---- a coded macroblock ----
  const uint32_t has_y2 = flags & AA_MB_HAS_Y2;
  // words of the chunk in use (packed storage; between macroblocks blk = the next free word)
  const uint32_t used = PK && L.nchunks ? static_cast<uint32_t>( L.blk - ( H.base + static_cast<size_t>( L.blk_index ) * 16 ) ) : 0u;
  // (+ 1: blk -- one past the macroblock's last value when all 400 are there -- stays inside the lane's own chunk: the step stores at it)
  if ( PK ? ( !L.nchunks || used + kMbWords + 1u > kChunkWords ) : L.blk_left < kMbBlocks ) {   // the chunk cannot take a whole macroblock: on to a new one
    const uint32_t c = pool_take( H );
    if ( c == kNoChunk ) {
      // nothing free right now: this lane sits the steps out and asks again at the next boundary pass (its wave-mates keep
      // decoding).  The host maps more heap when it sees lanes starve; if nothing comes for kMemWaitTicks the frame is
      // handed back unfinished and the host runs it again when memory has been released.
      const unsigned long long now = AA_NOW();
      if ( !L.mem_since ) { L.mem_since = now | 1ull; AA_AT_ADD( &H.pool->starving, 1u ); }
      else if ( now - L.mem_since > kMemWaitTicks ) { if ( mp ) finish_partition<PK>( L, smem, J, H, TOK_NO_MEMORY ); else finish_frame<PK>( L, J, H, TOK_NO_MEMORY ); }
      return;
    }
    L.mem_since = 0;
    if ( mp ) {
      L.chunk_ord = AA_LDS_ADD( smem, shared + offsetof( MpShared, nchunks ), 1u );
      J.chunk_list[1 + L.chunk_ord] = c;
    } else {
      J.chunk_list[1 + L.nchunks] = c;
      if constexpr ( MP ) L.chunk_ord = L.nchunks;
    }
    L.nchunks++;
    L.blk_index = c * kChunkBlocks;
    if constexpr ( PK ) {
      L.words += used;
      L.blk = H.base + static_cast<size_t>( L.blk_index ) * 16;
    } else {
      L.blk = H.base + static_cast<size_t>( L.blk_index ) * 16;
      L.blk_left = kChunkBlocks;
      zero_slot( L.blk );
    }
  }
  if constexpr ( PK ) {
    // the macroblock's words: kMaskSlots mask slots, then the values (coeff_pack.hh)
    L.mb_first = ( ( MP ? L.chunk_ord : L.nchunks - 1u ) << 15 ) | static_cast<uint32_t>( L.blk - ( H.base + static_cast<size_t>( L.blk_index ) * 16 ) );
    L.hdr = L.blk; L.blk = L.hdr + kMbMaskSlots; L.zzmask = 0;
  } else L.mb_first = L.blk_index;
  L.flags = flags; L.nz_mask = 0;
  const uint32_t kind = has_y2 ? Y_AFTER_Y2 : Y_WITHOUT_Y2;
  if ( L.ykind != kind ) { load_plane( smem, L.base + kPlaneY, J, kind ); L.ykind = kind; }
  L.yfirst = has_y2 ? 1u : 0u;
  setup_block( L, smem, has_y2 ? 0u : 1u );
}

template <bool PK, bool MP = false>
AA_HD inline void macroblock_boundary( Lane & L, uint8_t * smem, const Frame & J, const Heap & H )
{
  macroblock_boundary_body<PK, MP>( L, smem, J, H );
  preload( L, smem );           // (wherever the lane stands now -- a block's first node, an idle record, another partition's bytes)
}

template <bool MP = false> AA_HD inline bool at_boundary( const Lane & L ) { return L.rec == R_MBDONE || L.rec == R_MB || ( MP && L.rec == R_PARK ); }

// ---- one step: decode one bool (lanes with a node to decode; the others sit it out) -------------------------------------
// Straight-line code: what the bit means comes out of the node's record as bit fields and is applied with arithmetic;
// nothing is predicated (the coefficient store was until round 6's session 14: see there).  A lone wave gets one issue slot every 4 cycles whatever the
// instruction -- vector, scalar, branch or wait -- so every instruction saved here is 4 cycles per bool.
//
// The END OF A BLOCK is not part of the step (it was, as a predicated region: ~55 issue slots that the wave paid whenever ANY of
// its lanes was there -- four steps out of five at 22 lanes per wave).  A lane whose block has ended parks (R_BEND) and the wave
// runs tok::block_end for all parked lanes every kBendEvery steps: the wave pays a quarter of those slots per step, a lane
// waits (kBendEvery - 1) / 2 steps per block on average (a block is ~16 bools on inter frames, ~39 on key frames).
#ifndef AA_BEND_EVERY
#define AA_BEND_EVERY 4               /* build parameter (A/B runs): steps between block-end passes */
#endif
constexpr uint32_t kBendEvery = AA_BEND_EVERY;
static_assert( kPeriod % kBendEvery == 0 && kBendEvery <= 8, "a period is a whole number of step groups; Frame::max_steps allows for 8" );

template <bool PK, bool MP = false>
AA_HD inline void step( Lane & L, uint8_t * smem, const Frame & J )
{
  (void) J;
  // EVERY lane runs the step (idle lanes: see "nodes" above); no condition, no branch, nothing predicated.  The LDS reads of a step;
  // all addresses were known at the end of the previous one
#if AA_STEP_PRELOAD
  const uint32_t pbyte = L.pre_prob, raw = L.pre_raw, nband = L.pre_band;   // (asked for when the previous step -- or whoever moved the lane since -- knew the addresses)
  const V8 rec = L.pre_rec;
#else
  const uint32_t pbyte = *lds_at<const uint8_t>( smem, L.paddr );
  const uint32_t raw = *lds_at<const uint8_t>( smem, L.sbase | ( L.rpos & ( kRing - 1 ) ) );
  const V8 rec = *lds_at<const V8>( smem, L.rec );
  const uint32_t nband = *lds_at<const uint8_t>( smem, L.ia + 1u );       // 33 * band of the NEXT position (the only one the row can move to)
#endif
  const uint32_t was_idle = L.rec;              // (the record this step decodes: L.rec moves on below)

  // ... and while they travel: the renormalisation the previous step left undone (bool_decoder.hh:94-105; a lane's decoder is
  // normalised everywhere but between two steps -- the shift of an already normalised range is 0)
  const int shift = AA_CLZ( L.range ) - 24;
  L.range <<= ( shift & 31 );
  L.value <<= ( shift & 31 );
  L.sh += shift;
#if AA_STEP_SCHED_BARRIER && defined( __HIP_DEVICE_COMPILE__ )
  // Everything above this line -- the previous step's coefficient store and bookkeeping, this step's renormalisation -- needs none of the
  // four values asked for by the previous step's preload(), everything below does: the scheduler keeps the two apart (left alone it has
  // put the first wait 8 instructions behind the reads and the independent work behind the wait: a lone wave per SIMD then sits the
  // LDS round trip out).
  __builtin_amdgcn_sched_barrier( 0 );
#endif
  // top the window up by one byte whenever one fits (sh >= 0): a decode shifts out at most 7 bits, so the 8 bits being
  // compared are always real and the refill is never on the critical path.  Mask arithmetic, no condition.
  const uint32_t room = ~static_cast<uint32_t>( L.sh >> 31 );
  L.value |= ( raw << ( L.sh & 31 ) ) & room;
  L.sh -= static_cast<int32_t>( 8u & room );
  L.rpos -= room;
  const uint32_t tn = L.typeaddr + nband;       // (the row of the next position, but for the context the token leaves)

  // an idle lane (record address >= 512) decodes with probability 256: split = range, nothing of its decoder moves
  const uint32_t prob = was_idle >= 512u ? 256u : pbyte;
  // BoolDecoder::get (bool_decoder.hh:67-107)
  const uint32_t split = ( AA_MUL24( L.range - 1, prob ) + 256u ) >> 8;   // = 1 + (((range - 1) * prob) >> 8)
  const uint32_t bigsplit = split << 24;
  const bool bit = L.value >= bigsplit;
  L.range = bit ? L.range - split : split;
  L.value = bit ? L.value - bigsplit : L.value;

  // what the node says this bit means
  const uint32_t h = bit ? rec.y : rec.x;
  // where the lane goes next -- first: the next step's LDS reads wait for these addresses.  Position 16 ends the block like an EOB
  // (whose record half leads to R_BEND by itself)
  const uint32_t adv = AA_UBFE( h, 16, 1 );     // on to the next coefficient position?
  const uint32_t ia = L.ia + adv;
  // (a select by mask arithmetic: written as a conditional the compiler makes it a predicated region -- measured on MI355X, round 5:
  // 4 % of a wave step)
  const uint32_t rowaddr = AA_BFI( 0u - adv, tn + AA_UBFE( h, 19, 5 ), L.rowaddr );
  const uint32_t paddr = ( static_cast<int32_t>( h ) < 0 ? rowaddr : kXtab ) + AA_UBFE( h, 10, 5 );
  const uint32_t nextrec = ( ia & 16u ) ? static_cast<uint32_t>( R_BEND ) : ( h & 1023u );
  const uint32_t pos = L.ia;                    // (the position of the token this step completes, if it does)
  L.rowaddr = rowaddr; L.paddr = paddr; L.rec = nextrec; L.ia = ia;
  preload( L, smem );                           // the next step's reads travel while this one stores its coefficient
#if AA_STEP_SCHED_BARRIER && defined( __HIP_DEVICE_COMPILE__ )
  __builtin_amdgcn_sched_barrier( 0 );          // (... and nothing of what follows is done in front of them)
#endif

  const uint32_t xs = AA_UBFE( h, 15, 1 );
  const uint32_t mag = ( L.mag << xs ) | ( ( bit ? 1u : 0u ) & xs );    // extra bits shift in; everything else leaves it alone
  const uint32_t emit = AA_UBFE( h, 17, 1 );    // the sign: the token is complete (tokens.cc:126-133)
  const int32_t m = static_cast<int32_t>( mag + AA_UBFE( h, 24, 7 ) );
  const int16_t coeff = static_cast<int16_t>( bit ? -m : m );
  if constexpr ( PK ) {                         // the next value of the block, its zigzag position into the mask
    // The store is UNCONDITIONAL: blk is the next free word of the lane's macroblock (or the lane's sink while it has no chunk), and a
    // word there that no token completed is either overwritten by the token that does or never read (a reader takes popcount(mask)
    // values).  As `if ( emit )` it was the step's one predicated region -- an exec-mask save, a branch and a restore that nearly every
    // step of a wave of 30 lanes ran anyway, and a split of the step into basic blocks the scheduler could not move the LDS reads'
    // waits across (measured, session 13: the same source scheduled with the waits 17 instructions earlier was 6 % slower).
#if AA_STEP_STORE_ALWAYS == 2
    // (... at the lane's sink when no token was completed: with every lane of the wave storing at its own block the store touched 30
    // cache lines a step instead of the ~6 of the lanes that have a value -- alone 8 % faster, beside the reconstruction kernels 5 %
    // slower than the predicated store, session 14)
    *( emit ? L.blk : L.sink ) = coeff;
#elif AA_STEP_STORE_ALWAYS
    *L.blk = coeff;
#else
    if ( emit ) *L.blk = coeff;
#endif
    L.blk = bump_words( L.blk, emit );
    L.zzmask |= emit << ( pos & 31u );
  } else {
    const uint32_t zz = static_cast<uint32_t>( kZigzagNib >> ( ( pos & 15u ) * 4 ) ) & 15u;
    if ( emit ) L.blk[zz] = coeff;
    L.nonzero |= emit;
  }
  L.mag = emit ? 0u : mag;
}

// ---- the end of a block, for the lanes that wait for it (R_BEND): the block's flags and mask word, the macroblock's record
// if it was its last block, then the next block's contexts and first probability row ----
template <bool PK, bool MP = false>
AA_HD inline void block_end( Lane & L, uint8_t * smem, const Frame & J, const Heap & H )
{
  (void) H;
  if ( L.rec == R_BEND ) {
    const V8 nextblk = *lds_at<const V8>( smem, L.blkaddr );
    const bool nonzero = PK ? L.zzmask != 0 : L.nonzero != 0;
    const uint32_t ctxbits = nonzero ? L.ctxbits | L.nzsel : L.ctxbits & ~L.nzsel;
    if constexpr ( PK ) {
      // the block's mask into its slot at the head of the macroblock's words
      if ( nonzero ) { L.coeff_blocks++; L.hdr[L.blkslot] = static_cast<int16_t>( L.zzmask ); L.zzmask = 0; L.nz_mask |= L.blkbit; }
    } else {
      if ( nonzero ) { L.coeff_blocks++; L.blk += 16; L.blk_index++; L.blk_left--; zero_slot( L.blk ); L.nz_mask |= L.blkbit; }
    }
    const bool mbdone = L.blkaddr == kBlockTabOff + 8 * 25;
    if ( mbdone ) {                           // the macroblock is complete: its record (its column's flags: at the boundary pass)
      uint32_t flags = L.flags;
      flags |= L.nz_mask ? AA_MB_HAS_NONZERO : ( ( flags & AA_MB_HAS_Y2 ) ? AA_MB_LF_SKIP_INNER : 0u );
      if constexpr ( PK ) {
        store_mb_packed( J, MP ? L.mi_real : L.mi, L.nz_mask, L.mb_first, flags, static_cast<unsigned long long>( L.hdr - H.base ) );
        if ( !L.nz_mask ) L.blk = L.hdr;      // coded, and every block empty: the mask slots go back (the macroblock stores nothing)
      } else store_mb( J, MP ? L.mi_real : L.mi, L.nz_mask, L.mb_first, flags );
    }
    // the block after it (never a Y2)
    const uint32_t sel = AA_UBFE( nextblk.x, 16, 8 );
    const uint32_t nctx = AA_POPC( ctxbits & nextblk.y );
    L.typeaddr = L.base + 8u * ( sel & ( kBlkIsY - 1u ) );
    const uint32_t idx = ( sel >> 7 ) & L.yfirst;
    L.ia = kBandTabOff + idx;
    L.rowaddr = L.paddr = L.typeaddr + idx * 33u + nctx * 11u;
    L.ctxbits = ctxbits;
    L.blkaddr += 8;
    L.nzsel = nextblk.y;
    L.blkslot = nextblk.x >> 24;
    L.blkbit = 1u << L.blkslot;
    L.nonzero = 0; L.mag = 0;
    L.rec = mbdone ? static_cast<uint32_t>( R_MBDONE ) : 0u;
    preload( L, smem );
  }
}

// One period of a wave: kPeriod steps in groups of kBendEvery, each group followed by the block-end pass if a lane waits for
// one; the hot loop is left whenever a lane has reached a macroblock boundary (which only a block-end pass can bring about).
// `prof` (diagnostics, may be null): [0] += clock ticks spent in boundary passes, [1] += boundary passes, [2] += steps of the wave
template <bool PK, bool MP = false>
AA_HD inline void run_period( Lane & L, uint8_t * smem, const Frame & J, const Heap & H, unsigned long long * prof = nullptr )
{
  uint32_t it = 0;
  while ( it < kPeriod ) {
    if ( AA_ANY( at_boundary<MP>( L ) ) ) {
      const unsigned long long tb = prof ? AA_NOW() : 0ull;
      if ( at_boundary<MP>( L ) ) macroblock_boundary<PK, MP>( L, smem, J, H );
      if ( prof ) { prof[0] += AA_NOW() - tb; prof[1]++; }
      it++;                                                 // (a lane waiting for flags must not spin the period away)
      if ( !AA_ANY( L.rec < R_MBDONE ) ) break;             // nobody has anything to decode (no lane is at R_BEND out here)
    }
    const uint32_t it0 = it;
    do {
#if AA_STEP_UNROLL
#pragma unroll
#endif
      for ( uint32_t k = 0; k < kBendEvery; k++ ) step<PK, MP>( L, smem, J );
      it += kBendEvery;
      // (asked by ALL lanes, outside the predicated regions: wave-uniform)
      if ( AA_ANY( L.rec == R_BEND ) ) block_end<PK, MP>( L, smem, J, H );
    } while ( it < kPeriod && !AA_ANY( L.rec == R_MBDONE ) );
    if ( prof ) prof[2] += it - it0;
  }
  if ( L.rec != R_DONE ) L.steps += it;                     // (an upper bound: the iterations a lane sat out count too)
}

// ---- a lane's life ---------------------------------------------------------------------------------------------------
// the node and block tables, once per workgroup: word k of kTablesBytes / 4 (the caller spreads k over its threads)
AA_HD inline uint32_t table_word( uint32_t k )
{
  const uint32_t * n = reinterpret_cast<const uint32_t *>( &kNodeTable );
  const uint32_t * b = reinterpret_cast<const uint32_t *>( &kBlockTable );
  if ( k < sizeof( NodeTable ) / 4 ) return n[k];
  if ( k >= kBlockTabOff / 4 && k < ( kBlockTabOff + sizeof( BlockTable ) ) / 4 ) return b[k - kBlockTabOff / 4];
  if ( k >= kIdleTabOff / 4 && k < kIdleTabOff / 4 + 2 * kIdleRecs ) return idle_half( ( k - kIdleTabOff / 4 ) / 2 );
  if ( k >= kBandTabOff / 4 && k < kBandTabOff / 4 + 5 ) {          // 33 * band of positions 0 .. 17 (16, 17: never read as a row)
    uint32_t w = 0;
    for ( uint32_t i = 0; i < 4; i++ ) { const uint32_t at = 4 * ( k - kBandTabOff / 4 ) + i; w |= ( at < 16 ? 33u * ( static_cast<uint32_t>( kBandNib >> ( 4 * at ) ) & 15u ) : 0u ) << ( 8 * i ); }
    return w;
  }
  if ( k >= kXtab / 4 && k < kXtab / 4 + 7 ) {
    uint32_t w = 0;
    for ( uint32_t i = 0; i < 4; i++ ) { const uint32_t at = 4 * ( k - kXtab / 4 ) + i; w |= ( at < 27 ? static_cast<uint32_t>( kXtabInit[at] ) : 0u ) << ( 8 * i ); }
    return w;
  }
  return 0;
}

// A lane before its first frame: idle (R_DONE), every address the step reads in range -- the step runs for idle lanes too
AA_HD inline void init_lane( Lane & L, uint32_t sbase, uint32_t base, AA_GLOBAL int16_t * sink )
{
  L.sink = sink;
  L.sbase = sbase; L.base = base;
  L.rec = R_DONE; L.paddr = kXtab + kZeroX; L.ia = kBandTabOff; L.rowaddr = L.typeaddr = base;
  L.value = 0; L.range = 255; L.sh = 0; L.rpos = 0; L.mag = 0; L.zzmask = 0; L.nonzero = 0;
  L.blk = sink; L.hdr = nullptr;
  L.pend_wpos = L.pend_mwpos = kNoPend;
  L.steps = 0;
}

// SH: the layout of the above-row flags (tok::kAbove) -- that of the kernel the lane runs in: shared by the lanes of a frame
// (one lane per partition, the MP instantiations) or the lane's own
template <bool SH = false>
AA_HD inline void begin_frame( Lane & L, uint8_t * smem, uint32_t base, const Frame & J )
{
  // lane LDS: the UV and Y2 probability planes (the Y plane: when the first coded macroblock says which), zeroed above-row
  // flags / partition save area
  L.base = base;
  uint8_t * lds = smem + base;
  load_plane( smem, base + kPlaneUV, J, UV );
  load_plane( smem, base + kPlaneY2, J, Y2 );
  L.ykind = kNoPlane;
  if ( J.nparts > 1 ) for ( uint32_t k = 0; k < 128 / 4; k++ ) reinterpret_cast<uint32_t *>( lds + part_off( J.mbw, SH ) )[k] = 0;
  for ( uint32_t k = 0; k < above_bytes( J.mbw, SH ); k++ ) lds[kAbove + k] = 0;
  L.mi = 0; L.col = 0; L.row = 0; L.ctxbits = 0; L.coeff_blocks = 0; L.steps = 0;
  L.flags = L.nz_mask = L.mb_first = L.yfirst = 0;
  L.nonzero = L.nzsel = L.blkbit = L.blkslot = L.mag = 0;
  L.ia = kBandTabOff;
  L.typeaddr = L.rowaddr = base; L.paddr = kXtab + kZeroX;
  L.blkaddr = kBlockTabOff;
  L.blk = L.sink; L.blk_index = 0; L.blk_left = 0; L.nchunks = 0; L.mem_since = 0;     // the first coded macroblock takes the first chunk
  L.hdr = nullptr; L.zzmask = 0; L.words = 0;
  L.mi_real = 0; L.chunk_ord = 0;
  if ( J.mp_P > 1 ) {
    // one lane per partition: this lane's first row is row mp_p; the owner's slice holds what the lanes share
    L.mi_real = J.mp_p * J.mbw;
    if ( base == J.mp_owner ) {
      MpShared z {};
      z.left = J.mp_P;
      *reinterpret_cast<MpShared *>( lds + part_off( J.mbw, SH ) ) = z;
    }
  }
  start_partition( L, smem, J, J.mp_P > 1 ? J.mp_p : 0u );
  // flag ring: macroblocks [0, kMetaRing)
  for ( uint32_t k = 0; k < kMetaRing / 16; k++ ) {
    Chunk16 c; c.w[0] = c.w[1] = c.w[2] = c.w[3] = 0;
    if ( 16 * k < J.flags_padded ) c = load16( J.mbflags + 16 * k );
    lds_store16( smem, base + kMeta + 16 * k, c );
  }
  L.mwpos = kMetaRing;
  L.pend_mwpos = kNoPend;
  L.rec = R_MB;
  preload( L, smem );
}

} // namespace tok
} // namespace aa
