// Device-side entropy decode: the reference's HOT LOOP #1 -- Frame::parse_macroblock_headers and
// Frame::parse_tokens (frame.cc:95-137) over BoolDecoder (bool_decoder.hh:45-120) -- with one GPU lane per arithmetic-coded
// chain.  A VP8 partition is strictly serial, so the parallelism is across (stream, frame, partition kind): the host runs
// only the frame-header pre-pass (Parser::parse_header, ~1 k bools per frame, serial across the frames of a stream because
// of the persistent probability tables); everything per macroblock happens here, records written straight into HBM in the
// layout the reconstruction kernels read.
//
//   k_parse_mb_headers  one lane per frame: first partition behind the frame header -> aa_mb_info (modes, motion vectors,
//                       segment, loop-filter level), the flags byte array the token lanes read, intra row bitmasks.
//                       The code is parse_mb_header() of parse_common.hh, the same statements the host parser runs.
//   k_segment_fixup     only for streams that use segmentation: walks the frames of a stream in order, maintains the persistent
//                       segment map and gives frames that inherit it their ids and loop-filter levels.
//   k_parse_tokens      one lane per frame: DCT partitions -> coefficient blocks, nz_mask, coeff_index, flags.  Flat state
//                       machine, one bool per step (tok_fsm.hh).
//
// A lane is latency-bound (dependent ALU chain + one LDS read per bool), not throughput-bound: what makes this fast is the
// number of chains in flight (hundreds of streams x frames), which is why the kernels use few lanes per wave (divergent
// events -- macroblock / row boundaries -- then stall few neighbours) and spread over as many SIMDs as there are chains.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "device_types.h"
#include "tok_fsm.hh"
#include "coeff_pack.hh"

// The issue arbiter's priority of a worker wave (s_setprio; the reconstruction kernels run at 3).  At 0-2 a reconstruction wave on the same
// SIMD always issues first; at 3 the arbiter takes the OLDEST ready wave, which is the worker (resident for seconds): a wave step beside
// the reconstruction kernels 0.216 -> 0.200 us, the reconstruction kernels ~20 % slower, end to end 141.5 -> 146.2 M macroblocks/s
// (two runs each, profiles/r06_bench_sessions.md session 15; session 13: 141.5 -> 144.7).  The row kernels' hand-off chains still get the
// issue slots the worker leaves -- it waits for LDS 45 % of the time.
#ifndef AA_WORKER_PRIO
#define AA_WORKER_PRIO 3
#endif

namespace {

using aa::ParseJob;

// `order`: which job each slot of the launch takes -- the host sorts the frames of a batch by the length of their chains so
// that the lanes of a wave finish together (a wave lasts as long as its longest lane)
__global__ __launch_bounds__( 64 ) void k_parse_mb_headers( const ParseJob * jobs, const uint32_t * order, int n, int lanes )
{
  __builtin_amdgcn_s_setprio( 2 );      // short chains the long token chains wait for: ahead of them at the issue arbiter

  // Everything the parse reads with a data-dependent index lives in LDS: the constant trees and probability tables (one copy per
  // workgroup) and the frame's own header parameters (one copy per lane).  Read out of the kernel's constant data and out of the job
  // in HBM they were two dependent global loads in front of nearly every bool: 170 ms per hand-over of inter frames, half a second
  // for key frames (16 tree decodes per B_PRED macroblock), on the path every frame's token lane waits for.
  __shared__ aa::HeaderTables s_tables;
  __shared__ aa::HeaderParams s_params[16];
  const int lane = threadIdx.x;
  for ( uint32_t k = lane; k < sizeof( aa::HeaderTables ) / 4; k += 64 ) reinterpret_cast<uint32_t *>( &s_tables )[k] = reinterpret_cast<const uint32_t *>( &aa::kHeaderTables )[k];
  const int slot = blockIdx.x * lanes + lane;
  const bool mine = lane < lanes && lane < 16 && slot < n;
  const ParseJob * Jp = mine ? &jobs[order[slot]] : nullptr;
  if ( Jp ) {
    const uint32_t * src = reinterpret_cast<const uint32_t *>( static_cast<const aa::HeaderParams *>( &Jp->fp ) );
    for ( uint32_t k = 0; k < sizeof( aa::HeaderParams ) / 4; k++ ) reinterpret_cast<uint32_t *>( &s_params[lane] )[k] = src[k];
  }
  __syncthreads();
  if ( !Jp ) return;
  const ParseJob & J = *Jp;
  if ( J.nmb == 0 ) return;                      // a frame the host header pre-pass rejected
  const aa::HeaderParams & fp = s_params[lane];
  const aa::HeaderTables & T = s_tables;
  aa::BoolReader32 bd;
  aa::BoolState st; st.bitpos = fp.bd_bitpos; st.range = fp.bd_range; st.active = fp.bd_active;
  bd.resume( J.data + fp.first_off, fp.first_size, st );
  const unsigned mbw = fp.mbw, mbh = fp.mbh;
  const unsigned words_per_row = ( mbw + 63 ) / 64;
  uint32_t intra = 0, split = 0;
  unsigned mi = 0;
  // (one lane per partition: a second copy of the flags, every partition's rows back to back -- aa::mp_flag_index, with
  // everything it reads out of the job held in registers: the loop must not wait for HBM)
  const uint32_t mp_stride = J.mp_stride, mp_base = J.flags_padded, nparts = fp.nparts;
  for ( unsigned row = 0; row < mbh; row++ ) {
    unsigned long long word = 0;
    const uint32_t mp_row = mp_stride ? mp_base + ( row % nparts ) * mp_stride + ( row / nparts ) * mbw : 0u;
    for ( unsigned col = 0; col < mbw; col++, mi++ ) {
      const uint8_t flags = aa::parse_mb_header( bd, fp, T, J.mbs, mi, col, row, static_cast<uint8_t *>( nullptr ) );
      J.mbflags[mi] = flags;
      if ( mp_stride ) J.mbflags[mp_row + col] = flags;
      if ( !( flags & AA_MB_INTER ) ) { intra++; word |= 1ull << ( col & 63 ); }
      else if ( J.mbs[mi].y_mode == aa::SPLITMV ) split = 1;
      if ( ( col & 63 ) == 63 || col + 1 == mbw ) { J.intra_rows[row * words_per_row + ( col >> 6 )] = word; word = 0; }
    }
  }
  J.summary->num_intra_mbs = intra;
  J.summary->has_split = split;
}

__global__ __launch_bounds__( 256 ) void k_segment_fixup( const ParseJob * jobs, const aa_seg_stream * streams, const uint32_t * order )
{
  const aa_seg_stream s = streams[blockIdx.x];
  for ( uint32_t k = 0; k < s.count; k++ ) {
    const uint32_t o = order[s.first + k];
    const ParseJob & J = jobs[o & 0x7FFFFFFFu];
    if ( !J.fp.seg_enabled || J.nmb == 0 ) continue;
    if ( o >> 31 ) {                                    // the map restarts at all-3 (Segmentation ctor, decoder_state.hh:170-176)
      for ( uint32_t mi = threadIdx.x; mi < J.nmb; mi += blockDim.x ) s.map[mi] = 3;
      __syncthreads();
    }
    for ( uint32_t mi = threadIdx.x; mi < J.nmb; mi += blockDim.x ) aa::segment_fixup( J.fp, J.mbs[mi], s.map[mi] );
    __syncthreads();
  }
}

// ---- token workers ----------------------------------------------------------------------------------------------------
// A workgroup = one wave; `lanes` of its threads are token lanes.  A lane without a frame takes the next job of the queue; a
// wave leaves when none of its lanes has a frame and the queue has nothing for them (or its grid was told to retire: the
// host wants the stream the grid was launched on).  Work arriving later is picked up by the lanes that become free -- or by a
// grid the host launches when too few workgroups are alive (runtime.cpp, ensure_workers).
struct WorkerArgs {
  aa::TokQueue * q;
  unsigned long long * slots;             // q->mask + 1 job pointers
  aa::Heap heap;
  uint32_t * exited;                      // this grid's count of workgroups that have left (HBM)
  const uint32_t * retire;                // grids of this slot up to generation *retire take no new jobs (pinned host memory, mapped)
  uint32_t gen;                           // this grid's generation
  uint32_t spread;                        // workgroups the GPU holds: a wave takes its share of a short queue, not all it can carry
  unsigned long long * prof;              // diagnostics (null: off): 8 counters summed over the waves of all grids, see aa_tok_mirror::prof
  unsigned long long linger_ticks;        // a workgroup without work stays this long (100 MHz ticks) before it leaves
  int lanes;
  uint32_t lane_bytes;
  uint32_t mp_hint;                       // one lane per partition: most partitions a frame in flight has (a wave leaves that many lanes per ticket)
  // Per-CU admission (null: off).  The LDS request is exactly what a workgroup's lanes need, so a FOURTH worker workgroup fits a CU where
  // the shape wants three -- the reconstruction kernels' pace beside the workers is the LDS the workers leave, and their row kernels'
  // hand-off chains run at the pace of the CU with the least.  One grid spreads evenly; grids that overlap (top-ups, remnants) do not.
  // A workgroup counts itself in on the CU it landed on (HW_ID: shader engine, array, CU; XCC_ID) and LEAVES AT ONCE when that CU has
  // `cu_cap` already -- the host sees it among the exited and tops up, and the dispatcher, which fills the emptiest CUs first, puts the
  // replacement elsewhere.
  uint32_t * cu_slots;
  uint32_t cu_cap;
  int16_t * sink;                         // 8 KB nobody reads: where a lane without a chunk stores (tok::step), a 64-byte line per workgroup (mod 64)
};

// up to `want` tickets for this wave (called by ONE lane): -> first ticket in *base, how many as the result; 0 = the queue is
// EMPTY (nothing published that is not taken).  Losing the compare-and-swap to another wave is not "empty": hundreds of waves
// look at the same word when a grid starts, and a wave that gave up then would leave with the queue full (round 3, first
// version: ~80 of 1024 workgroups survived their first look, the GPU ran at 8 % of its lanes).  So: retry until taken or empty,
// backing off by a wave-dependent number of cycles.
// When the queue is short the wave takes only its share (queue length / workgroups the GPU holds, rounded up): a wave steps
// faster the fewer lanes it carries, and the other workgroups -- alive or about to start -- look at the same queue.
__device__ inline uint32_t queue_take( aa::TokQueue * q, uint32_t want, uint32_t spread, uint32_t * base )
{
  uint32_t h = AA_AT_LOAD( &q->head );
  for ( uint32_t tries = 0; ; tries++ ) {
    const uint32_t avail = AA_AT_LOAD( &q->publish ) - h;
    if ( static_cast<int32_t>( avail ) <= 0 ) return 0;
    const uint32_t share = ( avail + spread - 1u ) / spread;
    const uint32_t n = want < share ? want : share;
    uint32_t expect = h;
    if ( __hip_atomic_compare_exchange_strong( &q->head, &expect, h + n, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT ) ) { *base = h; return n; }
    h = expect;
    // lost: somebody else moved the head.  Wait a little before the next try, differently per workgroup and per try
    const uint32_t r = ( blockIdx.x * 2654435761u + tries * 40503u ) >> 27;       // 0..31
    for ( uint32_t k = 0; k <= ( r & ( tries < 4 ? 3u : tries < 8 ? 15u : 31u ) ); k++ ) __builtin_amdgcn_s_sleep( 16 );
    if ( tries > 2048 ) h = AA_AT_LOAD( &q->head );
  }
}

// position of the n-th set bit of `mask` (n < popcount)
__device__ inline int nth_set_bit( unsigned long long mask, uint32_t n )
{
  for ( uint32_t k = 0; k < n; k++ ) mask &= mask - 1ull;
  return __ffsll( static_cast<long long>( mask ) ) - 1;
}

// PK: the coefficients are stored packed (tok_fsm.hh); MP: frames with several DCT partitions may get a lane per partition.  A
// context runs one instantiation for all its frames.
template <bool PK, bool MP>
__global__ __launch_bounds__( 64 ) void k_token_workers( const WorkerArgs a )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) uint8_t smem[];
  const int lane = threadIdx.x;
#if AA_WORKER_PRIO
  __builtin_amdgcn_s_setprio( AA_WORKER_PRIO );
#endif
  uint32_t cu_slot = 0;
  if ( a.cu_slots ) {
    // HW_REG_HW_ID (4): CU_ID [11:8], SH_ID [12], SE_ID [15:13]; HW_REG_XCC_ID (20) [3:0]
    const uint32_t hw = __builtin_amdgcn_s_getreg( 4 | ( 8 << 6 ) | ( 7 << 11 ) ), xcc = __builtin_amdgcn_s_getreg( 20 | ( 0 << 6 ) | ( 3 << 11 ) );
    cu_slot = ( ( xcc & 15u ) << 8 ) | ( hw & 255u );
    uint32_t before = 0;
    if ( lane == 0 ) before = AA_AT_ADD( &a.cu_slots[cu_slot], 1u );
    before = __shfl( before, 0 );
    if ( before >= a.cu_cap ) {             // this CU has its share of worker workgroups: make room for the reconstruction kernels
      if ( lane == 0 ) { AA_AT_ADD( &a.cu_slots[cu_slot], 0xFFFFFFFFu ); AA_AT_ADD( &a.cu_slots[AA_CU_SLOTS - 1], 1u ); AA_AT_ADD( a.exited, 1u ); }
      return;
    }
  }
  for ( uint32_t k = lane; k < aa::tok::kTablesBytes / 4; k += 64 ) reinterpret_cast<uint32_t *>( smem )[k] = aa::tok::table_word( k );
  __syncthreads();
  aa::tok::Lane L;
  aa::tok::Frame F {};
  // (threads that are no token lanes run the steps too -- idle, at the addresses of lane 0: reads only)
  const bool is_lane = lane < a.lanes;
  aa::tok::init_lane( L, aa::tok::ring_addr( is_lane ? lane : 0 ), aa::tok::slice_addr( is_lane ? lane : 0, a.lanes, a.lane_bytes ),
                      (AA_GLOBAL int16_t *) ( a.sink + ( blockIdx.x & 63u ) * 32u + ( lane & 31 ) ) );
  aa::tok::preload( L, smem );              // (the tables are in LDS: an idle lane's record, probability, band)
  uint32_t backoff = 0;
  unsigned long long idle_since = 0;      // the wave has had no frame since (0: it has one)
  uint32_t looks = 0;
  bool retiring = false;                  // sticky: once the host has told this grid to retire, its waves take no more jobs
  // diagnostics: where a wave's time goes (100 MHz ticks): [0] boundary passes [1] their number [2] steps [3] looking for / starting
  // frames [4] ring top-ups [5] periods (hot loops + boundary passes) [6] lane-periods with a frame [7] periods
  unsigned long long prof[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
  const bool profiling = a.prof != nullptr;
  for ( ;; ) {
    const unsigned long long t_a = profiling ? wall_clock64() : 0ull;
    const bool idle = is_lane && L.rec == aa::tok::R_DONE;
    const unsigned long long idle_mask = __ballot( idle );
    bool looked = false;
    if ( idle_mask && !retiring ) {
      if ( backoff ) backoff--;
      else {
        looked = true;
        const int first = __ffsll( static_cast<long long>( idle_mask ) ) - 1;
        uint32_t base = 0, got = 0;
        if ( lane == first ) {
          // (the retire word lives in host memory: a read over the bus -- every 8th look is often enough)
          if ( ( looks++ & 7u ) == 0 && __hip_atomic_load( a.retire, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM ) >= a.gen ) got = 0xFFFFFFFFu;
          else {
            // (one lane per partition: leave lanes for the partitions of the frames drawn -- what is left over draws again next period)
            uint32_t want = static_cast<uint32_t>( __popcll( idle_mask ) );
            if constexpr ( MP ) want = want / a.mp_hint > 1u ? want / a.mp_hint : 1u;
            got = queue_take( a.q, want, a.spread, &base );
          }
        }
        base = __shfl( base, first ); got = __shfl( got, first );
        if ( got == 0xFFFFFFFFu ) { retiring = true; got = 0; }
        const uint32_t rank = static_cast<uint32_t>( __popcll( idle_mask & ( ( 1ull << lane ) - 1ull ) ) );
        const bool mine = idle && rank < got;
        if ( got ) {
          // the slot, the job, the compressed frame, the header kernel's flags: written by other agents / other XCDs, and
          // this CU may hold stale lines of the recycled memory they live in
          __builtin_amdgcn_fence( __ATOMIC_ACQUIRE, "agent" );
          const aa::ParseJob * job = nullptr;
          if ( mine ) job = reinterpret_cast<const aa::ParseJob *>( AA_AT_LOAD( &a.slots[( base + rank ) & a.q->mask] ) );
          if constexpr ( MP ) {
            // the tickets' frames get their lanes (tok::mp_deal): a lane per partition where the wave has them idle
            const uint32_t n_idle = static_cast<uint32_t>( __popcll( idle_mask ) );
            const int my_parts = mine ? static_cast<int>( job->nmb && job->mp_stride ? job->fp.nparts : 1u ) : 1;
            const aa::tok::MpDeal d = aa::tok::mp_deal( n_idle, got, rank, [&]( uint32_t t ) {
              return static_cast<uint32_t>( __shfl( my_parts, nth_set_bit( idle_mask, t ) ) ); } );      // (t is wave-uniform: every lane takes part)
            // the job of my ticket is in the registers of the lane that holds the ticket (rank d.ticket)
            const long long jp = __shfl( static_cast<long long>( reinterpret_cast<uintptr_t>( job ) ), nth_set_bit( idle_mask, d.ticket ) );
            if ( idle && d.any ) {
              const aa::ParseJob * my_job = reinterpret_cast<const aa::ParseJob *>( static_cast<uintptr_t>( jp ) );
              if ( d.n > 1u ) {
                const int owner_lane = nth_set_bit( idle_mask, d.start + aa::tok::mp_owner_partition( my_job ) );
                F = aa::tok::frame_of_partition( my_job, d.part, aa::tok::slice_addr( owner_lane, a.lanes, a.lane_bytes ) );
              } else F = aa::tok::frame_of( my_job );
              if ( my_job->nmb == 0 ) { L.rec = aa::tok::R_DONE; aa::tok::preload( L, smem ); }                  // (never queued; belt and braces)
              else aa::tok::begin_frame<MP>( L, smem, L.base, F );
            }
          } else if ( mine ) {
            F = aa::tok::frame_of( job );
            if ( F.nmb == 0 ) { L.rec = aa::tok::R_DONE; aa::tok::preload( L, smem ); }                        // (never queued; belt and braces)
            else aa::tok::begin_frame<MP>( L, smem, L.base, F );
          }
        } else backoff = 3;               // nothing there: the busy lanes of this wave should not pay for a look every period
      }
    }
    const bool active = is_lane && L.rec != aa::tok::R_DONE;
    if ( !__any( active ) ) {
      if ( retiring ) break;              // (the lanes' frames are through: the stream this grid holds is the host's again)
      if ( looked ) {
        // No lane has a frame and the queue is empty.  The wave does not leave at once: the host hands frames over in bursts,
        // and a grid whose waves left in the gap between two bursts would have to be launched again -- on a worker stream that a
        // few long-running leftovers of the old grid may hold for seconds.  It stays, asleep between looks, for `linger`.
        const unsigned long long now = wall_clock64();
        if ( !idle_since ) idle_since = now | 1ull;
        else if ( now - idle_since > a.linger_ticks ) break;
        for ( int k = 0; k < 16; k++ ) __builtin_amdgcn_s_sleep( 127 );          // ~30 us
      }
      backoff = 0;
      continue;
    }
    idle_since = 0;
    const unsigned long long t_b = profiling ? wall_clock64() : 0ull;
    if ( active ) aa::tok::top_up<MP>( L, smem, F );
    const unsigned long long t_c = profiling ? wall_clock64() : 0ull;
    aa::tok::run_period<PK, MP>( L, smem, F, a.heap, profiling ? prof : nullptr );
    if ( profiling ) {
      const unsigned long long t_d = wall_clock64();
      prof[3] += t_b - t_a; prof[4] += t_c - t_b; prof[5] += t_d - t_c;
      prof[6] += static_cast<unsigned long long>( __popcll( __ballot( active ) ) ); prof[7]++;
      // (waves stay for as long as there is work: the sums go out every 1024 periods, not only when the wave leaves)
      if ( ( prof[7] & 1023ull ) == 0 ) {
        if ( lane == 0 ) for ( int k = 0; k < 8; k++ ) __hip_atomic_fetch_add( &a.prof[k], prof[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
        for ( int k = 0; k < 8; k++ ) prof[k] = 0;
      }
    }
  }
  if ( lane == 0 ) {
    if ( profiling ) for ( int k = 0; k < 8; k++ ) __hip_atomic_fetch_add( &a.prof[k], prof[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT );
    if ( a.cu_slots ) AA_AT_ADD( &a.cu_slots[cu_slot], 0xFFFFFFFFu );
    AA_AT_ADD( a.exited, 1u );
  }
}

// jobs[order[i]] -> the queue, in this order (longest chains first).  `order` lists live frames only: a ticket is a pointer into
// the batch arena, and nobody waits for the lane of a rejected or released frame before the arena is recycled.
__global__ __launch_bounds__( 64 ) void k_enqueue_jobs( aa::TokQueue * q, unsigned long long * slots, const ParseJob * jobs, const uint32_t * order, int n )
{
  __shared__ uint32_t s_base;
  if ( threadIdx.x == 0 ) s_base = AA_AT_ADD( &q->reserve, static_cast<uint32_t>( n ) );
  __syncthreads();
  const uint32_t base = s_base, mask = q->mask;
  for ( int i = threadIdx.x; i < n; i += 64 ) AA_AT_STORE( &slots[( base + i ) & mask], reinterpret_cast<unsigned long long>( &jobs[order[i]] ) );
  __syncthreads();
  if ( threadIdx.x == 0 ) {
    __builtin_amdgcn_fence( __ATOMIC_RELEASE, "agent" );
    asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );                 // (one wave: this waits for every lane's stores)
    while ( AA_AT_LOAD( &q->publish ) != base ) __builtin_amdgcn_s_sleep( 8 );     // in order (earlier reservers are running)
    AA_AT_STORE( &q->publish, base + static_cast<uint32_t>( n ) );
  }
}

// One wave returns chunk numbers to the pool: thread t brings n_mine of them (ids, or first, first + 1, ...).  ONE reservation
// and ONE publication per wave -- threads of a wave must never wait for each other's publication (lock step).
__device__ inline void pool_push_wave( const aa::Heap & H, const AA_GLOBAL uint32_t * ids, uint32_t first, uint32_t n_mine )
{
  __shared__ uint32_t s_off[65];
  __shared__ uint32_t s_pbase;
  s_off[threadIdx.x + 1] = n_mine;
  __syncthreads();
  if ( threadIdx.x == 0 ) {
    s_off[0] = 0;
    for ( int i = 1; i <= 64; i++ ) s_off[i] += s_off[i - 1];
    s_pbase = s_off[64] ? AA_AT_ADD( &H.pool->reserve, s_off[64] ) : 0u;
  }
  __syncthreads();
  const uint32_t b = s_pbase + s_off[threadIdx.x], mask = H.pool->mask, total = s_off[64];
  for ( uint32_t i = 0; i < n_mine; i++ ) AA_AT_STORE( &H.ring[( b + i ) & mask], ids ? ids[i] : first + i );
  __syncthreads();
  if ( threadIdx.x == 0 && total ) {
    __builtin_amdgcn_fence( __ATOMIC_RELEASE, "agent" );
    asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
    while ( AA_AT_LOAD( &H.pool->publish ) != s_pbase ) __builtin_amdgcn_s_sleep( 8 );
    AA_AT_STORE( &H.pool->publish, s_pbase + total );
    AA_AT_ADD_REL( &H.pool->avail, static_cast<int32_t>( total ) );
  }
}
// chunks [first, first + count) of newly mapped heap -> the pool
__global__ __launch_bounds__( 64 ) void k_pool_push_range( aa::Heap heap, uint32_t first, uint32_t count )
{
  const uint32_t per = ( count + 63u ) / 64u, lo = threadIdx.x * per;
  const uint32_t mine = lo >= count ? 0u : ( count - lo < per ? count - lo : per );
  pool_push_wave( heap, nullptr, first + lo, mine );
}
// the chunks of released frames -> the pool: lists[i][0] = count, then the chunk numbers
struct FreeLists { const uint32_t * l[AA_MAX_BATCH]; };
__global__ __launch_bounds__( 64 ) void k_pool_free_lists( aa::Heap heap, const FreeLists lists, int n )
{
  const int i = blockIdx.x * 64 + threadIdx.x;
  const AA_GLOBAL uint32_t * l = ( i < n && lists.l[i] ) ? (const AA_GLOBAL uint32_t *) lists.l[i] : nullptr;
  pool_push_wave( heap, l ? l + 1 : nullptr, 0, l ? l[0] : 0u );
}
// the counters the host steers by, gathered by ONE thread into pinned host memory (a single writer: no torn or reordered view)
__global__ void k_mirror_counters( const aa::TokQueue * q, const aa::CoeffPool * pool, const uint32_t * exited, int n_grids, const unsigned long long * prof, aa_tok_mirror * out, uint32_t seq )
{
  if ( threadIdx.x || blockIdx.x ) return;
  if ( prof ) for ( int k = 0; k < 8; k++ ) out->prof[k] = AA_AT_LOAD( &prof[k] );
  out->q_head = AA_AT_LOAD( &q->head ); out->q_publish = AA_AT_LOAD( &q->publish ); out->q_reserve = AA_AT_LOAD( &q->reserve );
  if ( pool ) { out->pool_avail = AA_AT_LOAD( &pool->avail ); out->pool_starving = AA_AT_LOAD( &pool->starving ); }
  for ( int g = 0; g < n_grids; g++ ) out->exited[g] = AA_AT_LOAD( &exited[g] );
  __builtin_amdgcn_fence( __ATOMIC_RELEASE, "" );
  asm volatile( "s_waitcnt vmcnt(0)" ::: "memory" );
  __hip_atomic_store( &out->seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM );
}

} // namespace

namespace aa {

// ALFALFA_AMD_PARSE_LANES=n: n token lanes per workgroup, whatever that does to occupancy (experiments); 0 / unset: chosen
// per launch by token_launch_shape().
static int parse_lanes_env()
{
  static const int lanes = [] {
    const char * e = getenv( "ALFALFA_AMD_PARSE_LANES" );
    const int v = e ? atoi( e ) : 0;
    return v < 0 ? 0 : ( v > 64 ? 64 : v );
  }();
  return lanes;
}
static int header_lanes() { const int v = parse_lanes_env(); return v ? std::min( v, 16 ) : 16; }      // (k_parse_mb_headers keeps 16 lanes' parameters in LDS)

// Token workgroups stay for as long as there is work, and what they leave of a CU -- LDS, registers -- is all the reconstruction
// kernels (milliseconds each, on the high-priority stream) ever get.  The shape is therefore chosen for the reconstruction
// kernels' sake as much as for the lanes':
//   * `n` workgroups (waves) per CU, default 3 (round 5; 4 = one per SIMD before): a worker wave holds ~245 VGPRs, so with three
//     of them one SIMD of every CU is the reconstruction kernels' alone and the other three keep ~265 free (a loop-filter wave
//     of 256 still fits);
//   * ROUND 6: 30 lanes per workgroup and an LDS request of exactly what they need (33 792 B at 1080p), 3 x 256 workgroups launched.
//     The branch-free step (tok_fsm.hh) made a wave step 24 % faster, and the plateau of the pipelined run became the
//     reconstruction chain beside the workers -- whose pace is the number of its workgroups a CU holds, i.e. the LDS the workers
//     leave: 3 x 37 lanes left 38.5 KB = TWO loop-filter workgroups (19 KB each) or three of k_recon_inter4 (11 KB); 3 x 30 lanes
//     leave 58.6 KB = THREE loop-filter workgroups or five inter ones.  Measured on the driver's command, two runs per shape
//     (profiles/r06_bench_sessions.md): 3 x 37 125.6 / 128.5 M macroblocks/s, 3 x 32 126.8 / 130.4 (52.5 KB: still two loop-filter
//     workgroups), **3 x 30 136.4 / 135.0 / 138.8**, 3 x 27 136.1, 2 x 45 129.7 / 130.7, 2 x 48 134.4, 2 x 59 125.2.  Rounds 3-5
//     padded the request to just over 1 / (n + 1) of the CU's LDS so that an (n + 1)-th workgroup could not land on a CU; with
//     the exact request it could (4 x 33 KB fit), so the host launches no more than n per CU and leaves the placement to the
//     dispatcher (which spreads workgroups by free resources).  A step costs a wave the same whatever its width, so lanes per
//     workgroup trade token capacity (90 chains per CU now, 111 before) against the reconstruction's occupancy one for one.
// ALFALFA_AMD_WGS_PER_CU / ALFALFA_AMD_MAX_LANES / ALFALFA_AMD_LDS_EXACT=0 (round 5's padded request) / ALFALFA_AMD_WGS_CAP_PER_CU: experiments.
constexpr uint32_t kLdsPerCu = 160u * 1024u, kLdsGranule = 512u;
static uint32_t env_u32( const char * name, uint32_t dflt ) { const char * e = getenv( name ); return e ? static_cast<uint32_t>( atoi( e ) ) : dflt; }
static const uint32_t kWgsPerCu = std::max( 1u, std::min( 16u, env_u32( "ALFALFA_AMD_WGS_PER_CU", 3u ) ) );
static const uint32_t kMaxLanes = std::max( 1u, std::min( 64u, env_u32( "ALFALFA_AMD_MAX_LANES", 30u ) ) );
struct TokenShape { int lanes; uint32_t lds; int per_cu; };
static TokenShape token_launch_shape( uint32_t lane_bytes )
{
  for ( uint32_t n = kWgsPerCu; n >= 1; n-- ) {
    const uint32_t request = std::min( 65536u, ( kLdsPerCu / ( n + 1 ) / kLdsGranule + 1 ) * kLdsGranule );      // smallest request of which n + 1 do not fit
    if ( request < tok::kTablesBytes + lane_bytes || request * n > kLdsPerCu ) continue;
    const int lanes = static_cast<int>( std::min<uint32_t>( kMaxLanes, ( request - tok::kTablesBytes ) / lane_bytes ) );
    return { lanes, request, static_cast<int>( n ) };
  }
  return { 0, 0, 0 };
}

int launch_parse_mb_headers( const ParseJob * jobs, const uint32_t * order, int n, void * stream )
{
  const int lanes = header_lanes();
  hipLaunchKernelGGL( k_parse_mb_headers, dim3( ( n + lanes - 1 ) / lanes ), dim3( 64 ), 0, static_cast<hipStream_t>( stream ), jobs, order, n, lanes );
  return static_cast<int>( hipGetLastError() );
}

int launch_segment_fixup( const ParseJob * jobs, const aa_seg_stream * streams, int n_streams, const uint32_t * order, void * stream )
{
  hipLaunchKernelGGL( k_segment_fixup, dim3( n_streams ), dim3( 256 ), 0, static_cast<hipStream_t>( stream ), jobs, streams, order );
  return static_cast<int>( hipGetLastError() );
}

// Shape of a worker grid for slices of `lane_bytes`: lanes per workgroup, LDS request, workgroups the GPU holds at once
void token_worker_shape( uint32_t lane_bytes, int n_cus, int * lanes_out, uint32_t * lds_out, int * wgs_per_cu_out )
{
  int lanes = parse_lanes_env();
  uint32_t lds = 0;
  if ( lanes ) {
    if ( static_cast<uint32_t>( lanes ) * lane_bytes + tok::kTablesBytes > 65536u ) lanes = static_cast<int>( ( 65536u - tok::kTablesBytes ) / lane_bytes );
    lds = static_cast<uint32_t>( lanes ) * lane_bytes + tok::kTablesBytes;
  } else {
    const TokenShape sh = token_launch_shape( lane_bytes );
    lanes = sh.lanes; lds = sh.lds;
  }
  (void) n_cus;
  // (round 6 default: a workgroup asks for what its lanes need and no more, and the host launches kWgsPerCu workgroups per CU --
  // see the note above token_launch_shape; ALFALFA_AMD_LDS_EXACT=0 brings back the padded request that keeps an (n + 1)-th out)
  static const bool lds_exact = env_u32( "ALFALFA_AMD_LDS_EXACT", 1u ) != 0u;
  static const uint32_t wgs_cap = env_u32( "ALFALFA_AMD_WGS_CAP_PER_CU", lds_exact && !parse_lanes_env() ? kWgsPerCu : 0u );
  if ( lds_exact && lanes > 0 ) lds = ( ( tok::kTablesBytes + static_cast<uint32_t>( lanes ) * lane_bytes + kLdsGranule - 1u ) / kLdsGranule ) * kLdsGranule;
  *lanes_out = lanes; *lds_out = lds;
  int per_cu = lds ? static_cast<int>( kLdsPerCu / lds ) : 0;
  if ( wgs_cap && per_cu > static_cast<int>( wgs_cap ) ) per_cu = static_cast<int>( wgs_cap );
  *wgs_per_cu_out = per_cu;
}

int launch_token_workers( TokQueue * q, unsigned long long * slots, const Heap & heap, uint32_t * exited, const uint32_t * retire, uint32_t gen, uint32_t spread,
                          unsigned long long * prof, unsigned long long linger_ticks, int wgs, int lanes, uint32_t lane_bytes, uint32_t lds, bool packed, uint32_t mp_hint, void * stream,
                          uint32_t * cu_slots, uint32_t cu_cap )
{
  if ( lanes < 1 || wgs < 1 ) return static_cast<int>( hipErrorInvalidValue );
  WorkerArgs a;
  a.q = q; a.slots = slots; a.heap = heap; a.exited = exited; a.retire = retire; a.gen = gen; a.spread = spread ? spread : 1u; a.prof = prof; a.linger_ticks = linger_ticks; a.lanes = lanes; a.lane_bytes = lane_bytes;
  a.mp_hint = mp_hint ? mp_hint : 1u;
  a.cu_slots = cu_cap ? cu_slots : nullptr; a.cu_cap = cu_cap;
  a.sink = reinterpret_cast<int16_t *>( cu_slots + AA_CU_SLOTS / 2 );       // (the upper half of the counters' buffer: slots are < 2048)
  // (mp_hint != 0: the context allows a lane per partition)
  if ( mp_hint ) {
    if ( packed ) hipLaunchKernelGGL( ( k_token_workers<true, true> ), dim3( wgs ), dim3( 64 ), lds, static_cast<hipStream_t>( stream ), a );
    else hipLaunchKernelGGL( ( k_token_workers<false, true> ), dim3( wgs ), dim3( 64 ), lds, static_cast<hipStream_t>( stream ), a );
  } else {
    if ( packed ) hipLaunchKernelGGL( ( k_token_workers<true, false> ), dim3( wgs ), dim3( 64 ), lds, static_cast<hipStream_t>( stream ), a );
    else hipLaunchKernelGGL( ( k_token_workers<false, false> ), dim3( wgs ), dim3( 64 ), lds, static_cast<hipStream_t>( stream ), a );
  }
  return static_cast<int>( hipGetLastError() );
}

int launch_enqueue_jobs( TokQueue * q, unsigned long long * slots, const ParseJob * jobs, const uint32_t * order, int n, void * stream )
{
  hipLaunchKernelGGL( k_enqueue_jobs, dim3( 1 ), dim3( 64 ), 0, static_cast<hipStream_t>( stream ), q, slots, jobs, order, n );
  return static_cast<int>( hipGetLastError() );
}

int launch_pool_push_range( const Heap & heap, uint32_t first, uint32_t count, void * stream )
{
  hipLaunchKernelGGL( k_pool_push_range, dim3( 1 ), dim3( 64 ), 0, static_cast<hipStream_t>( stream ), heap, first, count );
  return static_cast<int>( hipGetLastError() );
}

int launch_pool_free_lists( const Heap & heap, const uint32_t * const * lists, int n, void * stream )
{
  for ( int base = 0; base < n; base += AA_MAX_BATCH ) {
    FreeLists fl;
    const int cnt = std::min( AA_MAX_BATCH, n - base );
    for ( int i = 0; i < AA_MAX_BATCH; i++ ) fl.l[i] = i < cnt ? lists[base + i] : nullptr;
    hipLaunchKernelGGL( k_pool_free_lists, dim3( ( cnt + 63 ) / 64 ), dim3( 64 ), 0, static_cast<hipStream_t>( stream ), heap, fl, cnt );
    if ( hipError_t e = hipGetLastError() ) return static_cast<int>( e );
  }
  return 0;
}

int launch_mirror_counters( const TokQueue * q, const CoeffPool * pool, const uint32_t * exited, int n_grids, const unsigned long long * prof, aa_tok_mirror * out, uint32_t seq, void * stream )
{
  hipLaunchKernelGGL( k_mirror_counters, dim3( 1 ), dim3( 64 ), 0, static_cast<hipStream_t>( stream ), q, pool, exited, n_grids, prof, out, seq );
  return static_cast<int>( hipGetLastError() );
}

} // namespace aa
