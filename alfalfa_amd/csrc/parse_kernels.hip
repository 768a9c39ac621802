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

namespace {

using aa::ParseJob;

// `order`: which job each slot of the launch takes -- the host sorts the frames of a batch by the length of their chains so
// that the lanes of a wave finish together (a wave lasts as long as its longest lane)
__global__ __launch_bounds__( 64 ) void k_parse_mb_headers( const ParseJob * jobs, const uint32_t * order, int n, int lanes )
{
  __builtin_amdgcn_s_setprio( 2 );      // short chains the long token chains wait for: ahead of them at the issue arbiter

  const int lane = threadIdx.x;
  const int slot = blockIdx.x * lanes + lane;
  if ( lane >= lanes || slot >= n ) return;
  const ParseJob & J = jobs[order[slot]];
  if ( J.nmb == 0 ) return;                      // a frame the host header pre-pass rejected
  const aa::FrameParams & fp = J.fp;
  aa::BoolReader32 bd;
  aa::BoolState st; st.bitpos = fp.bd_bitpos; st.range = fp.bd_range; st.active = fp.bd_active;
  bd.resume( J.data + fp.first_off, fp.first_size, st );
  const unsigned mbw = fp.mbw, mbh = fp.mbh;
  const unsigned words_per_row = ( mbw + 63 ) / 64;
  uint32_t intra = 0, split = 0;
  unsigned mi = 0;
  for ( unsigned row = 0; row < mbh; row++ ) {
    unsigned long long word = 0;
    for ( unsigned col = 0; col < mbw; col++, mi++ ) {
      const uint8_t flags = aa::parse_mb_header( bd, fp, J.mbs, mi, col, row, static_cast<uint8_t *>( nullptr ) );
      J.mbflags[mi] = flags;
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

__global__ __launch_bounds__( 64 ) void k_parse_tokens( const ParseJob * jobs, const uint32_t * order, int n, int lanes, uint32_t lane_bytes )
{
  extern __shared__ __attribute__( ( aligned( 16 ) ) ) uint8_t smem[];
  const int lane = threadIdx.x;
  const int slot = blockIdx.x * lanes + lane;
  const int j = static_cast<int>( order[lane < lanes && slot < n ? slot : 0] );
  const bool active = lane < lanes && slot < n && jobs[j].nmb != 0;
  for ( uint32_t k = lane; k < aa::tok::kTablesBytes / 4; k += 64 ) reinterpret_cast<uint32_t *>( smem )[k] = aa::tok::table_word( k );
  __syncthreads();
  aa::tok::Lane L;
  aa::tok::Frame F = aa::tok::frame_of( &jobs[active ? j : 0] );
  L.rec = aa::tok::R_DONE;
  L.base = aa::tok::kTablesBytes;
  L.pend_wpos = L.pend_mwpos = aa::tok::kNoPend;
  if ( active ) aa::tok::begin_frame( L, smem, aa::tok::kTablesBytes + static_cast<uint32_t>( lane ) * lane_bytes, F );
  for ( ;; ) {
    if ( active ) aa::tok::top_up( L, smem, F );
    if ( !__any( L.rec != aa::tok::R_DONE ) ) break;
    aa::tok::run_period( L, smem, F );
  }
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
static int header_lanes() { const int v = parse_lanes_env(); return v ? v : 16; }

// Token workgroups live for seconds and a CU holds as many as its LDS takes, so the LDS they leave is all that the
// reconstruction kernels (milliseconds, launched on the high-priority stream) can start in while a parse is running.  The
// launch shape -- n workgroups of `lanes` lanes per CU -- is chosen for the most lanes per CU with `kLdsReserve` bytes left
// free on every CU: the LDS request is padded so that an (n + 1)-th workgroup does not fit.  Measured on the benchmark
// (profiles/r02_pipeline_experiments.md) the reserve buys nothing -- a pipelined run leaves ~30 % of the token slots empty at any
// time, reconstruction runs there -- so it defaults to 0; ALFALFA_AMD_LDS_RESERVE_KB=24 keeps a loop-filter workgroup's worth.
// (A step costs the wave the same whatever its width, but the rarer paths -- a coefficient emitted, a block or macroblock ended
// -- run whenever ANY lane needs them: beyond ~24 lanes a wave spends most steps in them.)
constexpr uint32_t kLdsPerCu = 160u * 1024u, kLdsGranule = 512u;
static uint32_t env_u32( const char * name, uint32_t dflt ) { const char * e = getenv( name ); return e ? static_cast<uint32_t>( atoi( e ) ) : dflt; }
static const uint32_t kLdsReserve = env_u32( "ALFALFA_AMD_LDS_RESERVE_KB", 0u ) * 1024u;      // (the two knobs: experiments)
static const uint32_t kMaxLanes = std::max( 1u, std::min( 64u, env_u32( "ALFALFA_AMD_MAX_LANES", 24u ) ) );
struct TokenShape { int lanes; uint32_t lds; };
static TokenShape token_launch_shape( uint32_t lane_bytes )
{
  TokenShape best { 0, 0 };
  int best_total = 0;
  for ( uint32_t n = 1; n <= 16; n++ ) {
    const uint32_t maxp = std::min<uint32_t>( 65536u, ( kLdsPerCu - kLdsReserve ) / n / kLdsGranule * kLdsGranule );
    if ( maxp < tok::kTablesBytes + lane_bytes ) break;
    const int lanes = static_cast<int>( std::min<uint32_t>( kMaxLanes, ( maxp - tok::kTablesBytes ) / lane_bytes ) );
    const uint32_t need = ( static_cast<uint32_t>( lanes ) * lane_bytes + tok::kTablesBytes + kLdsGranule - 1 ) / kLdsGranule * kLdsGranule;
    const uint32_t excl = ( kLdsPerCu / ( n + 1 ) / kLdsGranule + 1 ) * kLdsGranule;      // smallest request of which n + 1 do not fit
    const uint32_t p = std::max( need, excl );
    if ( p > maxp ) continue;
    const int total = static_cast<int>( n ) * lanes;
    if ( total >= best_total ) { best_total = total; best = { lanes, p }; }                // ties: more, narrower workgroups (n ascending)
  }
  return best;
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

int launch_parse_tokens( const ParseJob * jobs, const uint32_t * order, int n, int max_mbw, int max_nparts, void * stream )
{
  const uint32_t lane_bytes = tok::lane_lds_bytes( static_cast<uint32_t>( max_mbw ), max_nparts > 1 );
  int lanes = parse_lanes_env();
  uint32_t lds = 0;
  if ( lanes ) {
    if ( static_cast<uint32_t>( lanes ) * lane_bytes + tok::kTablesBytes > 65536u ) lanes = static_cast<int>( ( 65536u - tok::kTablesBytes ) / lane_bytes );
    lds = static_cast<uint32_t>( lanes ) * lane_bytes + tok::kTablesBytes;
  } else {
    const TokenShape sh = token_launch_shape( lane_bytes );
    lanes = sh.lanes; lds = sh.lds;
  }
  if ( lanes < 1 ) return static_cast<int>( hipErrorInvalidValue );
  hipLaunchKernelGGL( k_parse_tokens, dim3( ( n + lanes - 1 ) / lanes ), dim3( 64 ), lds, static_cast<hipStream_t>( stream ), jobs, order, n, lanes, lane_bytes );
  return static_cast<int>( hipGetLastError() );
}

} // namespace aa
