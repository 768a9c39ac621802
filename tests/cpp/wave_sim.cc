// Test harness (not product code): one WAVE of token workers replayed on the host -- `lanes` lanes in lock step over the
// same statements the GPU lanes run (alfalfa_amd/csrc/tok_fsm.hh), sharing what a workgroup shares (the tables in "LDS", the
// job queue, ONE coefficient pool that can be made scarce), with the wave-level control flow of k_token_workers / run_period
// written out over the lanes (ballots become loops).  What tests/cpp/fsm_sim.cc cannot see, this does: lanes at different
// places of different frames of different streams in one period loop, a boundary pass while wave-mates decode, lanes waiting
// for a chunk while others finish and give theirs back, frames handed back (TOK_NO_MEMORY) and run again, jobs arriving while
// the wave is busy.  Built by tests/test_wave_sim.py with plain g++:
//   g++ -shared wave_sim.cc ../../alfalfa_amd/csrc/parser.cpp
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <vector>

#include "../../alfalfa_amd/csrc/bool_reader.hh"
#include "../../alfalfa_amd/csrc/parser.hh"
#include "../../alfalfa_amd/csrc/tok_fsm.hh"
#include "../../alfalfa_amd/csrc/coeff_pack.hh"

namespace {
int16_t g_sink[64] = {};     // where lanes without a chunk store (tok::step: the store is unconditional)

void * aligned( size_t bytes )
{
  void * p = nullptr;
  if ( posix_memalign( &p, 256, bytes ? bytes : 256 ) ) return nullptr;
  std::memset( p, 0xA5, bytes ? bytes : 256 );
  return p;
}

// one frame as the runtime hands it to the device: the job record and the buffers it points at
struct Job {
  aa::ParseJob J;
  aa::FrameSummary sum;
  aa_frame_header hdr;
  uint32_t nmb = 0;
  int attempts = 0;
  bool collected = false;
  std::vector<void *> owned;
  Job() { std::memset( &J, 0, sizeof J ); std::memset( &sum, 0, sizeof sum ); std::memset( &hdr, 0, sizeof hdr ); }
  ~Job() { for ( void * p : owned ) free( p ); }
  template <class T> T * buf( size_t bytes ) { void * p = aligned( bytes ); owned.push_back( p ); return static_cast<T *>( p ); }
};

struct Stream {
  aa::Parser parser;
  std::vector<uint8_t> segmap;
  Stream( uint16_t w, uint16_t h ) : parser( w, h ), segmap( size_t( parser.mb_width() ) * parser.mb_height(), 3 ) {}
};

// header pre-pass (the product's Parser::parse_header), k_parse_mb_headers' loop, k_segment_fixup's loop: as in fsm_sim.cc
int prepare( Stream & S, const uint8_t * data, size_t size, bool packed, bool mp, Job & B )
{
  aa::ParseJob & J = B.J;
  try { S.parser.parse_header( data, size, B.hdr, J.fp ); }
  catch ( const aa::ParseError & e ) { return e.code; }
  const uint32_t nmb = uint32_t( J.fp.mbw ) * J.fp.mbh;
  B.nmb = nmb;
  J.size = uint32_t( size ); J.data_padded = ( J.size + 15 ) & ~15u;
  uint8_t * dev_data = B.buf<uint8_t>( J.data_padded );
  std::memcpy( dev_data, data, size );
  J.data = dev_data;
  J.nmb = nmb; J.flags_padded = ( nmb + 15 ) & ~15u;
  // one lane per partition allowed: a second copy of the flags, partition by partition (as the runtime lays it out)
  J.mp_stride = mp && J.fp.nparts > 1 ? aa::mp_flag_stride( J.fp.mbw, J.fp.mbh, J.fp.nparts ) : 0u;
  J.mbflags = B.buf<uint8_t>( J.flags_padded + size_t( J.fp.nparts ) * J.mp_stride );
  J.mbs = B.buf<aa_mb_info>( nmb * sizeof( aa_mb_info ) );
  J.chunk_list = B.buf<uint32_t>( size_t( aa::chunk_list_entries( nmb, J.mp_stride ? J.fp.nparts : 1u ) ) * 4 );
  J.packed_pos = packed ? B.buf<uint32_t>( size_t( nmb ) * 4 ) : nullptr;
  const unsigned words_per_row = ( J.fp.mbw + 63 ) / 64;
  J.intra_rows = B.buf<unsigned long long>( size_t( words_per_row ) * J.fp.mbh * 8 );
  J.summary = &B.sum;
  aa::BoolReader32 bd;
  aa::BoolState st; st.bitpos = J.fp.bd_bitpos; st.range = J.fp.bd_range; st.active = J.fp.bd_active;
  bd.resume( J.data + J.fp.first_off, J.fp.first_size, st );
  unsigned mi = 0; uint32_t intra = 0, split = 0;
  for ( unsigned row = 0; row < J.fp.mbh; row++ ) {
    unsigned long long word = 0;
    for ( unsigned col = 0; col < J.fp.mbw; col++, mi++ ) {
      const uint8_t flags = aa::parse_mb_header( bd, J.fp, aa::kHeaderTables, J.mbs, mi, col, row, static_cast<uint8_t *>( nullptr ) );
      J.mbflags[mi] = flags;
      if ( J.mp_stride ) J.mbflags[aa::mp_flag_index( J, row, col )] = flags;
      if ( !( flags & AA_MB_INTER ) ) { intra++; word |= 1ull << ( col & 63 ); }
      else if ( J.mbs[mi].y_mode == aa::SPLITMV ) split = 1;
      if ( ( col & 63 ) == 63 || col + 1 == J.fp.mbw ) { J.intra_rows[row * words_per_row + ( col >> 6 )] = word; word = 0; }
    }
  }
  B.sum.num_intra_mbs = intra; B.sum.has_split = split;
  if ( J.fp.seg_enabled ) {
    if ( S.parser.segment_map_reset() ) std::memset( S.segmap.data(), 3, S.segmap.size() );
    for ( uint32_t k = 0; k < nmb; k++ ) aa::segment_fixup( J.fp, J.mbs[k], S.segmap[k] );
  }
  return 0;
}

// the frame's own view of its records (what aa_stream_read_records gives): blocks back to back, coeff_index from 0
bool collect( const Job & B, const aa::Heap & H, bool packed, aa_mb_info * mbs, int16_t * coeffs )
{
  const aa::ParseJob & J = B.J;
  std::memcpy( mbs, J.mbs, B.nmb * sizeof( aa_mb_info ) );
  uint32_t running = 0;
  for ( uint32_t mi = 0; mi < B.nmb; mi++ ) {
    const uint32_t nblk = aa::pack::blocks_of( mbs[mi].nz_mask );
    if ( nblk ) {
      if ( running + nblk > B.sum.num_coeff_blocks ) return false;
      if ( packed ) {
        const uint32_t pos = J.packed_pos[mi];
        if ( ( pos >> 15 ) >= B.sum.num_chunks ) return false;
        const int16_t * w = H.base + aa::pack::word_offset( pos, J.chunk_list );
        const uint32_t used = aa::pack::expand_macroblock( w, mbs[mi].nz_mask, coeffs + size_t( running ) * 16 );
        if ( ( pos & ( aa::kChunkWords - 1u ) ) + used > aa::kChunkWords ) return false;
      } else {
        const uint32_t c = mbs[mi].coeff_index / aa::kChunkBlocks, o = mbs[mi].coeff_index % aa::kChunkBlocks;
        bool mine = false;
        for ( uint32_t k = 0; k < B.sum.num_chunks; k++ ) if ( J.chunk_list[1 + k] == c ) mine = true;
        if ( !mine || o + nblk > aa::kChunkBlocks ) return false;
        std::memcpy( coeffs + size_t( running ) * 16, H.base + size_t( mbs[mi].coeff_index ) * 16, size_t( nblk ) * 32 );
      }
    }
    mbs[mi].coeff_index = running; mbs[mi].reserved = 0;
    running += nblk;
  }
  return running == B.sum.num_coeff_blocks;
}

template <bool PK, bool MP>
void wave_period( std::vector<aa::tok::Lane> & L, std::vector<aa::tok::Frame> & F, const std::vector<int> & job_of, uint8_t * smem, const aa::Heap & H,
                  uint64_t * boundary_passes )
{
  using namespace aa::tok;
  const size_t n = L.size();
  auto any = [&]( auto pred ) { for ( size_t k = 0; k < n; k++ ) if ( job_of[k] >= 0 && pred( L[k] ) ) return true; return false; };
  // run_period, written out over the lanes of the wave (AA_ANY = a ballot over them)
  uint32_t it = 0;
  while ( it < kPeriod ) {
    if ( any( []( const Lane & l ) { return at_boundary<MP>( l ); } ) ) {
      for ( size_t k = 0; k < n; k++ ) if ( job_of[k] >= 0 && at_boundary<MP>( L[k] ) ) macroblock_boundary<PK, MP>( L[k], smem, F[k], H );
      ( *boundary_passes )++;
      it++;
      if ( !any( []( const Lane & l ) { return l.rec < R_MBDONE; } ) ) break;
    }
    for ( ;; ) {
      for ( uint32_t g = 0; g < kBendEvery; g++ )
        for ( size_t k = 0; k < n; k++ ) step<PK, MP>( L[k], smem, F[k] );       // (every lane, with or without a frame: the step has no condition)
      it += kBendEvery;
      if ( any( []( const Lane & l ) { return l.rec == R_BEND; } ) ) {
        for ( size_t k = 0; k < n; k++ ) if ( job_of[k] >= 0 ) block_end<PK, MP>( L[k], smem, F[k], H );
        if ( any( []( const Lane & l ) { return l.rec == R_MBDONE; } ) ) break;
      }
      if ( it >= kPeriod ) break;
    }
  }
  for ( size_t k = 0; k < n; k++ ) if ( job_of[k] >= 0 && L[k].rec != R_DONE ) L[k].steps += it;
}

} // namespace

extern "C" {

// n_streams streams of one size, frames_per_stream[s] frames each (data / sizes: stream-major).  The jobs of all frames go to
// the queue in an order drawn from `seed` that keeps every stream's frames in order (the device parses a stream's frames in
// any order -- only the pre-pass is sequential -- but the runtime enqueues them in order), `burst` at a time, the next burst
// when the wave has run `burst_gap` periods.  pool_chunks: chunks the pool holds in all (0: plenty).
// Outputs, frame by frame in stream-major order: mbs_out [frames * nmb], coeffs_out [frames * (25 * nmb) * 16],
// hdr_out [frames]; stats: [0] periods [1] boundary passes [2] frames handed back for lack of memory [3] peak chunks out
// [4] lanes that were busy in the busiest period.  -> 0 ok; 1..: a frame's pre-pass failed; 100: a check failed; 101: jobs are
// missing although no lane is busy; 102: a frame was handed back more than 64 times; 103: too many periods
int wave_sim_run( uint16_t w, uint16_t h, int n_streams, const int * frames_per_stream, const uint8_t * const * data, const size_t * sizes,
                  int lanes, uint32_t pool_chunks, int packed, int mp, int mp_hint, uint32_t seed, int burst, int burst_gap,
                  aa_frame_header * hdr_out, aa_mb_info * mbs_out, int16_t * coeffs_out, uint64_t * stats )
{
  using namespace aa::tok;
  std::vector<std::unique_ptr<Stream>> streams;
  std::vector<std::unique_ptr<Job>> jobs;
  std::vector<int> first_job( n_streams + 1, 0 );
  bool multi = false;
  {
    int f = 0;
    for ( int s = 0; s < n_streams; s++ ) {
      streams.emplace_back( new Stream( w, h ) );
      first_job[s] = f;
      for ( int k = 0; k < frames_per_stream[s]; k++, f++ ) {
        jobs.emplace_back( new Job );
        if ( const int rc = prepare( *streams[s], data[f], sizes[f], packed != 0, mp != 0, *jobs.back() ) ) return rc > 0 ? rc : -rc;
        multi = multi || jobs.back()->J.fp.nparts > 1;
      }
    }
    first_job[n_streams] = f;
  }
  const int n_jobs = static_cast<int>( jobs.size() );
  const uint32_t nmb = jobs.empty() ? 0u : jobs[0]->nmb;
  const uint32_t mbw = jobs.empty() ? 0u : jobs[0]->J.fp.mbw;
  // queue order: interleave the streams at random, each stream's frames in order
  std::vector<int> order;
  {
    std::vector<int> next( first_job.begin(), first_job.end() - 1 );
    uint32_t x = seed * 2654435761u + 12345u;
    while ( static_cast<int>( order.size() ) < n_jobs ) {
      x = x * 1664525u + 1013904223u;
      int s = static_cast<int>( ( x >> 8 ) % static_cast<uint32_t>( n_streams ) );
      for ( int tries = 0; tries < n_streams && next[s] >= first_job[s + 1]; tries++ ) s = ( s + 1 ) % n_streams;
      order.push_back( next[s]++ );
    }
  }
  // the heap and its pool
  const uint32_t worst = aa::chunk_list_entries( nmb, mp ? 8u : 1u ) - 1u;
  const uint32_t heap_chunks = pool_chunks ? pool_chunks : worst * static_cast<uint32_t>( std::min( n_jobs, lanes ) + 2 );
  int16_t * heap_mem = static_cast<int16_t *>( aligned( size_t( heap_chunks ) * aa::kChunkBlocks * 32 ) );
  uint32_t ring_entries = 1; while ( ring_entries < heap_chunks ) ring_entries <<= 1;
  std::vector<uint32_t> ring( ring_entries, 0xDEADBEEFu );
  aa::CoeffPool pool; std::memset( &pool, 0, sizeof pool ); pool.mask = ring_entries - 1;
  aa::Heap H; H.base = heap_mem; H.pool = &pool; H.ring = ring.data();
  {
    uint32_t stride = 1;
    for ( uint32_t c : { 7u, 11u, 13u, 5u, 3u } ) if ( heap_chunks % c ) { stride = c; break; }
    for ( uint32_t k = 0; k < heap_chunks; k++ ) pool_push( H, nullptr, ( k * stride + 2u ) % heap_chunks, 1 );
  }
  // the workgroup's LDS: tables + one slice per lane
  const uint32_t lane_bytes = lane_lds_bytes( mbw, multi, mp );
  std::vector<uint8_t> store( kTablesBytes + size_t( lanes ) * lane_bytes + 16 );
  uint8_t * smem = reinterpret_cast<uint8_t *>( ( reinterpret_cast<uintptr_t>( store.data() ) + 15 ) & ~uintptr_t( 15 ) );
  std::memset( smem, 0xA5, kTablesBytes + size_t( lanes ) * lane_bytes );
  for ( uint32_t k = 0; k < kTablesBytes / 4; k++ ) reinterpret_cast<uint32_t *>( smem )[k] = table_word( k );
  std::vector<Lane> L( lanes );
  std::vector<Frame> F( lanes );
  std::vector<int> job_of( lanes, -1 );
  for ( int k = 0; k < lanes; k++ ) {
    std::memset( static_cast<void *>( &L[k] ), 0xA5, sizeof( Lane ) );
    init_lane( L[k], ring_addr( static_cast<uint32_t>( k ) ), slice_addr( static_cast<uint32_t>( k ), static_cast<uint32_t>( lanes ), lane_bytes ), &g_sink[k & 63] );
    preload( L[k], smem );
  }
  std::deque<int> queue, held;
  size_t published = 0;
  int done = 0, bad = 0;
  uint64_t periods = 0, boundary_passes = 0, handed_back = 0, peak_out = 0, peak_busy = 0, idle_periods = 0, mp_frames = 0, parked = 0;
  auto out_index = [&]( int j ) { return static_cast<size_t>( j ); };
  while ( done < n_jobs ) {
    // frames that were handed back go to the queue again one at a time, when the pool has what such a frame may take (the
    // runtime waits for exactly that before it runs a frame again: resolve_summary)
    if ( !held.empty() ) {
      const aa::ParseJob & HJ = jobs[held.front()]->J;
      const int32_t need = static_cast<int32_t>( aa::chunk_list_entries( HJ.nmb, HJ.mp_stride ? HJ.fp.nparts : 1u ) ) - 1;
      bool running = false;
      for ( int k = 0; k < lanes; k++ ) if ( job_of[k] >= 0 ) running = true;
      if ( pool.avail >= std::min<int32_t>( need, static_cast<int32_t>( heap_chunks ) ) ) { queue.push_back( held.front() ); held.pop_front(); }
      else if ( !running && queue.empty() && published == order.size() ) { free( heap_mem ); return 102; }       // nothing left that could give chunks back
    }
    // jobs arrive in bursts
    if ( published < order.size() && ( periods % static_cast<uint64_t>( std::max( 1, burst_gap ) ) == 0 || queue.empty() ) )
      for ( int b = 0; b < std::max( 1, burst ) && published < order.size(); b++ ) queue.push_back( order[published++] );
    // idle lanes take the next jobs (queue_take: all idle lanes of the wave at once, first come first served)
    {
      // k_token_workers' take: the wave asks for as many tickets as it has idle lanes (mp: idle / hint, so that lanes are left for
      // partitions), then deals its idle lanes out to the tickets' frames (tok::mp_deal, the device's own statements)
      std::vector<int> idle;
      for ( int k = 0; k < lanes; k++ ) if ( job_of[k] < 0 ) idle.push_back( k );
      uint32_t want = static_cast<uint32_t>( idle.size() );
      if ( mp && want ) want = std::max( 1u, want / static_cast<uint32_t>( std::max( 1, mp_hint ) ) );
      const uint32_t got = std::min<uint32_t>( want, static_cast<uint32_t>( queue.size() ) );
      std::vector<int> ticket( queue.begin(), queue.begin() + got );
      queue.erase( queue.begin(), queue.begin() + got );
      for ( uint32_t t = 0; t < got; t++ ) jobs[ticket[t]]->attempts++;
      auto parts_of = [&]( uint32_t t ) { const aa::ParseJob & J = jobs[ticket[t]]->J; return J.nmb && J.mp_stride ? uint32_t( J.fp.nparts ) : 1u; };
      for ( uint32_t rank = 0; rank < idle.size() && got; rank++ ) {
        const MpDeal d = mp ? mp_deal( static_cast<uint32_t>( idle.size() ), got, rank, parts_of )
                            : MpDeal { rank, 0, 1, rank, rank < got };
        if ( !d.any ) continue;
        const int q = idle[rank], j = ticket[d.ticket];
        job_of[q] = j;
        if ( d.n > 1 ) {
          F[q] = frame_of_partition( &jobs[j]->J, d.part, L[idle[d.start + mp_owner_partition( &jobs[j]->J )]].base );
          if ( d.part == 0 ) mp_frames++;
        } else F[q] = frame_of( &jobs[j]->J );
        if ( mp ) begin_frame<true>( L[q], smem, L[q].base, F[q] ); else begin_frame<false>( L[q], smem, L[q].base, F[q] );
      }
    }
    uint64_t busy = 0;
    for ( int k = 0; k < lanes; k++ ) if ( job_of[k] >= 0 ) { if ( mp ) top_up<true>( L[k], smem, F[k] ); else top_up<false>( L[k], smem, F[k] ); busy++; }
    peak_busy = std::max( peak_busy, busy );
    if ( !busy ) { if ( ++idle_periods > 1000 ) { free( heap_mem ); return 101; } periods++; continue; }
    idle_periods = 0;
    if ( mp ) { if ( packed ) wave_period<true, true>( L, F, job_of, smem, H, &boundary_passes ); else wave_period<false, true>( L, F, job_of, smem, H, &boundary_passes ); }
    else { if ( packed ) wave_period<true, false>( L, F, job_of, smem, H, &boundary_passes ); else wave_period<false, false>( L, F, job_of, smem, H, &boundary_passes ); }
    periods++;
    {
      const int64_t out_now = static_cast<int64_t>( heap_chunks ) - pool.avail;
      peak_out = std::max<uint64_t>( peak_out, out_now > 0 ? static_cast<uint64_t>( out_now ) : 0u );
    }
    for ( int k = 0; k < lanes; k++ ) if ( job_of[k] >= 0 && L[k].rec == R_PARK ) parked++;
    // frames that are through: what the host does when it sees `done`
    for ( int k = 0; k < lanes; k++ ) {
      if ( job_of[k] < 0 || L[k].rec != R_DONE ) continue;
      const int j = job_of[k];
      Job & B = *jobs[j];
      job_of[k] = -1;
      if ( F[k].mp_P > 1 ) {                          // one lane per partition: the frame is through when its LAST lane is
        if ( !B.sum.done || B.collected ) continue;
        B.collected = true;
      }
      if ( !B.sum.done || B.J.chunk_list[0] != B.sum.num_chunks ) { bad = 1; done++; continue; }
      if ( B.sum.status == aa::TOK_NO_MEMORY ) {
        // handed back: its chunks return to the pool, the frame goes to the queue again (resolve_summary)
        pool_push( H, B.J.chunk_list + 1, 0, B.sum.num_chunks );
        handed_back++;
        if ( B.attempts > 64 ) { free( heap_mem ); return 102; }
        B.sum.done = 0; B.sum.status = 0; B.sum.num_chunks = 0; B.collected = false;
        held.push_back( j );          // (resolve_summary: run again when the pool holds what the frame may need)
        continue;
      }
      if ( B.sum.status != aa::TOK_OK ) { bad = 1; done++; continue; }
      B.hdr.num_coeff_blocks = B.sum.num_coeff_blocks;
      B.hdr.num_intra_mbs = B.sum.num_intra_mbs;
      B.hdr.has_intra_mb = B.sum.num_intra_mbs != 0;
      hdr_out[out_index( j )] = B.hdr;
      if ( !collect( B, H, packed != 0, mbs_out + out_index( j ) * nmb, coeffs_out + out_index( j ) * size_t( 25 ) * nmb * 16 ) ) bad = 1;
      // every chunk number distinct
      for ( uint32_t a = 0; a < B.sum.num_chunks; a++ ) for ( uint32_t b = 0; b < a; b++ ) if ( B.J.chunk_list[1 + a] == B.J.chunk_list[1 + b] ) bad = 1;
      pool_push( H, B.J.chunk_list + 1, 0, B.sum.num_chunks );      // released: the chunks go back (k_pool_free_lists)
      done++;
    }
    if ( periods > ( 1ull << 32 ) ) { free( heap_mem ); return 103; }
  }
  if ( pool.avail != static_cast<int32_t>( heap_chunks ) ) bad = 1;            // every chunk came back, once
  if ( stats ) { stats[0] = periods; stats[1] = boundary_passes; stats[2] = handed_back; stats[3] = peak_out; stats[4] = peak_busy; stats[5] = mp_frames; stats[6] = parked; }
  free( heap_mem );
  return bad ? 100 : 0;
}

// tok::mp_deal for every rank of a wave with n_idle idle lanes that drew `got` tickets of parts[t] partitions each:
// out[rank] = { any, ticket, part, n, start }
void wave_sim_deal( uint32_t n_idle, uint32_t got, const uint32_t * parts, uint32_t * out )
{
  for ( uint32_t rank = 0; rank < n_idle; rank++ ) {
    const aa::tok::MpDeal d = aa::tok::mp_deal( n_idle, got, rank, [&]( uint32_t t ) { return parts[t]; } );
    out[5 * rank + 0] = d.any; out[5 * rank + 1] = d.ticket; out[5 * rank + 2] = d.part; out[5 * rank + 3] = d.n; out[5 * rank + 4] = d.start;
  }
}

} // extern "C"
