// Test harness (not product code): replays the DEVICE parse algorithm on the host, one lane at a time, so that the state
// machine of alfalfa_amd/csrc/tok_fsm.hh and the 32-bit boolean decoder of parse_common.hh can be checked against the host
// parser without a GPU.  Built by tests/test_fsm_sim.py with plain g++ from the product's own sources:
//   g++ -shared fsm_sim.cc ../../alfalfa_amd/csrc/parser.cpp
// It follows parse_kernels.hip statement for statement: header pre-pass (Parser::parse_header, the real product code),
// k_parse_mb_headers' loop, k_segment_fixup's loop, k_parse_tokens' loop.
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../alfalfa_amd/csrc/bool_reader.hh"
#include "../../alfalfa_amd/csrc/parser.hh"
#include "../../alfalfa_amd/csrc/tok_fsm.hh"
#include "../../alfalfa_amd/csrc/coeff_pack.hh"

namespace {
int16_t g_sink = 0;          // where a lane without a chunk stores (tok::step: the store is unconditional)
struct Sim {
  aa::Parser parser;
  uint32_t pool_chunks = 0;        // coefficient chunks the pool offers per frame (0: plenty)
  bool packed = false;             // the lanes store packed coefficients (tok_fsm.hh), expanded below as k_dense_index / k_expand_coeffs do
  uint32_t last_status = 0, last_chunks = 0, last_words = 0;
  std::vector<uint8_t> segmap;     // the stream's persistent segment map as the device keeps it
  Sim( uint16_t w, uint16_t h ) : parser( w, h ), segmap( size_t( parser.mb_width() ) * parser.mb_height(), 3 ) {}
};
void * aligned( size_t bytes ) { void * p = nullptr; if ( posix_memalign( &p, 256, bytes ? bytes : 256 ) ) return nullptr; std::memset( p, 0xA5, bytes ); return p; }
}

extern "C" {

void * fsm_sim_create( uint16_t w, uint16_t h ) { return new Sim( w, h ); }
void fsm_sim_destroy( void * s ) { delete static_cast<Sim *>( s ); }

// -> 0 ok, else the aa_status of the header pre-pass.  mbs: nmb records; coeffs: 25*nmb+1 blocks; steps: diagnostics
int fsm_sim_frame( void * handle, const uint8_t * data, size_t size, aa_frame_header * hdr, aa_mb_info * mbs, int16_t * coeffs,
                   uint32_t * steps )
{
  Sim & S = *static_cast<Sim *>( handle );
  aa::ParseJob J;
  std::memset( &J, 0, sizeof J );
  try { S.parser.parse_header( data, size, *hdr, J.fp ); }
  catch ( const aa::ParseError & e ) { return e.code; }
  const uint32_t nmb = uint32_t( J.fp.mbw ) * J.fp.mbh;
  // device buffers as the runtime lays them out: 16-byte aligned, padded, garbage-filled
  J.size = uint32_t( size ); J.data_padded = ( J.size + 15 ) & ~15u;
  uint8_t * dev_data = static_cast<uint8_t *>( aligned( J.data_padded ) );
  std::memcpy( dev_data, data, size );
  J.data = dev_data;
  J.nmb = nmb; J.flags_padded = ( nmb + 15 ) & ~15u;
  J.mbflags = static_cast<uint8_t *>( aligned( J.flags_padded ) );
  J.mbs = static_cast<aa_mb_info *>( aligned( nmb * sizeof( aa_mb_info ) ) );
  J.chunk_list = static_cast<uint32_t *>( aligned( size_t( aa::chunk_list_entries( nmb ) ) * 4 ) );
  J.packed_pos = S.packed ? static_cast<uint32_t *>( aligned( size_t( nmb ) * 4 ) ) : nullptr;
  // the coefficient heap as the runtime sets it up: chunks handed out through the pool's ring -- here in an order that is
  // neither ascending nor contiguous, and with `pool_chunks` of them only (0: as many as the worst case needs)
  const uint32_t worst_chunks = aa::chunk_list_entries( nmb ) - 1;
  const uint32_t heap_chunks = worst_chunks + 3;
  const uint32_t avail_chunks = S.pool_chunks ? S.pool_chunks : heap_chunks;
  int16_t * heap_mem = static_cast<int16_t *>( aligned( size_t( heap_chunks ) * aa::kChunkBlocks * 32 ) );
  uint32_t ring_entries = 1; while ( ring_entries < heap_chunks ) ring_entries <<= 1;
  std::vector<uint32_t> ring( ring_entries, 0xDEADBEEFu );
  aa::CoeffPool pool; std::memset( &pool, 0, sizeof pool ); pool.mask = ring_entries - 1;
  aa::Heap H; H.base = heap_mem; H.pool = &pool; H.ring = ring.data();
  uint32_t stride = 1;
  for ( uint32_t c : { 7u, 11u, 13u, 5u, 3u } ) if ( heap_chunks % c ) { stride = c; break; }       // (coprime: a permutation)
  for ( uint32_t k = 0; k < avail_chunks && k < heap_chunks; k++ ) aa::tok::pool_push( H, nullptr, ( k * stride + 2u ) % heap_chunks, 1 );
  const unsigned words_per_row = ( J.fp.mbw + 63 ) / 64;
  J.intra_rows = static_cast<unsigned long long *>( aligned( size_t( words_per_row ) * J.fp.mbh * 8 ) );
  aa::FrameSummary sum; std::memset( &sum, 0, sizeof sum );
  J.summary = &sum;

  // ---- k_parse_mb_headers ----
  {
    aa::BoolReader32 bd;
    aa::BoolState st; st.bitpos = J.fp.bd_bitpos; st.range = J.fp.bd_range; st.active = J.fp.bd_active;
    bd.resume( J.data + J.fp.first_off, J.fp.first_size, st );
    unsigned mi = 0; uint32_t intra = 0, split = 0;
    for ( unsigned row = 0; row < J.fp.mbh; row++ ) {
      unsigned long long word = 0;
      for ( unsigned col = 0; col < J.fp.mbw; col++, mi++ ) {
        const uint8_t flags = aa::parse_mb_header( bd, J.fp, aa::kHeaderTables, J.mbs, mi, col, row, static_cast<uint8_t *>( nullptr ) );
        J.mbflags[mi] = flags;
        if ( !( flags & AA_MB_INTER ) ) { intra++; word |= 1ull << ( col & 63 ); }
        else if ( J.mbs[mi].y_mode == aa::SPLITMV ) split = 1;
        if ( ( col & 63 ) == 63 || col + 1 == J.fp.mbw ) { J.intra_rows[row * words_per_row + ( col >> 6 )] = word; word = 0; }
      }
    }
    sum.num_intra_mbs = intra; sum.has_split = split;
  }
  // ---- k_segment_fixup ----
  if ( J.fp.seg_enabled ) {
    if ( S.parser.segment_map_reset() ) std::memset( S.segmap.data(), 3, S.segmap.size() );
    for ( uint32_t mi = 0; mi < nmb; mi++ ) aa::segment_fixup( J.fp, J.mbs[mi], S.segmap[mi] );
  }
  // ---- k_parse_tokens ----
  {
    const uint32_t bytes = aa::tok::kTablesBytes + aa::tok::lane_lds_bytes( J.fp.mbw, J.fp.nparts > 1 );
    std::vector<uint8_t> store( bytes + 16 );
    uint8_t * smem = reinterpret_cast<uint8_t *>( ( reinterpret_cast<uintptr_t>( store.data() ) + 15 ) & ~uintptr_t( 15 ) );
    std::memset( smem, 0xA5, bytes );
    for ( uint32_t k = 0; k < aa::tok::kTablesBytes / 4; k++ ) reinterpret_cast<uint32_t *>( smem )[k] = aa::tok::table_word( k );
    aa::tok::Lane L;
    std::memset( &L, 0xA5, sizeof L );
    aa::tok::Frame F = aa::tok::frame_of( &J );
    const uint32_t lane_bytes = aa::tok::lane_lds_bytes( J.fp.mbw, J.fp.nparts > 1 );
    aa::tok::init_lane( L, aa::tok::ring_addr( 0 ), aa::tok::slice_addr( 0, 1, lane_bytes ), &g_sink );
    aa::tok::preload( L, smem );
    aa::tok::begin_frame( L, smem, L.base, F );
    for ( ;; ) {
      aa::tok::top_up( L, smem, F );
      if ( L.rec == aa::tok::R_DONE ) break;
      if ( S.packed ) aa::tok::run_period<true>( L, smem, F, H );
      else aa::tok::run_period<false>( L, smem, F, H );
    }
  }
  S.last_status = sum.status; S.last_chunks = sum.num_chunks; S.last_words = sum.packed_words;
  int bad = 0;
  if ( !sum.done || J.chunk_list[0] != sum.num_chunks ) bad = 1;
  if ( getenv( "FSM_SIM_DEBUG" ) ) fprintf( stderr, "done %u list0 %u chunks %u avail %d of %u status %u blocks %u\n", sum.done, J.chunk_list[0], sum.num_chunks, pool.avail, avail_chunks, sum.status, sum.num_coeff_blocks );
  // every chunk number distinct and one the pool handed out; chunks not taken are still in the pool
  for ( uint32_t k = 0; k < sum.num_chunks; k++ ) for ( uint32_t j = 0; j < k; j++ ) if ( J.chunk_list[1 + k] == J.chunk_list[1 + j] ) bad = 1;
  if ( pool.avail + static_cast<int32_t>( sum.num_chunks ) != static_cast<int32_t>( std::min( avail_chunks, heap_chunks ) ) ) bad = 1;
  hdr->num_coeff_blocks = sum.num_coeff_blocks;
  hdr->num_intra_mbs = sum.num_intra_mbs;
  hdr->has_intra_mb = sum.num_intra_mbs != 0;
  if ( steps ) *steps = sum.steps;
  std::memcpy( mbs, J.mbs, nmb * sizeof( aa_mb_info ) );
  // the frame's own view of its coefficients (what aa_stream_read_records gives): blocks back to back in parse order,
  // coeff_index counted from the frame's first block; every macroblock's blocks must lie inside one of the frame's chunks
  if ( sum.status == aa::TOK_OK && S.packed ) {
    // k_dense_index: coeff_index = blocks stored before the macroblock; k_expand_coeffs: every macroblock's words -> dense blocks
    uint32_t running = 0, words = 0;
    for ( uint32_t mi = 0; mi < nmb; mi++ ) {
      const uint32_t nblk = aa::pack::blocks_of( mbs[mi].nz_mask );
      if ( nblk ) {
        const uint32_t pos = J.packed_pos[mi], ord = pos >> 15, off = pos & ( aa::kChunkWords - 1u );
        if ( ord >= sum.num_chunks || running + nblk > sum.num_coeff_blocks ) { bad = 1; break; }
        const int16_t * w = heap_mem + aa::pack::word_offset( pos, J.chunk_list );
        // the record's own 40-bit word offset (what the reconstruction kernels follow) names the same place
        if ( ( static_cast<unsigned long long>( mbs[mi].reserved ) << 32 | mbs[mi].coeff_index ) != static_cast<unsigned long long>( w - heap_mem ) ) bad = 1;
        // as the reconstruction kernels read it: a block's mask from its slot, its values behind the values of the stored blocks
        // before it in parse order (a prefix sum over the masks' populations), one raster position at a time
        const int16_t * wb = w + aa::pack::kMaskSlots;
        for ( uint32_t p = 0, b = 0; p < 25; p++ ) {
          const uint32_t blk = aa::pack::parse_order_block( p );
          if ( !( ( mbs[mi].nz_mask >> blk ) & 1u ) ) continue;
          const uint32_t mask = static_cast<uint16_t>( w[blk] );
          if ( !mask ) bad = 1;                               // a stored block holds a coefficient
          for ( uint32_t j = 0; j < 16; j++ ) coeffs[( size_t( running ) + b ) * 16 + j] = aa::pack::value_at( mask, wb, j );
          wb += aa::pack::popc( mask );
          b++;
        }
        // ... and the host-side form (aa_stream_read_records) must say the same
        std::vector<int16_t> again( size_t( nblk ) * 16 );
        const uint32_t used = aa::pack::expand_macroblock( w, mbs[mi].nz_mask, again.data() );
        if ( used != static_cast<uint32_t>( wb - w ) || std::memcmp( again.data(), coeffs + size_t( running ) * 16, again.size() * 2 ) ) bad = 1;
        if ( off + used > aa::kChunkWords ) bad = 1;          // a macroblock never straddles a chunk
        words += used;
      }
      mbs[mi].coeff_index = running; mbs[mi].reserved = 0;      // the frame's own view, as aa_stream_read_records gives it
      running += nblk;
    }
    if ( running != sum.num_coeff_blocks ) bad = 1;
    // the lane's word count: what the macroblocks took + what was left unused at the end of the chunks it moved on from
    if ( sum.packed_words < words || sum.packed_words > words + sum.num_chunks * aa::kMbWords ) bad = 1;
  } else if ( sum.status == aa::TOK_OK ) {
    uint32_t running = 0;
    for ( uint32_t mi = 0; mi < nmb; mi++ ) {
      const uint32_t nblk = static_cast<uint32_t>( __builtin_popcount( mbs[mi].nz_mask ) );
      if ( nblk ) {
        const uint32_t c = mbs[mi].coeff_index / aa::kChunkBlocks, o = mbs[mi].coeff_index % aa::kChunkBlocks;
        bool mine = false;
        for ( uint32_t k = 0; k < sum.num_chunks; k++ ) if ( J.chunk_list[1 + k] == c ) mine = true;
        if ( !mine || o + nblk > aa::kChunkBlocks || running + nblk > sum.num_coeff_blocks ) { bad = 1; break; }
        std::memcpy( coeffs + size_t( running ) * 16, heap_mem + size_t( mbs[mi].coeff_index ) * 16, size_t( nblk ) * 32 );
      }
      mbs[mi].coeff_index = running;
      running += nblk;
    }
    if ( running != sum.num_coeff_blocks ) bad = 1;
    if ( sum.packed_words ) bad = 1;
  }
  // the intra row masks must say what the records say
  for ( unsigned row = 0; row < J.fp.mbh; row++ ) for ( unsigned col = 0; col < J.fp.mbw; col++ ) {
    const bool bit = ( J.intra_rows[row * words_per_row + ( col >> 6 )] >> ( col & 63 ) ) & 1;
    if ( bit != !( J.mbs[row * J.fp.mbw + col].flags & AA_MB_INTER ) ) bad = 1;
  }
  free( dev_data ); free( J.mbflags ); free( J.mbs ); free( J.chunk_list ); free( J.packed_pos ); free( heap_mem ); free( J.intra_rows );
  if ( sum.status != aa::TOK_OK ) return 200 + static_cast<int>( sum.status );
  return bad ? 100 : 0;
}

// Hand-over of a boolean decoder between window widths: after every one of the first `n` bools of the host reader (64-bit
// window) its exported state must let the 32-bit reader continue with the same bits.  probs cycle through a fixed pattern.
// -> number of hand-over points at which the two readers disagreed within the next `look` bools
int fsm_sim_handover_check( const uint8_t * data, size_t size, int n, int look )
{
  static const uint8_t probs[8] = { 128, 1, 255, 200, 37, 128, 250, 90 };
  int bad = 0;
  aa::BoolReader host( data, size );
  for ( int k = 0; k < n; k++ ) {
    aa::BoolReader h2 = host;
    aa::BoolReader32 dev;
    dev.resume( data, static_cast<uint32_t>( size ), host.state() );
    for ( int j = 0; j < look; j++ ) if ( h2.get( probs[( k + j ) & 7] ) != dev.get( probs[( k + j ) & 7] ) ) { bad++; break; }
    host.get( probs[k & 7] );
  }
  return bad;
}

// A host core in the role of a token lane (runtime.cpp, host lanes): Parser::parse_header + aa::parse_frame_body must produce the
// records Parser::parse produces.  The frames of a stream one after the other (the handle keeps a second parser: header state only).
// -> 0 equal, 1 records differ, 2 coefficient blocks differ, 3 counts differ, 4 not eligible (segmentation: not checked), 5 frame rejected by both parsers, 6 by one
struct BodySim { aa::Parser full, hdr_only; BodySim( uint16_t w, uint16_t h ) : full( w, h ), hdr_only( w, h ) {} };
void * fsm_sim_body_create( uint16_t w, uint16_t h ) { return new BodySim( w, h ); }
void fsm_sim_body_destroy( void * p ) { delete static_cast<BodySim *>( p ); }
int fsm_sim_body_frame( void * handle, const uint8_t * data, size_t size )
{
  BodySim & S = *static_cast<BodySim *>( handle );
  const size_t nmb = size_t( S.full.mb_width() ) * S.full.mb_height();
  std::vector<aa_mb_info> m1( nmb ), m2( nmb );
  std::vector<int16_t> c1( nmb * 25 * 16 + 16 ), c2( nmb * 25 * 16 + 16 );
  std::vector<uint8_t> above( size_t( S.full.mb_width() ) * 9 );
  std::memset( static_cast<void *>( m1.data() ), 0, nmb * sizeof( aa_mb_info ) ); std::memset( static_cast<void *>( m2.data() ), 0, nmb * sizeof( aa_mb_info ) );
  aa_frame_header h1, h2; aa::FrameParams fp;
  int e1 = 0, e2 = 0;
  try { S.full.parse( data, size, h1, m1.data(), c1.data() ); } catch ( const aa::ParseError & e ) { e1 = e.code; }
  try { S.hdr_only.parse_header( data, size, h2, fp ); } catch ( const aa::ParseError & e ) { e2 = e.code; }
  if ( e1 || e2 ) return e1 == e2 ? 5 : 6;        // rejected by both (the same way) / by one only
  if ( fp.seg_enabled ) return 4;
  uint32_t blocks = 0, intra = 0;
  aa::parse_frame_body( data, fp, m2.data(), c2.data(), above.data(), &blocks, &intra );
  if ( blocks != h1.num_coeff_blocks || intra != h1.num_intra_mbs ) return 3;
  if ( std::memcmp( m1.data(), m2.data(), nmb * sizeof( aa_mb_info ) ) ) return 1;
  if ( std::memcmp( c1.data(), c2.data(), size_t( blocks ) * 32 ) ) return 2;
  return 0;
}

// The same host-lane parse with the frame's bytes ending EXACTLY at an inaccessible page (the pinned arena's end, runtime.cpp):
// a decoder that has run out of partition reads zeros (bool_decoder.hh:56-65) and must not touch memory for them.  Runs in a
// forked child so that a stray read is a result, not a dead test process.
// -> 0 parsed, 4 segmentation (not eligible), 5 rejected by the header pre-pass, 100 the child died (a read past the buffer), 101 could not set the check up
int fsm_sim_body_guarded( uint16_t w, uint16_t h, const uint8_t * data, size_t size, int conceal )
{
  aa::Parser hdr_only( w, h );
  hdr_only.set_error_concealment( conceal != 0 );
  aa_frame_header hd; aa::FrameParams fp;
  try { hdr_only.parse_header( data, size, hd, fp ); } catch ( const aa::ParseError & ) { return 5; }
  if ( fp.seg_enabled ) return 4;
  const size_t page = static_cast<size_t>( sysconf( _SC_PAGESIZE ) ), span = ( ( size + page - 1 ) / page + 1 ) * page;
  uint8_t * region = static_cast<uint8_t *>( mmap( nullptr, span + page, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0 ) );
  if ( region == MAP_FAILED ) return 101;
  if ( mprotect( region + span, page, PROT_NONE ) ) { munmap( region, span + page ); return 101; }
  uint8_t * copy = region + span - size;                    // the frame's last byte is the last byte in front of the guard page
  std::memcpy( copy, data, size );
  int result = 101;
  const pid_t pid = fork();
  if ( pid == 0 ) {
    const size_t nmb = size_t( hdr_only.mb_width() ) * hdr_only.mb_height();
    std::vector<aa_mb_info> m( nmb );
    std::vector<int16_t> c( nmb * 25 * 16 + 16 );
    std::vector<uint8_t> above( size_t( hdr_only.mb_width() ) * 9 );
    uint32_t blocks = 0, intra = 0;
    aa::parse_frame_body( copy, fp, m.data(), c.data(), above.data(), &blocks, &intra );
    _exit( 0 );
  } else if ( pid > 0 ) {
    int st = 0;
    if ( waitpid( pid, &st, 0 ) == pid ) result = ( WIFEXITED( st ) && WEXITSTATUS( st ) == 0 ) ? 0 : 100;
  }
  munmap( region, span + page );
  return result;
}

// the pool offers only `chunks` coefficient chunks to the next frames (0: plenty) -> a frame that needs more is handed back
// with TOK_NO_MEMORY (fsm_sim_frame returns 202)
void fsm_sim_set_pool_chunks( void * handle, uint32_t chunks ) { static_cast<Sim *>( handle )->pool_chunks = chunks; }
uint32_t fsm_sim_last_chunks( void * handle ) { return static_cast<Sim *>( handle )->last_chunks; }
// the lanes store packed coefficients from the next frame on (expanded before they are returned)
void fsm_sim_set_packed( void * handle, int on ) { static_cast<Sim *>( handle )->packed = on != 0; }
uint32_t fsm_sim_last_words( void * handle ) { return static_cast<Sim *>( handle )->last_words; }

// persistent state for comparison with the host parser's
void fsm_sim_segmap( void * handle, uint8_t * out ) { Sim & S = *static_cast<Sim *>( handle ); std::memcpy( out, S.segmap.data(), S.segmap.size() ); }
void fsm_sim_probs( void * handle, uint8_t * out )
{
  const aa::ProbTables & t = static_cast<Sim *>( handle )->parser.probs();
  std::memcpy( out, t.coeff, 1056 ); std::memcpy( out + 1056, t.y_mode, 4 ); std::memcpy( out + 1060, t.uv_mode, 3 ); std::memcpy( out + 1063, t.mv, 38 );
}

} // extern "C"
