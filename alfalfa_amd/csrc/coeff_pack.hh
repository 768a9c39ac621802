// Packed coefficient storage -> the dense 16-coefficient blocks reconstruction reads (tok_fsm.hh, "Packed coefficients", says
// what the token lanes write).  The same statements run in k_dense_index / k_expand_coeffs (parse_kernels.hip), in
// aa_stream_read_records (host side, frames whose words were copied back) and in the host simulation of the token lanes
// (tests/cpp/fsm_sim.cc), which compares the expanded blocks with the host parser's.
//
// A stored block = one mask word (bit k: zigzag position k holds a coefficient; tokens.cc:50-135 walks a block in zigzag
// order) + the coefficients of the set bits as int16, lowest zigzag position first.  Dense, coefficient k of the zigzag scan
// sits at raster position zigzag[k] of the block (the order the IDCT reads).
#pragma once
#include "tok_fsm.hh"

namespace aa {
namespace pack {

// raster position -> zigzag position (the inverse of tok::kZigzagNib)
constexpr uint64_t kInvZigzagNib = tok::nib( { 0, 1, 5, 6, 2, 4, 7, 12, 3, 8, 11, 13, 9, 10, 14, 15 } );
constexpr bool inverse_ok()
{
  for ( unsigned k = 0; k < 16; k++ )
    if ( ( ( kInvZigzagNib >> ( 4 * ( ( tok::kZigzagNib >> ( 4 * k ) ) & 15u ) ) ) & 15u ) != k ) return false;
  return true;
}
static_assert( inverse_ok(), "kInvZigzagNib is not the inverse of the zigzag scan" );

AA_HD inline uint32_t popc( uint32_t v )
{
#if defined( __HIP_DEVICE_COMPILE__ )
  return static_cast<uint32_t>( __popc( v ) );
#else
  return static_cast<uint32_t>( __builtin_popcount( v ) );
#endif
}

// stored blocks of a macroblock
AA_HD inline uint32_t blocks_of( uint32_t nz_mask ) { return popc( nz_mask & 0x01FFFFFFu ); }

// the coefficient at raster position j (0..15) of the stored block whose words start at w
template <class P> AA_HD inline int16_t value_at( P w, uint32_t j )
{
  const uint32_t mask = static_cast<uint16_t>( w[0] );
  const uint32_t k = static_cast<uint32_t>( kInvZigzagNib >> ( 4 * j ) ) & 15u;
  if ( !( ( mask >> k ) & 1u ) ) return 0;
  return static_cast<int16_t>( w[1 + popc( mask & ( ( 1u << k ) - 1u ) )] );
}
// words the block takes
template <class P> AA_HD inline uint32_t block_words( P w ) { return 1u + popc( static_cast<uint16_t>( w[0] ) ); }

// where a macroblock's words start in the heap (in 16-bit words from the heap's base): pos = packed_pos[mb], list = the
// frame's chunk list
template <class L> AA_HD inline size_t word_offset( uint32_t pos, L list )
{
  return static_cast<size_t>( list[1u + ( pos >> 15 )] ) * kChunkWords + ( pos & ( kChunkWords - 1u ) );
}

// One macroblock, all 16 raster positions (host side; the kernel gives a position to each of 16 lanes): -> words consumed
inline uint32_t expand_macroblock( const int16_t * words, uint32_t nz_mask, int16_t * dense )
{
  const uint32_t n = blocks_of( nz_mask );
  const int16_t * w = words;
  for ( uint32_t b = 0; b < n; b++ ) {
    for ( uint32_t j = 0; j < 16; j++ ) dense[16 * b + j] = value_at( w, j );
    w += block_words( w );
  }
  return static_cast<uint32_t>( w - words );
}

} // namespace pack
} // namespace aa
