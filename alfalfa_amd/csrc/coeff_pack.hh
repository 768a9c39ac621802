// Packed coefficient storage (tok_fsm.hh, "Packed coefficients", says what the token lanes write) and how it is read: by the
// reconstruction kernels directly (kernels.hip, round 6: no dense copy is made any more), by aa_stream_read_records (host side,
// frames whose words were copied back) and by the host simulation of the token lanes (tests/cpp/fsm_sim.cc), which compares the
// expanded blocks with the host parser's.
//
// A macroblock's words (round 6 layout) = kMaskSlots mask words, slot b for block b of nz_mask's numbering (0..15 Y in raster
// order, 16..19 U, 20..23 V, 24 Y2), written only for the blocks whose nz_mask bit is set (the others: whatever the memory held)
// -- bit k of a mask: zigzag position k holds a coefficient (tokens.cc:50-135 walks a block in zigzag order) -- followed by the
// coefficients themselves as int16, block after block in PARSE order (Y2 first, then Y, U, V: macroblock.cc:475-502), lowest
// zigzag position first.  Fixed slots instead of "mask word, values, mask word, values" (rounds 3-5): where a block's values start
// is then a prefix sum over 25 masks that 16 lanes form together, not a walk through the words -- which is what lets the
// reconstruction kernels read the words themselves.  Dense, coefficient k of the zigzag scan sits at raster position zigzag[k] of
// the block (the order the IDCT reads).
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

constexpr uint32_t kMaskSlots = 25;
// zigzag position of raster position j
AA_HD inline uint32_t zigzag_of( uint32_t j ) { return static_cast<uint32_t>( kInvZigzagNib >> ( 4 * j ) ) & 15u; }
// the coefficient at raster position j (0..15) of a stored block: mask = its mask word, v = its first value
template <class P> AA_HD inline int16_t value_at( uint32_t mask, P v, uint32_t j )
{
  const uint32_t k = zigzag_of( j );
  if ( !( ( mask >> k ) & 1u ) ) return 0;
  return static_cast<int16_t>( v[popc( mask & ( ( 1u << k ) - 1u ) )] );
}
// block numbers in parse order: 24 (Y2), 0 .. 23
AA_HD inline uint32_t parse_order_block( uint32_t p ) { return p == 0 ? 24u : p - 1u; }

// where a macroblock's words start in the heap (in 16-bit words from the heap's base): pos = packed_pos[mb], list = the
// frame's chunk list
template <class L> AA_HD inline size_t word_offset( uint32_t pos, L list )
{
  return static_cast<size_t>( list[1u + ( pos >> 15 )] ) * kChunkWords + ( pos & ( kChunkWords - 1u ) );
}

// One macroblock: its stored blocks, dense, back to back in parse order -> words the macroblock takes (0 when it stores nothing:
// the lane gives the mask slots of such a macroblock back)
inline uint32_t expand_macroblock( const int16_t * words, uint32_t nz_mask, int16_t * dense )
{
  if ( !( nz_mask & 0x01FFFFFFu ) ) return 0;
  const int16_t * v = words + kMaskSlots;
  uint32_t n = 0;
  for ( uint32_t p = 0; p < 25; p++ ) {
    const uint32_t b = parse_order_block( p );
    if ( !( ( nz_mask >> b ) & 1u ) ) continue;
    const uint32_t mask = static_cast<uint16_t>( words[b] );
    for ( uint32_t j = 0; j < 16; j++ ) dense[16 * n + j] = value_at( mask, v, j );
    v += popc( mask );
    n++;
  }
  return static_cast<uint32_t>( v - words );
}

} // namespace pack
} // namespace aa
