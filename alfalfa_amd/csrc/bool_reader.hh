// VP8 boolean entropy decoder, host side.
//
// Same arithmetic as the reference's BoolDecoder (bool_decoder.hh:45-120: split = 1 + (((range-1)*prob)>>8),
// renormalise while range < 128, bytes past the end read as zero), but kept in a 64-bit MSB-aligned window
// that is refilled up to 7 bytes at a time and renormalised with one count-leading-zeros instead of a
// bit-at-a-time loop.  The sequence of returned bits is identical (RFC 6386 section 7 decoder).
#pragma once
#include <cstddef>
#include <cstdint>

#include "parse_common.hh"

namespace aa {

class BoolReader
{
  const uint8_t * begin_ = nullptr;
  const uint8_t * p_ = nullptr;
  const uint8_t * end_ = nullptr;
  uint64_t value_ = 0;   // window, most significant bits first
  int count_ = -8;       // valid bits in the window beyond the 8 being compared
  uint32_t range_ = 255;

  static constexpr int kWindow = 64;
  static constexpr int kLotsOfBits = 0x40000000;

  void fill()
  {
    int shift = kWindow - 8 - ( count_ + 8 );
    while ( shift >= 0 ) {
      if ( p_ == end_ ) { count_ += kLotsOfBits; break; }   // exhausted: zeros stream in forever
      count_ += 8;
      value_ |= static_cast<uint64_t>( *p_++ ) << shift;
      shift -= 8;
    }
  }

public:
  BoolReader() = default;
  BoolReader( const uint8_t * data, size_t size ) { reset( data, size ); }

  void reset( const uint8_t * data, size_t size )
  {
    begin_ = p_ = data; end_ = data + size; value_ = 0; count_ = -8; range_ = 255;
    fill();
  }

  // Window-independent form of the current state (parse_common.hh): a reader of any window width, e.g. a GPU lane,
  // continues from it and returns the same bits.
  BoolState state() const
  {
    BoolReader t = *this;
    if ( t.count_ < 0 ) t.fill();     // this reader refills lazily: make the 8 bits being compared whole first
    const int real = t.count_ >= kLotsOfBits / 2 ? t.count_ - kLotsOfBits : t.count_;   // valid bits below the active byte
    BoolState st;
    st.bitpos = static_cast<uint32_t>( 8 * ( t.p_ - t.begin_ ) - real );   // (past the end: zeros that were shifted in)
    st.range = static_cast<uint8_t>( t.range_ );
    st.active = static_cast<uint8_t>( t.value_ >> ( kWindow - 8 ) );
    return st;
  }

  inline int get( const uint32_t prob )
  {
    const uint32_t split = 1 + ( ( ( range_ - 1 ) * prob ) >> 8 );
    if ( count_ < 0 ) fill();
    const uint64_t bigsplit = static_cast<uint64_t>( split ) << ( kWindow - 8 );
    int bit;
    uint32_t range;
    if ( value_ >= bigsplit ) { range = range_ - split; value_ -= bigsplit; bit = 1; }
    else { range = split; bit = 0; }
    const int shift = __builtin_clz( range ) - 24;   // range in [1,255] -> shift so that range >= 128
    range_ = range << shift;
    value_ <<= shift;
    count_ -= shift;
    return bit;
  }

  inline int flag() { return get( 128 ); }

  inline int literal( int bits )   // Unsigned<width>: MSB first (vp8_header_structures.hh:52-66)
  {
    int v = 0;
    while ( bits-- ) v = ( v << 1 ) | get( 128 );
    return v;
  }

  inline int signed_literal( int bits )   // Signed<width>: magnitude, then sign (vp8_header_structures.hh:71-84)
  {
    const int v = literal( bits );
    return get( 128 ) ? -v : v;
  }

  // Tree<>: tree.cc:35-57.  Leaves are stored as -value, inner nodes as positive even indices.
  inline int tree( const int8_t * nodes, const uint8_t * probs )
  {
    int i = 0;
    while ( ( i = nodes[ i + get( probs[ i >> 1 ] ) ] ) > 0 ) {}
    return -i;
  }
};

} // namespace aa
