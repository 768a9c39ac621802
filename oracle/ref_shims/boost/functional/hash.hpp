/* Build glue for oracle/_ref only (test infrastructure; never linked into the product).
 * The reference uses boost::hash_combine / hash_range solely for RasterHandle::hash(),
 * DecoderState::hash() and minihash (raster.cc:29, decoder.cc:35, probability_tables.cc:32).
 * boost is absent from this image; this stand-in keeps those functions callable.  The
 * numeric hash values are NOT those of real boost -- irrelevant for pixel parity. */
#pragma once
#include <cstddef>
#include <functional>
namespace boost {
template <class T> inline void hash_combine( std::size_t & seed, const T & v )
{ seed ^= std::hash<T>()( v ) + 0x9e3779b9 + ( seed << 6 ) + ( seed >> 2 ); }
template <class It> inline void hash_range( std::size_t & seed, It first, It last )
{ for ( ; first != last; ++first ) hash_combine( seed, *first ); }
}
