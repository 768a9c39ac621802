// alfalfa.hh -- C++ mirror of the reference's decoder surface for THIS path, over the C ABI (alfalfa_amd.h).
//
// Same class and method names as excamera/alfalfa so that callers compile against either implementation:
//   exceptions  Invalid / Unsupported / LogicError                     (util/exception.hh:76-98)
//   Optional<T> initialized() / get() / get_or()                       (util/optional.hh)
//   Chunk       non-owning {buffer,size} view, bounds-checked          (util/chunk.hh:38-143)
//   IVF         container index: width/height/frame_count/frame(i)     (util/ivf.hh, ivf.cc:36-82)
//   VP8Raster   Y()/U()/V().at(col,row), width/height (padded), display_width/height, dump(), display_rectangle_as_planar()
//                                                                      (util/raster.hh:48-90, decoder/vp8_raster.hh:53-316)
//   RasterHandle  get() / operator const VP8Raster & -- lazy, cached host copy of a device-resident raster
//                                                                      (decoder/raster_handle.hh:77-123)
//   References  last / golden / alternative                            (decoder/decoder.hh:123-149)
//   Decoder     Decoder(width,height), get_frame_output, parse_and_decode_frame, get_references, example_raster,
//               get_width/get_height                                   (decoder/decoder.hh:244-300)
//   FramePlayer / FilePlayer (= Player)  decode / advance / eof / cur_frame_no   (decoder/player.hh:40-97)
//   FileDescriptor (write only), YUV4MPEGHeader, YUV4MPEGFrameWriter  -- what vp8decode / xc-decode-bundle write with
//                                                                      (util/file_descriptor.hh, input/yuv4mpeg.{hh,cc})
//   Decoder::get_frame_outputs  -- NOT in the reference: one frame of each of N decoders as one GPU batch step
//   DecoderState / ProbabilityTables / Segmentation / FilterAdjustments  public fields, ==, hash(), serialize / deserialize
//               (decoder/decoder.hh:57-225);  Decoder( DecoderState, References ), get_state(), ==, get_hash(), minihash(),
//               minihash_match()  (decoder.hh:244-300; hashes = boost::hash_combine, pre-1.81 formula)
// What is NOT mirrored (host-side plumbing outside the hot path): Frame<> object graphs
// (decompress_frame / parse_frame<F> / decode_frame<F> are replaced by get_frame_output), MutableRasterHandle.
//   EncoderStateSerializer / EncoderStateDeserializer, Decoder / FramePlayer / FilePlayer ::serialize, ::deserialize
//               -- the reference's `.state` files, byte-compatible    (decoder/enc_state_serializer.hh, decoder.cc:48-81)
//
// Everything lives in namespace alfalfa_amd; define ALFALFA_AMD_GLOBAL_NAMES before including to also export the
// names into the global namespace (drop-in for code written against the reference headers).
#pragma once

#include <cassert>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <new>
#include <ostream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include <unistd.h>

extern "C" {
#include "../alfalfa_amd.h"
}

namespace alfalfa_amd {

// ---------------------------------------------------------------- exceptions (what() strings as in the reference)
class internal_error : public std::runtime_error
{
public:
  internal_error( const std::string & attempt, const std::string & error ) : std::runtime_error( attempt + ": " + error ) {}
  explicit internal_error( const std::string & whole ) : std::runtime_error( whole ) {}
};
class Invalid : public internal_error
{
public:
  explicit Invalid( const std::string & e ) : internal_error( "invalid bitstream", e ) {}
  Invalid( const std::string & whole, int ) : internal_error( whole ) {}
};
class Unsupported : public internal_error
{
public:
  explicit Unsupported( const std::string & e ) : internal_error( "unsupported bitstream", e ) {}
  Unsupported( const std::string & whole, int ) : internal_error( whole ) {}
};
class LogicError : public internal_error
{
public:
  LogicError() : internal_error( "internal error", "logic error" ) {}
};
class DeviceError : public std::runtime_error     // HIP failure or no GPU: the path has no CPU fallback
{
public:
  explicit DeviceError( const std::string & e ) : std::runtime_error( e ) {}
};

inline void check( const aa_status s )
{
  if ( s == AA_OK ) return;
  const std::string msg = aa_last_error();     // already carries the reference's "invalid bitstream: ..." prefix
  switch ( s ) {
  case AA_ERR_INVALID: throw Invalid( msg, 0 );
  case AA_ERR_UNSUPPORTED: throw Unsupported( msg, 0 );
  case AA_ERR_OUT_OF_RANGE: throw std::out_of_range( msg );
  case AA_ERR_LOGIC: throw LogicError();
  case AA_ERR_ARGUMENT: throw std::invalid_argument( msg );
  default: throw DeviceError( msg );
  }
}

// ---------------------------------------------------------------- Optional (util/optional.hh: same members; storage in place)
template <class T>
class Optional
{
  bool initialized_ = false;
  typename std::aligned_storage<sizeof( T ), alignof( T )>::type storage_ {};      // (zeroed: gcc -Wmaybe-uninitialized sees through the move of an empty Optional)
  T * ptr() { return reinterpret_cast<T *>( &storage_ ); }
  const T * ptr() const { return reinterpret_cast<const T *>( &storage_ ); }
  void destroy() { if ( initialized_ ) { ptr()->~T(); initialized_ = false; } }
public:
  Optional() {}
  Optional( T && other ) : initialized_( true ) { new ( &storage_ ) T( std::move( other ) ); }
  Optional( const T & other ) : initialized_( true ) { new ( &storage_ ) T( other ); }
  template <typename... Targs>
  Optional( const bool is_present, Targs &&... args ) : initialized_( is_present ) { if ( initialized_ ) new ( &storage_ ) T( std::forward<Targs>( args )... ); }
  Optional( Optional<T> && other ) : initialized_( other.initialized_ ) { if ( initialized_ ) new ( &storage_ ) T( std::move( *other.ptr() ) ); }
  Optional( const Optional<T> & other ) : initialized_( other.initialized_ ) { if ( initialized_ ) new ( &storage_ ) T( *other.ptr() ); }
  ~Optional() { destroy(); }
  template <typename... Targs>
  void initialize( Targs &&... args ) { destroy(); new ( &storage_ ) T( std::forward<Targs>( args )... ); initialized_ = true; }
  const Optional & operator=( Optional<T> && other ) { if ( this != &other ) { destroy(); if ( other.initialized_ ) { new ( &storage_ ) T( std::move( *other.ptr() ) ); initialized_ = true; } } return *this; }
  const Optional & operator=( const Optional<T> & other ) { if ( this != &other ) { destroy(); if ( other.initialized_ ) { new ( &storage_ ) T( *other.ptr() ); initialized_ = true; } } return *this; }
  bool operator==( const Optional<T> & other ) const { return initialized_ ? ( other.initialized_ && get() == other.get() ) : !other.initialized_; }
  bool operator!=( const Optional<T> & other ) const { return !operator==( other ); }
  bool initialized() const { return initialized_; }
  const T & get() const { if ( !initialized_ ) throw std::runtime_error( "attempt to get uninitialized Optional" ); return *ptr(); }
  T & get() { if ( !initialized_ ) throw std::runtime_error( "attempt to get uninitialized Optional" ); return *ptr(); }
  const T & get_or( const T & fallback ) const { return initialized_ ? *ptr() : fallback; }
  void clear() { destroy(); }
};
template <class T> Optional<T> make_optional( const bool init, const T & v ) { return Optional<T>( init, v ); }

// ---------------------------------------------------------------- Chunk
class Chunk
{
  const uint8_t * buffer_;
  uint64_t size_;
  void bounds_check( const uint64_t length ) const { if ( length > size_ ) throw std::out_of_range( "attempted to read past end of chunk" ); }
public:
  Chunk( const uint8_t * buffer, const uint64_t size ) : buffer_( buffer ), size_( size ) {}
  explicit Chunk( const std::string & s ) : buffer_( reinterpret_cast<const uint8_t *>( s.data() ) ), size_( s.size() ) {}
  explicit Chunk( const std::vector<uint8_t> & v ) : buffer_( v.data() ), size_( v.size() ) {}
  const uint8_t * buffer() const { return buffer_; }
  const uint64_t & size() const { return size_; }
  Chunk operator()( const uint64_t offset ) const { return operator()( offset, size_ - offset ); }
  Chunk operator()( const uint64_t offset, const uint64_t length ) const { bounds_check( offset ); bounds_check( offset + length ); return Chunk( buffer_ + offset, length ); }
  std::string to_string() const { return std::string( reinterpret_cast<const char *>( buffer_ ), size_ ); }
  const uint8_t & octet() const { bounds_check( 1 ); return *buffer_; }
  uint16_t le16() const { bounds_check( 2 ); return static_cast<uint16_t>( buffer_[0] | ( buffer_[1] << 8 ) ); }
  uint64_t le32() const { bounds_check( 4 ); return uint64_t( buffer_[0] ) | ( uint64_t( buffer_[1] ) << 8 ) | ( uint64_t( buffer_[2] ) << 16 ) | ( uint64_t( buffer_[3] ) << 24 ); }
  uint64_t bits( const uint64_t bit_offset, const uint64_t bit_length ) const
  {
    const uint64_t byte_len = 1 + ( bit_offset + bit_length - 1 ) / 8;
    bounds_check( byte_len );
    if ( byte_len > 8 || bit_length > 63 ) throw std::out_of_range( "bit offset and length not supported" );
    uint64_t val = 0;
    for ( uint64_t i = 0; i < byte_len; i++ ) val |= uint64_t( buffer_[i] ) << ( i * 8 );
    return ( val >> bit_offset ) & ( ( uint64_t( 1 ) << bit_length ) - 1 );
  }
};

// ---------------------------------------------------------------- IVF
class IVF
{
  std::vector<uint8_t> data_;     // the reference mmaps; a host read is equivalent for callers
  std::string fourcc_;
  uint16_t width_ = 0, height_ = 0;
  uint32_t frame_rate_ = 0, time_scale_ = 0, frame_count_ = 0, expected_decoder_minihash_ = 0;
  std::vector<std::pair<uint64_t, uint32_t>> frame_index_;
public:
  static constexpr int supported_header_len = 32;
  static constexpr int frame_header_len = 12;
  explicit IVF( const std::string & filename )
  {
    std::ifstream in( filename, std::ios::binary );
    if ( !in ) throw std::runtime_error( "open (" + filename + "): cannot open file" );
    data_.assign( std::istreambuf_iterator<char>( in ), std::istreambuf_iterator<char>() );
    try {
      const Chunk file( data_.data(), data_.size() );
      const Chunk header = file( 0, supported_header_len );
      fourcc_ = header( 8, 4 ).to_string();
      width_ = header( 12, 2 ).le16(); height_ = header( 14, 2 ).le16();
      frame_rate_ = static_cast<uint32_t>( header( 16, 4 ).le32() ); time_scale_ = static_cast<uint32_t>( header( 20, 4 ).le32() );
      frame_count_ = static_cast<uint32_t>( header( 24, 4 ).le32() ); expected_decoder_minihash_ = static_cast<uint32_t>( header( 28, 4 ).le32() );
      if ( header( 0, 4 ).to_string() != "DKIF" ) throw Invalid( "missing IVF file header" );
      if ( header( 4, 2 ).le16() != 0 ) throw Unsupported( "not an IVF version 0 file" );
      if ( header( 6, 2 ).le16() != supported_header_len ) throw Unsupported( "unsupported IVF header length" );
      frame_index_.reserve( frame_count_ );
      uint64_t position = supported_header_len;
      for ( uint32_t i = 0; i < frame_count_; i++ ) {
        const uint32_t frame_len = static_cast<uint32_t>( file( position, frame_header_len ).le32() );
        (void) file( position + frame_header_len, frame_len );
        frame_index_.emplace_back( position + frame_header_len, frame_len );
        position += frame_header_len + frame_len;
      }
    } catch ( const std::out_of_range & ) {
      throw Invalid( "IVF file truncated" );
    }
  }
  const std::string & fourcc() const { return fourcc_; }
  uint16_t width() const { return width_; }
  uint16_t height() const { return height_; }
  uint32_t frame_rate() const { return frame_rate_; }
  uint32_t time_scale() const { return time_scale_; }
  uint32_t frame_count() const { return frame_count_; }
  Chunk frame( const uint32_t & index ) const { const auto & e = frame_index_.at( index ); return Chunk( data_.data() + e.first, e.second ); }
  size_t size() const { return data_.size(); }
  uint32_t expected_decoder_minihash() const { return expected_decoder_minihash_; }
};

// ---------------------------------------------------------------- rasters
class Plane     // TwoD<uint8_t> as seen by raster users: at(col,row), width(), height()
{
  unsigned width_, height_;
  std::vector<uint8_t> storage_;
public:
  Plane( const unsigned w, const unsigned h ) : width_( w ), height_( h ), storage_( size_t( w ) * h ) {}
  uint8_t & at( const unsigned column, const unsigned row ) { return storage_[size_t( row ) * width_ + column]; }
  const uint8_t & at( const unsigned column, const unsigned row ) const { return storage_[size_t( row ) * width_ + column]; }
  unsigned width() const { return width_; }
  unsigned height() const { return height_; }
  std::vector<uint8_t>::const_iterator begin() const { return storage_.begin(); }
  std::vector<uint8_t>::const_iterator end() const { return storage_.end(); }
  bool operator==( const Plane & o ) const { return width_ == o.width_ && height_ == o.height_ && storage_ == o.storage_; }
};

// util/ssim.hh:31 -- the reference calls into libx264 (pixel_ssim_wxh over the whole plane, stride = width, / count)
inline double ssim( const Plane & image, const Plane & other_image )
{
  if ( image.width() != other_image.width() || image.height() != other_image.height() ) throw std::invalid_argument( "ssim: planes of different size" );
  double q = 0;
  check( aa_ssim_host( &image.at( 0, 0 ), &other_image.at( 0, 0 ), static_cast<int>( image.width() ), static_cast<int>( image.height() ), &q ) );
  return q;
}

class VP8Raster
{
  uint16_t display_width_, display_height_, width_, height_;
  Plane Y_, U_, V_;
public:
  static unsigned macroblock_dimension( const unsigned num ) { return ( num + 15 ) / 16; }
  VP8Raster( const unsigned display_width, const unsigned display_height )
    : display_width_( display_width ), display_height_( display_height ),
      width_( 16 * macroblock_dimension( display_width ) ), height_( 16 * macroblock_dimension( display_height ) ),
      Y_( width_, height_ ), U_( width_ / 2, height_ / 2 ), V_( width_ / 2, height_ / 2 ) {}
  Plane & Y() { return Y_; }
  Plane & U() { return U_; }
  Plane & V() { return V_; }
  const Plane & Y() const { return Y_; }
  const Plane & U() const { return U_; }
  const Plane & V() const { return V_; }
  uint16_t width() const { return width_; }
  uint16_t height() const { return height_; }
  uint16_t display_width() const { return display_width_; }
  uint16_t display_height() const { return display_height_; }
  uint16_t chroma_display_width() const { return ( 1 + display_width_ ) / 2; }
  uint16_t chroma_display_height() const { return ( 1 + display_height_ ) / 2; }
  bool operator==( const VP8Raster & o ) const { return Y_ == o.Y_ && U_ == o.U_ && V_ == o.V_; }
  bool operator!=( const VP8Raster & o ) const { return !operator==( o ); }
  double quality( const VP8Raster & other ) const { return ssim( Y(), other.Y() ); }        // raster.cc:63-66: SSIM of the luma planes
  void copy_from( const VP8Raster & other )                                                  // raster.cc:78-83
  {
    if ( width_ != other.width_ || height_ != other.height_ ) throw std::invalid_argument( "copy_from: rasters of different size" );
    Y_ = other.Y_; U_ = other.U_; V_ = other.V_;
  }
  std::vector<Chunk> display_rectangle_as_planar() const     // raster.cc:85-104
  {
    std::vector<Chunk> ret;
    for ( uint16_t row = 0; row < display_height(); row++ ) ret.emplace_back( &Y().at( 0, row ), display_width() );
    for ( uint16_t row = 0; row < chroma_display_height(); row++ ) ret.emplace_back( &U().at( 0, row ), chroma_display_width() );
    for ( uint16_t row = 0; row < chroma_display_height(); row++ ) ret.emplace_back( &V().at( 0, row ), chroma_display_width() );
    return ret;
  }
  void dump( FILE * file ) const                              // raster.cc:107-114
  {
    for ( const auto & chunk : display_rectangle_as_planar() )
      if ( 1 != fwrite( chunk.buffer(), chunk.size(), 1, file ) ) throw std::runtime_error( "fwrite returned short write" );
  }
};

// One HIP device context per process by default (device = $ALFALFA_AMD_DEVICE or 0), so that the reference's
// `Decoder( width, height )` signature keeps working unchanged.
class GpuContext
{
  aa_ctx * ctx_ = nullptr;
public:
  explicit GpuContext( const int device ) { aa_runtime_prepare(); check( aa_ctx_create( device, &ctx_ ) ); }
  ~GpuContext() { aa_ctx_destroy( ctx_ ); }
  GpuContext( const GpuContext & ) = delete;
  GpuContext & operator=( const GpuContext & ) = delete;
  aa_ctx * get() const { return ctx_; }
  void sync() const { check( aa_ctx_sync( ctx_ ) ); }
  static const std::shared_ptr<GpuContext> & process_default()
  {
    static const std::shared_ptr<GpuContext> ctx = [] {
      const char * e = std::getenv( "ALFALFA_AMD_DEVICE" );
      return std::make_shared<GpuContext>( e ? std::atoi( e ) : 0 );
    }();
    return ctx;
  }
};

namespace detail {
struct StreamOwner     // shared by a Decoder and every RasterHandle it handed out (rasters outlive the Decoder object)
{
  std::shared_ptr<GpuContext> ctx;
  aa_stream * stream = nullptr;
  uint16_t width, height;
  StreamOwner( std::shared_ptr<GpuContext> c, const uint16_t w, const uint16_t h ) : ctx( std::move( c ) ), width( w ), height( h )
  { check( aa_stream_create( ctx->get(), w, h, &stream ) ); }
  ~StreamOwner() { aa_stream_destroy( stream ); }
  StreamOwner( const StreamOwner & ) = delete;
  StreamOwner & operator=( const StreamOwner & ) = delete;
};
struct RasterState
{
  std::shared_ptr<StreamOwner> owner;
  int frame_index;                         // >= 0: the raster decoded frame `frame_index` produced; -1: a snapshot (blank / imported reference)
  std::unique_ptr<VP8Raster> host;         // filled on first get() (snapshots: at creation)
  ~RasterState() { if ( owner && frame_index >= 0 ) (void) aa_stream_release_frame( owner->stream, frame_index ); }   // raster_handle.cc:113-122
};
inline void hash_combine( size_t & seed, const size_t v ) { seed ^= v + 0x9e3779b9 + ( seed << 6 ) + ( seed >> 2 ); }   // boost < 1.81
}

// VP8MutableRasterHandle (raster_handle.hh:77-100): a raster the caller may still write, for a display size; moving it into a
// RasterHandle freezes it (raster_handle.cc:166-170).  Lives in host memory -- a decoder that is given it as a reference uploads it.
class MutableRasterHandle
{
  std::unique_ptr<VP8Raster> raster_;
  friend class RasterHandle;
public:
  MutableRasterHandle( const unsigned int display_width, const unsigned int display_height )
    : raster_( new VP8Raster( static_cast<uint16_t>( display_width ), static_cast<uint16_t>( display_height ) ) ) {}
  operator const VP8Raster & () const { return *raster_; }
  operator VP8Raster & () { return *raster_; }
  const VP8Raster & get() const { return *raster_; }
  VP8Raster & get() { return *raster_; }
};

class RasterHandle
{
  std::shared_ptr<detail::RasterState> state_;
  friend class Decoder;
public:
  RasterHandle() = default;
  RasterHandle( MutableRasterHandle && mutable_raster )              // raster_handle.cc:166-170
    : state_( std::make_shared<detail::RasterState>() ) { state_->frame_index = -1; state_->host = std::move( mutable_raster.raster_ ); }
  RasterHandle( std::shared_ptr<detail::StreamOwner> owner, const int frame_index )
    : state_( std::make_shared<detail::RasterState>() ) { state_->owner = std::move( owner ); state_->frame_index = frame_index; }
  // a snapshot of References::last / golden / alternative (which = 0, 1, 2) as they stand: blank or imported rasters that no
  // decoded frame of this decoder produced
  static RasterHandle snapshot( std::shared_ptr<detail::StreamOwner> owner, const int which )
  {
    RasterHandle h( owner, -1 );
    auto r = std::unique_ptr<VP8Raster>( new VP8Raster( owner->width, owner->height ) );
    check( aa_stream_reference_download( owner->stream, which, &r->Y().at( 0, 0 ), &r->U().at( 0, 0 ), &r->V().at( 0, 0 ) ) );
    h.state_->host = std::move( r );
    return h;
  }
  // lazy D2H: the raster stays in HBM until somebody looks at the pixels (raster_handle.hh:100-106 `get()`)
  const VP8Raster & get() const
  {
    if ( !state_ ) throw LogicError();
    if ( !state_->host ) {
      auto r = std::unique_ptr<VP8Raster>( new VP8Raster( state_->owner->width, state_->owner->height ) );
      if ( state_->frame_index >= 0 )
        check( aa_stream_download( state_->owner->stream, state_->frame_index, &r->Y().at( 0, 0 ), &r->U().at( 0, 0 ), &r->V().at( 0, 0 ) ) );
      state_->host = std::move( r );
    }
    return *state_->host;
  }
  operator const VP8Raster & () const { return get(); }
  int frame_index() const { return state_ ? state_->frame_index : -1; }
  const std::shared_ptr<detail::StreamOwner> & owner() const { if ( !state_ ) throw LogicError(); return state_->owner; }
  // HashCachedRaster::hash = BaseRaster::raw_hash (raster.cc:52-61, raster_handle.cc:196-206)
  size_t hash() const
  {
    if ( !state_ ) throw LogicError();
    if ( state_->frame_index >= 0 ) { uint64_t h = 0; check( aa_stream_raster_hash( state_->owner->stream, state_->frame_index, &h ) ); return h; }
    const VP8Raster & r = get();
    size_t h = 0;
    for ( const Plane * p : { &r.Y(), &r.U(), &r.V() } ) for ( const uint8_t v : *p ) detail::hash_combine( h, v );
    return h;
  }
  bool operator==( const RasterHandle & o ) const { return hash() == o.hash(); }      // raster_handle.cc:184-194: by hash
  bool operator!=( const RasterHandle & o ) const { return !operator==( o ); }
};

enum reference_frame { CURRENT_FRAME, LAST_FRAME, GOLDEN_FRAME, ALTREF_FRAME };     // modemv_data.hh

struct References
{
  RasterHandle last, golden, alternative;
  References() = default;
  References( const RasterHandle & l, const RasterHandle & g, const RasterHandle & a ) : last( l ), golden( g ), alternative( a ) {}
  References( MutableRasterHandle && raster ) : last( std::move( raster ) ), golden( last ), alternative( last ) {}   // decoder.cc:165-169
  References( const uint16_t width, const uint16_t height ) : References( MutableRasterHandle( width, height ) ) {}     // decoder.cc:161-163
  const VP8Raster & at( const reference_frame reference_id ) const
  {
    switch ( reference_id ) {
    case LAST_FRAME: return last;
    case GOLDEN_FRAME: return golden;
    case ALTREF_FRAME: return alternative;
    default: throw LogicError();
    }
  }
  bool operator==( const References & o ) const { return last == o.last && golden == o.golden && alternative == o.alternative; }
  bool operator!=( const References & o ) const { return !operator==( o ); }
};

// ---------------------------------------------------------------- decoder state files (decoder/enc_state_serializer.hh:58-179)
// Byte-compatible with the reference: a file written by either implementation loads in the other (aa_stream_serialize).
class EncoderStateSerializer
{
  std::vector<uint8_t> data_;
public:
  void append( const std::vector<uint8_t> & bytes ) { data_.insert( data_.end(), bytes.begin(), bytes.end() ); }
  const std::vector<uint8_t> & data() const { return data_; }
  void write( FILE * file ) const { if ( !data_.empty() && std::fwrite( data_.data(), data_.size(), 1, file ) != 1 ) throw std::runtime_error( "fwrite returned short write" ); }
  void write( const std::string & filename ) const
  {
    FILE * f = std::fopen( filename.c_str(), "wb" );
    if ( !f ) throw std::runtime_error( "cannot open " + filename );
    try { write( f ); } catch ( ... ) { std::fclose( f ); throw; }
    std::fclose( f );
  }
};
class EncoderStateDeserializer
{
  std::vector<uint8_t> data_;
public:
  explicit EncoderStateDeserializer( const std::string & filename )
  {
    std::ifstream in( filename, std::ios::binary );
    if ( !in ) throw std::runtime_error( "cannot open " + filename );
    data_.assign( std::istreambuf_iterator<char>( in ), std::istreambuf_iterator<char>() );
  }
  explicit EncoderStateDeserializer( const char * filename ) : EncoderStateDeserializer( std::string( filename ) ) {}
  explicit EncoderStateDeserializer( std::vector<uint8_t> bytes ) : data_( std::move( bytes ) ) {}
  const std::vector<uint8_t> & data() const { return data_; }
  size_t size() const { return data_.size(); }
  template <typename T, typename F, typename... Ps>
  static T build( F f, Ps... ps ) { EncoderStateDeserializer idata( f ); return T::deserialize( idata, ps... ); }     // enc_state_serializer.hh:125-128
};

// ---------------------------------------------------------------- DecoderState and what it is made of (decoder/decoder.hh:57-225)
// Plain values with the reference's field names.  hash() / serialize() / deserialize() go through the C ABI (a scratch
// aa_parser loaded with the value), so they are the product's code, not a second implementation.
struct ProbabilityTables
{
  uint8_t coeff_probs[4][8][3][11];
  uint8_t y_mode_probs[4], uv_mode_probs[3], motion_vector_probs[2][19];
  ProbabilityTables() { std::memset( this, 0, sizeof *this ); }
  bool operator==( const ProbabilityTables & o ) const { return std::memcmp( this, &o, sizeof *this ) == 0; }
  bool operator!=( const ProbabilityTables & o ) const { return !operator==( o ); }
};
struct FilterAdjustments
{
  int8_t loopfilter_ref_adjustments[4] = { 0, 0, 0, 0 }, loopfilter_mode_adjustments[4] = { 0, 0, 0, 0 };
  bool operator==( const FilterAdjustments & o ) const
  { return std::memcmp( loopfilter_ref_adjustments, o.loopfilter_ref_adjustments, 4 ) == 0 && std::memcmp( loopfilter_mode_adjustments, o.loopfilter_mode_adjustments, 4 ) == 0; }
};
struct SegmentationMap       // TwoD<uint8_t> of the macroblocks (the reference sizes its map by PIXEL dimensions and uses this corner)
{
  unsigned width_ = 0, height_ = 0;
  std::vector<uint8_t> storage_;
  SegmentationMap() = default;
  SegmentationMap( const unsigned w, const unsigned h, const uint8_t v ) : width_( w ), height_( h ), storage_( size_t( w ) * h, v ) {}
  uint8_t & at( const unsigned column, const unsigned row ) { return storage_.at( size_t( row ) * width_ + column ); }
  const uint8_t & at( const unsigned column, const unsigned row ) const { return storage_.at( size_t( row ) * width_ + column ); }
  unsigned width() const { return width_; }
  unsigned height() const { return height_; }
  bool operator==( const SegmentationMap & o ) const { return width_ == o.width_ && height_ == o.height_ && storage_ == o.storage_; }
};
struct Segmentation
{
  bool absolute_segment_adjustments = false;
  int8_t segment_quantizer_adjustments[4] = { 0, 0, 0, 0 }, segment_filter_adjustments[4] = { 0, 0, 0, 0 };
  SegmentationMap map;
  Segmentation() = default;
  Segmentation( const unsigned mb_width, const unsigned mb_height ) : map( mb_width, mb_height, 3 ) {}
  bool operator==( const Segmentation & o ) const
  {
    return absolute_segment_adjustments == o.absolute_segment_adjustments && map == o.map
           && std::memcmp( segment_quantizer_adjustments, o.segment_quantizer_adjustments, 4 ) == 0
           && std::memcmp( segment_filter_adjustments, o.segment_filter_adjustments, 4 ) == 0;
  }
};

struct DecoderState
{
  uint16_t width, height;
  ProbabilityTables probability_tables;
  Optional<Segmentation> segmentation;
  Optional<FilterAdjustments> filter_adjustments;

  DecoderState( const unsigned s_width, const unsigned s_height )          // defaults of a fresh decoder (decoder.cc:226-232)
    : width( s_width ), height( s_height )
  {
    aa_parser * p = nullptr;
    check( aa_parser_create( width, height, &p ) );
    std::vector<uint8_t> blob( aa_parser_state_size( p ) );
    const aa_status st = aa_parser_export_state( p, blob.data(), blob.size() );
    aa_parser_destroy( p );
    check( st );
    *this = from_blob( blob );
  }
  bool operator==( const DecoderState & o ) const
  {
    return width == o.width && height == o.height && probability_tables == o.probability_tables
           && segmentation == o.segmentation && filter_adjustments == o.filter_adjustments;
  }
  bool operator!=( const DecoderState & o ) const { return !operator==( o ); }

  // the C ABI's flat form (aa_parser_export_state / aa_stream_export_state): "AAST" u16 version, width, height, 1101
  // probabilities, segmentation {enabled, absolute, quant[4], lf[4]}, filter adjustments {enabled, ref[4], mode[4]}, map
  std::vector<uint8_t> to_blob() const
  {
    const unsigned mbw = ( width + 15u ) / 16u, mbh = ( height + 15u ) / 16u;
    std::vector<uint8_t> b( 4 + 6 + 1101 + 10 + 9 + size_t( mbw ) * mbh, 0 );
    uint8_t * o = b.data();
    std::memcpy( o, "AAST", 4 ); o += 4;
    const uint16_t hdr[3] = { 1, width, height };
    std::memcpy( o, hdr, 6 ); o += 6;
    std::memcpy( o, probability_tables.coeff_probs, 1056 ); std::memcpy( o + 1056, probability_tables.y_mode_probs, 4 );
    std::memcpy( o + 1060, probability_tables.uv_mode_probs, 3 ); std::memcpy( o + 1063, probability_tables.motion_vector_probs, 38 ); o += 1101;
    if ( segmentation.initialized() ) {
      const Segmentation & sg = segmentation.get();
      o[0] = 1; o[1] = sg.absolute_segment_adjustments; std::memcpy( o + 2, sg.segment_quantizer_adjustments, 4 ); std::memcpy( o + 6, sg.segment_filter_adjustments, 4 );
    }
    o += 10;
    if ( filter_adjustments.initialized() ) {
      o[0] = 1; std::memcpy( o + 1, filter_adjustments.get().loopfilter_ref_adjustments, 4 ); std::memcpy( o + 5, filter_adjustments.get().loopfilter_mode_adjustments, 4 );
    }
    o += 9;
    if ( segmentation.initialized() && segmentation.get().map.width() == mbw && segmentation.get().map.height() == mbh )
      std::memcpy( o, segmentation.get().map.storage_.data(), size_t( mbw ) * mbh );
    else std::memset( o, 3, size_t( mbw ) * mbh );
    return b;
  }
  static DecoderState from_blob( const std::vector<uint8_t> & b )
  {
    if ( b.size() < 4 + 6 + 1101 + 19 || std::memcmp( b.data(), "AAST", 4 ) != 0 ) throw Invalid( "decoder state: not a state blob" );
    uint16_t hdr[3];
    std::memcpy( hdr, b.data() + 4, 6 );
    DecoderState st( hdr[1], hdr[2], 0 );
    const unsigned mbw = ( st.width + 15u ) / 16u, mbh = ( st.height + 15u ) / 16u;
    if ( b.size() != 4 + 6 + 1101 + 19 + size_t( mbw ) * mbh ) throw Invalid( "decoder state: blob size" );
    const uint8_t * i = b.data() + 10;
    std::memcpy( st.probability_tables.coeff_probs, i, 1056 ); std::memcpy( st.probability_tables.y_mode_probs, i + 1056, 4 );
    std::memcpy( st.probability_tables.uv_mode_probs, i + 1060, 3 ); std::memcpy( st.probability_tables.motion_vector_probs, i + 1063, 38 ); i += 1101;
    if ( i[0] ) {
      Segmentation sg( mbw, mbh );
      sg.absolute_segment_adjustments = i[1]; std::memcpy( sg.segment_quantizer_adjustments, i + 2, 4 ); std::memcpy( sg.segment_filter_adjustments, i + 6, 4 );
      std::memcpy( sg.map.storage_.data(), i + 19, size_t( mbw ) * mbh );
      st.segmentation.initialize( std::move( sg ) );
    }
    i += 10;
    if ( i[0] ) {
      FilterAdjustments fa;
      std::memcpy( fa.loopfilter_ref_adjustments, i + 1, 4 ); std::memcpy( fa.loopfilter_mode_adjustments, i + 5, 4 );
      st.filter_adjustments.initialize( fa );
    }
    return st;
  }
  size_t hash() const                                            // DecoderState::hash, decoder.cc:266-281
  {
    ScratchParser p( *this );
    uint64_t h = 0;
    check( aa_parser_state_hash( p.p, &h ) );
    return h;
  }
  size_t serialize( EncoderStateSerializer & odata ) const;      // decoder.cc:283-313 (defined below the serializer)
  static DecoderState deserialize( EncoderStateDeserializer & idata );   // decoder.cc:315-330: the whole input is one DECODER_STATE
private:
  DecoderState( const uint16_t w, const uint16_t h, int ) : width( w ), height( h ) {}
  struct ScratchParser
  {
    aa_parser * p = nullptr;
    explicit ScratchParser( const DecoderState & st )
    {
      check( aa_parser_create( st.width, st.height, &p ) );
      const std::vector<uint8_t> b = st.to_blob();
      const aa_status s = aa_parser_import_state( p, b.data(), b.size() );
      if ( s != AA_OK ) { aa_parser_destroy( p ); check( s ); }
    }
    ~ScratchParser() { aa_parser_destroy( p ); }
    ScratchParser( const ScratchParser & ) = delete;
    ScratchParser & operator=( const ScratchParser & ) = delete;
  };
};

class DecoderHash                                                 // decoder.hh:227-242, decoder.cc:149-153,482-514
{
  size_t state_hash_, last_hash_, golden_hash_, alt_hash_;
public:
  DecoderHash( const size_t state_hash, const size_t last_hash, const size_t golden_hash, const size_t alt_hash )
    : state_hash_( state_hash ), last_hash_( last_hash ), golden_hash_( golden_hash ), alt_hash_( alt_hash ) {}
  size_t hash() const
  {
    size_t h = 0;
    detail::hash_combine( h, state_hash_ ); detail::hash_combine( h, last_hash_ ); detail::hash_combine( h, golden_hash_ ); detail::hash_combine( h, alt_hash_ );
    return h;
  }
  std::string str() const
  {
    char buf[160];
    std::snprintf( buf, sizeof buf, "%zx (%zx_%zx_%zx_%zx)", hash(), state_hash_, last_hash_, golden_hash_, alt_hash_ );
    return buf;
  }
  bool operator==( const DecoderHash & o ) const { return state_hash_ == o.state_hash_ && last_hash_ == o.last_hash_ && golden_hash_ == o.golden_hash_ && alt_hash_ == o.alt_hash_; }
  bool operator!=( const DecoderHash & o ) const { return !operator==( o ); }
};

// UncompressedChunk (uncompressed_chunk.hh:48-76): the frame tag read and checked, the frame itself still a view of the caller's bytes
enum CorruptionLevel { NO_CORRUPTION, CORRUPTED_RESIDUES, CORRUPTED_FIRST_PARTITION, CORRUPTED_FRAME };
class UncompressedChunk
{
  Chunk frame_;
  bool key_frame_ = false, show_frame_ = false, experimental_ = false, accept_partial_ = false;
  CorruptionLevel corruption_level_ = NO_CORRUPTION;
  friend class Decoder;
public:
  UncompressedChunk( const Chunk & frame, const uint16_t expected_width, const uint16_t expected_height, const bool accept_partial )
    : frame_( frame ), accept_partial_( accept_partial )
  {
    int key = 0, show = 0, experimental = 0, corruption = 0;
    check( aa_parse_frame_tag( frame.buffer(), frame.size(), expected_width, expected_height, accept_partial ? 1 : 0, &key, &show, &experimental, &corruption ) );
    key_frame_ = key != 0; show_frame_ = show != 0; experimental_ = experimental != 0; corruption_level_ = static_cast<CorruptionLevel>( corruption );
  }
  bool key_frame() const { return key_frame_; }
  bool show_frame() const { return show_frame_; }
  bool experimental() const { return experimental_; }
  CorruptionLevel corruption_level() const { return corruption_level_; }
};

// KeyFrame / InterFrame (frame.hh:46-127) as the callers of the two-step decode use them (frontend/xc-enc.cc:286-300): what
// parse_frame returns and decode_frame takes.  The macroblocks themselves stay in the decoder (records in HBM or in its
// staging memory); callers that walk or edit macroblocks (xc-dump, xc-terminate-chunk) are not served by this type.
namespace detail {
struct ParsedFrame
{
  std::shared_ptr<StreamOwner> owner;
  int index = -1;
  aa_frame_header header;
  ParsedFrame() { std::memset( &header, 0, sizeof header ); }
};
}
class KeyFrame
{
  detail::ParsedFrame f_;
  friend class Decoder;
  explicit KeyFrame( detail::ParsedFrame f ) : f_( std::move( f ) ) {}
public:
  bool show_frame() const { return f_.header.show_frame != 0; }
  uint8_t dct_partition_count() const { return f_.header.num_dct_partitions; }
  const aa_frame_header & native_header() const { return f_.header; }
};
class InterFrame
{
  detail::ParsedFrame f_;
  friend class Decoder;
  explicit InterFrame( detail::ParsedFrame f ) : f_( std::move( f ) ) {}
public:
  bool show_frame() const { return f_.header.show_frame != 0; }
  uint8_t dct_partition_count() const { return f_.header.num_dct_partitions; }
  const aa_frame_header & native_header() const { return f_.header; }
};

class Decoder
{
  std::shared_ptr<detail::StreamOwner> owner_;
  // Handles of the frames the stream's References currently name, and snapshots for rasters no frame of this decoder
  // produced (blank, imported).  Everything else lives exactly as long as the caller keeps its RasterHandle.
  RasterHandle refs_[3];
  int ref_slot_[3] = { -1, -1, -1 };       // identity of the raster each of them is (aa_stream_reference_slots)
  void sync_references( const RasterHandle * fresh )
  {
    int idx[3] = { -1, -1, -1 }, slot[3] = { -1, -1, -1 };
    check( aa_stream_references( owner_->stream, &idx[0], &idx[1], &idx[2] ) );
    check( aa_stream_reference_slots( owner_->stream, slot ) );
    RasterHandle next[3];
    for ( int i = 0; i < 3; i++ ) {
      for ( int k = 0; k < i && !next[i].state_; k++ ) if ( slot[k] == slot[i] ) next[i] = next[k];
      for ( int k = 0; k < 3 && !next[i].state_; k++ ) if ( refs_[k].state_ && ref_slot_[k] == slot[i] ) next[i] = refs_[k];   // still the same raster
      if ( !next[i].state_ && fresh && idx[i] >= 0 && fresh->frame_index() == idx[i] ) next[i] = *fresh;
      if ( !next[i].state_ ) next[i] = RasterHandle::snapshot( owner_, i );     // blank / imported: no frame of ours made it
    }
    for ( int i = 0; i < 3; i++ ) { refs_[i] = next[i]; ref_slot_[i] = slot[i]; }
  }
  std::pair<bool, RasterHandle> adopt( const int index, const bool shown )
  {
    RasterHandle h( owner_, index );
    sync_references( &h );
    return std::make_pair( shown, h );
  }
public:
  Decoder( const uint16_t width, const uint16_t height ) : Decoder( GpuContext::process_default(), width, height ) {}
  Decoder( std::shared_ptr<GpuContext> ctx, const uint16_t width, const uint16_t height )
    : owner_( std::make_shared<detail::StreamOwner>( std::move( ctx ), width, height ) ) { sync_references( nullptr ); }
  // Decoder( DecoderState, References ) (decoder.cc:43-46): continue from a state somebody else reached
  Decoder( const DecoderState & state, const References & references ) : Decoder( GpuContext::process_default(), state, references ) {}
  Decoder( std::shared_ptr<GpuContext> ctx, const DecoderState & state, const References & references )
    : owner_( std::make_shared<detail::StreamOwner>( std::move( ctx ), state.width, state.height ) )
  {
    const std::vector<uint8_t> blob = state.to_blob();
    check( aa_stream_import_state( owner_->stream, blob.data(), blob.size() ) );
    const RasterHandle * in[3] = { &references.last, &references.golden, &references.alternative };
    const void * planes[3][3];
    int is_host[3];
    for ( int i = 0; i < 3; i++ ) {
      const RasterHandle & h = *in[i];
      const VP8Raster * host = nullptr;
      if ( h.frame_index() >= 0 && h.owner()->ctx == owner_->ctx ) {      // stays in HBM: device-to-device
        void * y, * u, * v;
        check( aa_stream_raster_device( h.owner()->stream, h.frame_index(), &y, &u, &v ) );
        planes[i][0] = y; planes[i][1] = u; planes[i][2] = v; is_host[i] = 0;
      } else {
        host = &h.get();
        if ( host->display_width() != state.width || host->display_height() != state.height ) throw Invalid( "reference raster size differs from the decoder state's" );
        planes[i][0] = &host->Y().at( 0, 0 ); planes[i][1] = &host->U().at( 0, 0 ); planes[i][2] = &host->V().at( 0, 0 ); is_host[i] = 1;
      }
    }
    check( aa_stream_set_references( owner_->stream, planes, is_host ) );
    sync_references( nullptr );
  }

  uint16_t get_width() const { return owner_->width; }
  uint16_t get_height() const { return owner_->height; }

  // Decoder::get_frame_output (decoder.cc:125-135): (shown, raster) -- hidden frames still update the references
  std::pair<bool, RasterHandle> get_frame_output( const Chunk & compressed_frame )
  {
    int index = -1, shown = 0;
    check( aa_stream_decode( owner_->stream, compressed_frame.buffer(), compressed_frame.size(), &index, &shown ) );
    return adopt( index, shown != 0 );
  }
  // N independent decoders, one frame each, as ONE batch step on the GPU (aa_decode_batch): what fills the chip when an
  // ExCamera bundle or a set of streams is decoded (the reference loops over its decoders one after the other).
  // All decoders must live on the same GpuContext.  Returns (shown, raster) per decoder.
  // The loop-filter level search of Encoder::apply_best_loopfilter_settings (encoder.cc:459-516) for the frame an encoder is
  // about to write (serialised with any provisional level): candidates level_lo..level_hi filtered and scored against `original`
  // as ONE GPU batch; this decoder is not advanced.  -> { best level, its SSIM } by the reference's rule.
  std::pair<uint8_t, double> search_loopfilter_level( const Chunk & frame, const VP8Raster & original, const uint8_t level_lo, const uint8_t level_hi )
  {
    if ( original.width() != VP8Raster( owner_->width, owner_->height ).width() ) throw std::invalid_argument( "search_loopfilter_level: original of another size" );
    int best = 0; double q = 0;
    check( aa_stream_lf_search( owner_->stream, frame.buffer(), frame.size(), &original.Y().at( 0, 0 ), level_lo, level_hi, &best, &q, nullptr, nullptr ) );
    return { static_cast<uint8_t>( best ), q };
  }

  static std::vector<std::pair<bool, RasterHandle>> get_frame_outputs( const std::vector<Decoder *> & decoders, const std::vector<Chunk> & frames )
  {
    if ( decoders.size() != frames.size() || decoders.empty() ) throw std::invalid_argument( "get_frame_outputs: one frame per decoder" );
    std::vector<aa_stream *> streams( decoders.size() );
    std::vector<int> index( decoders.size() ), shown( decoders.size() );
    for ( size_t i = 0; i < decoders.size(); i++ ) {
      aa_frame_header h;
      check( aa_stream_parse( decoders[i]->owner_->stream, frames[i].buffer(), frames[i].size(), &index[i], &h ) );
      shown[i] = h.show_frame; streams[i] = decoders[i]->owner_->stream;
    }
    check( aa_decode_batch( decoders[0]->owner_->ctx->get(), streams.data(), static_cast<int>( streams.size() ), index.data() ) );
    std::vector<std::pair<bool, RasterHandle>> out;
    for ( size_t i = 0; i < decoders.size(); i++ ) out.push_back( decoders[i]->adopt( index[i], shown[i] != 0 ) );
    return out;
  }
  // LOOK-AHEAD of ONE stream (not in the reference, whose Player decodes frame by frame: player.cc:134-144).  hand_over() gives the next
  // frames of this decoder's stream to the library in one call -- the header pre-pass runs in the call, every frame BODY is then an
  // independent chain that the context's host lanes parse side by side (VP8 has no backward adaptation: decoder_state.hh:92-97) --
  // and returns at once; decode_handed_over( k ) reconstructs the k-th of them (in order) and gives what get_frame_output gives.
  // A frame the bitstream parser refuses is reported when ITS turn comes, as frame-by-frame decoding would (the frames in front of it
  // decode; the ones behind it in the same call were not appended and are handed over again by the caller).
  struct HandedOver { int index; aa_status status; std::string error; };
  std::vector<HandedOver> hand_over( const std::vector<Chunk> & frames )
  {
    std::vector<aa_frame_in> in( frames.size() );
    std::vector<int> index( frames.size(), -1 );
    for ( size_t i = 0; i < frames.size(); i++ ) { in[i].stream = owner_->stream; in[i].data = frames[i].buffer(); in[i].size = frames[i].size(); }
    const aa_status st = frames.empty() ? AA_OK : aa_submit_frames( owner_->ctx->get(), in.data(), static_cast<int>( in.size() ), index.data(), 0 );
    const std::string msg = st == AA_OK ? std::string() : std::string( aa_last_error() );
    std::vector<HandedOver> out;
    bool failed = false;
    for ( size_t i = 0; i < frames.size(); i++ ) {
      if ( index[i] >= 0 && !failed ) { out.push_back( { index[i], AA_OK, std::string() } ); continue; }
      if ( !failed ) { out.push_back( { -1, st == AA_OK ? AA_ERR_LOGIC : st, msg } ); failed = true; }      // the first refused frame carries the error;
      else out.push_back( { -1, AA_ERR_LOGIC, std::string() } );                                           // the ones behind it were not looked at
    }
    return out;
  }
  std::pair<bool, RasterHandle> decode_handed_over( const HandedOver & f )
  {
    if ( f.status != AA_OK ) {
      switch ( f.status ) {
      case AA_ERR_INVALID: throw Invalid( f.error, 0 );
      case AA_ERR_UNSUPPORTED: throw Unsupported( f.error, 0 );
      case AA_ERR_OUT_OF_RANGE: throw std::out_of_range( f.error );
      default: throw DeviceError( f.error );
      }
    }
    aa_frame_header h;
    check( aa_stream_frame_header( owner_->stream, f.index, &h ) );      // (waits for this frame's parse, not for the others')
    aa_stream * one[1] = { owner_->stream };
    check( aa_decode_batch( owner_->ctx->get(), one, 1, &f.index ) );
    return adopt( f.index, h.show_frame != 0 );
  }
  // The two-step form (decoder.hh:262-270, decoder.cc:83-118): decompress_frame reads the tag, parse_frame<KeyFrame|InterFrame> runs
  // DecoderState::parse_and_apply (the entropy decode, on a host core), decode_frame reconstructs + filters + updates the
  // references on the GPU.  A frame must be decoded before the next one is parsed into the same decoder (as every caller does).
  UncompressedChunk decompress_frame( const Chunk & compressed_frame ) const
  { return UncompressedChunk( compressed_frame, owner_->width, owner_->height, error_concealment() ); }
  template <class FrameType>
  FrameType parse_frame( const UncompressedChunk & decompressed_frame )
  {
    if ( std::is_same<FrameType, KeyFrame>::value != decompressed_frame.key_frame() ) throw LogicError();     // (the reference asserts)
    const bool was = error_concealment();
    if ( was != decompressed_frame.accept_partial_ ) set_error_concealment( decompressed_frame.accept_partial_ );
    detail::ParsedFrame f;
    f.owner = owner_;
    const aa_status st = aa_stream_parse( owner_->stream, decompressed_frame.frame_.buffer(), decompressed_frame.frame_.size(), &f.index, &f.header );
    if ( was != decompressed_frame.accept_partial_ ) set_error_concealment( was );
    check( st );
    return FrameType( std::move( f ) );
  }
  template <class FrameType>
  std::pair<bool, RasterHandle> decode_frame( const FrameType & frame )
  {
    if ( frame.f_.owner != owner_ ) throw std::invalid_argument( "decode_frame: the frame was parsed by another decoder" );
    aa_stream * one[1] = { owner_->stream };
    check( aa_decode_batch( owner_->ctx->get(), one, 1, &frame.f_.index ) );
    return adopt( frame.f_.index, frame.show_frame() );
  }
  Optional<RasterHandle> parse_and_decode_frame( const Chunk & compressed_frame )   // decoder.cc:137-141
  {
    const std::pair<bool, RasterHandle> out = get_frame_output( compressed_frame );
    return make_optional( out.first, out.second );
  }
  References get_references() const { return References { refs_[0], refs_[1], refs_[2] }; }
  const VP8Raster & example_raster() const { return refs_[0].get(); }
  aa_stream * native_handle() const { return owner_->stream; }
  // decoder.hh:298-299: frames that end early are accepted instead of thrown at the caller (salsify-receiver.cc:192)
  void set_error_concealment( const bool val ) { check( aa_stream_set_error_concealment( owner_->stream, val ? 1 : 0 ) ); }
  bool error_concealment() const { return aa_stream_error_concealment( owner_->stream ) != 0; }

  DecoderState get_state() const
  {
    std::vector<uint8_t> blob( aa_stream_state_size( owner_->stream ) );
    check( aa_stream_export_state( owner_->stream, blob.data(), blob.size() ) );
    return DecoderState::from_blob( blob );
  }
  DecoderHash get_hash() const                                    // decoder.cc:143-147
  {
    uint64_t parts[4];
    check( aa_stream_decoder_hash( owner_->stream, parts, nullptr ) );
    return DecoderHash( parts[0], parts[1], parts[2], parts[3] );
  }
  uint32_t minihash() const { return static_cast<uint32_t>( get_hash().hash() ); }                  // decoder.cc:516-519
  bool minihash_match( const uint32_t other_minihash ) const { return other_minihash == 0 || minihash() == other_minihash; }   // :522-529
  bool operator==( const Decoder & o ) const { return get_state() == o.get_state() && get_references() == o.get_references(); }   // decoder.cc:155-158
  bool operator!=( const Decoder & o ) const { return !operator==( o ); }

  // Decoder::serialize (decoder.cc:54-69): DecoderState + the LAST reference raster, in the reference's wire format
  size_t serialize( EncoderStateSerializer & odata ) const
  {
    size_t n = 0;
    check( aa_stream_serialize( owner_->stream, nullptr, 0, &n ) );
    std::vector<uint8_t> bytes( n );
    check( aa_stream_serialize( owner_->stream, bytes.data(), bytes.size(), &n ) );
    odata.append( bytes );
    return n;
  }
  // Decoder::deserialize (decoder.cc:48-52,71-81); the frame size comes from the blob ([DECODER][len][DECODER_STATE][len][w][h]...)
  static Decoder deserialize( EncoderStateDeserializer & idata ) { return deserialize( idata, GpuContext::process_default() ); }
  static Decoder deserialize( EncoderStateDeserializer & idata, std::shared_ptr<GpuContext> ctx )
  {
    const std::vector<uint8_t> & b = idata.data();
    if ( b.size() < 14 || b[0] != 11 || b[5] != 4 ) throw Invalid( "decoder state: not a serialized Decoder" );
    Decoder d( std::move( ctx ), static_cast<uint16_t>( b[10] | ( b[11] << 8 ) ), static_cast<uint16_t>( b[12] | ( b[13] << 8 ) ) );
    check( aa_stream_deserialize( d.owner_->stream, b.data(), b.size() ) );
    for ( RasterHandle & r : d.refs_ ) r = RasterHandle();
    d.sync_references( nullptr );           // golden and alternative alias the loaded LAST raster (decoder.cc:171-175)
    return d;
  }
};

inline size_t DecoderState::serialize( EncoderStateSerializer & odata ) const
{
  ScratchParser p( *this );
  size_t n = 0;
  check( aa_parser_serialize_state( p.p, nullptr, 0, &n ) );
  std::vector<uint8_t> bytes( n );
  check( aa_parser_serialize_state( p.p, bytes.data(), bytes.size(), &n ) );
  odata.append( bytes );
  return n;
}
inline DecoderState DecoderState::deserialize( EncoderStateDeserializer & idata )
{
  const std::vector<uint8_t> & b = idata.data();
  if ( b.size() < 9 || b[0] != 4 ) throw Invalid( "decoder state: expected DECODER_STATE" );
  const uint16_t w = static_cast<uint16_t>( b[5] | ( b[6] << 8 ) ), h = static_cast<uint16_t>( b[7] | ( b[8] << 8 ) );
  aa_parser * p = nullptr;
  check( aa_parser_create( w, h, &p ) );
  aa_status st = aa_parser_deserialize_state( p, b.data(), b.size() );
  std::vector<uint8_t> blob( aa_parser_state_size( p ) );
  if ( st == AA_OK ) st = aa_parser_export_state( p, blob.data(), blob.size() );
  aa_parser_destroy( p );
  check( st );
  return from_blob( blob );
}

class FramePlayer
{
  uint16_t width_, height_;
protected:
  Decoder decoder_;
public:
  FramePlayer( const uint16_t width, const uint16_t height ) : width_( width ), height_( height ), decoder_( width, height ) {}
  explicit FramePlayer( EncoderStateDeserializer & idata )                                  // player.cc:43-54
    : width_( 0 ), height_( 0 ), decoder_( Decoder::deserialize( idata ) ) { width_ = decoder_.get_width(); height_ = decoder_.get_height(); }
  static FramePlayer deserialize( EncoderStateDeserializer & idata ) { return FramePlayer( idata ); }
  size_t serialize( EncoderStateSerializer & odata ) const { return decoder_.serialize( odata ); }   // player.cc:56-58
  Optional<RasterHandle> decode( const Chunk & chunk ) { return decoder_.parse_and_decode_frame( chunk ); }   // player.cc:60-63
  const VP8Raster & example_raster() const { return decoder_.example_raster(); }
  uint16_t width() const { return width_; }
  uint16_t height() const { return height_; }
  const Decoder & current_decoder() const { return decoder_; }
  Decoder & mutable_decoder() { return decoder_; }
  void set_error_concealment( const bool value ) { decoder_.set_error_concealment( value ); }     // player.hh:72
  References current_references() const { return decoder_.get_references(); }
};

class FilePlayer : public FramePlayer
{
  IVF file_;
  unsigned int frame_no_ = 0;
  std::string filename_;
  // look-ahead (set_look_ahead; 0 = the reference's frame-by-frame decode): frames [window_first_, window_first_ + window_.size()) of the
  // file have been handed over (Decoder::hand_over), frame_no_ is the next to reconstruct
  unsigned int look_ahead_ = 0, window_first_ = 0;
  std::vector<Decoder::HandedOver> window_;
  Optional<RasterHandle> decode_next()
  {
    if ( look_ahead_ == 0 ) return decode( file_.frame( frame_no_++ ) );
    if ( frame_no_ >= window_first_ + window_.size() || frame_no_ < window_first_ ) {
      std::vector<Chunk> next;
      for ( unsigned int k = frame_no_; k < file_.frame_count() && k < frame_no_ + look_ahead_; k++ ) next.push_back( file_.frame( k ) );
      window_ = decoder_.hand_over( next );
      window_first_ = frame_no_;
    }
    const Decoder::HandedOver f = window_[frame_no_ - window_first_];
    frame_no_++;
    if ( f.index < 0 && f.status == AA_ERR_LOGIC && f.error.empty() ) {   // behind a refused frame of its call: not looked at yet -- by itself, now
      window_.clear();
      return decode( file_.frame( frame_no_ - 1 ) );
    }
    const std::pair<bool, RasterHandle> out = decoder_.decode_handed_over( f );
    return make_optional( out.first, out.second );
  }
  FilePlayer( const std::string & filename, IVF && file )
    : FramePlayer( file.width(), file.height() ), file_( std::move( file ) ), filename_( filename )
  {
    if ( file_.fourcc() != "VP80" ) throw Unsupported( "not a VP8 file" );
    while ( frame_no_ < file_.frame_count() ) {          // start at the first key frame (player.cc:96-105)
      if ( !( file_.frame( frame_no_ ).octet() & 1 ) ) break;
      frame_no_++;
    }
  }
  FilePlayer( const std::string & filename, IVF && file, EncoderStateDeserializer & idata )      // player.cc:108-124: continue from a state
    : FramePlayer( idata ), file_( std::move( file ) ), filename_( filename )
  {
    if ( file_.fourcc() != "VP80" ) throw Unsupported( "not a VP8 file" );
    if ( file_.width() != decoder_.get_width() || file_.height() != decoder_.get_height() ) throw Unsupported( "state vs. file dimension mismatch" );
    if ( !decoder_.minihash_match( file_.expected_decoder_minihash() ) ) throw Invalid( "Decoder state / IVF mismatch" );   // player.cc:118-121
  }
public:
  explicit FilePlayer( const std::string & filename ) : FilePlayer( filename, IVF( filename ) ) {}
  static FilePlayer deserialize( EncoderStateDeserializer & idata, const std::string & filename ) { return FilePlayer( filename, IVF( filename ), idata ); }
  RasterHandle advance()                                  // player.cc:134-144
  {
    while ( !eof() ) {
      Optional<RasterHandle> raster = decode_next();
      if ( raster.initialized() ) return raster.get();
    }
    throw Unsupported( "hidden frames at end of file" );
  }
  // (not in the reference) hand the next `frames` frames of the file over at a time: their entropy decode runs frame-parallel on the
  // library's host lanes while earlier frames are reconstructed; the output is what frame-by-frame decoding gives
  void set_look_ahead( const unsigned int frames ) { look_ahead_ = frames; window_.clear(); }
  bool eof() const { return frame_no_ == file_.frame_count(); }
  unsigned int cur_frame_no() const { return frame_no_ - 1; }
  long unsigned int original_size() const { return file_.frame( cur_frame_no() ).size(); }
};

using Player = FilePlayer;

// ---------------------------------------------------------------- input/frame_input.hh, input/ivf_reader.{hh,cc}
class FrameInput
{
public:
  virtual Optional<RasterHandle> get_next_frame() = 0;
  virtual uint16_t display_width() = 0;
  virtual uint16_t display_height() = 0;
  virtual ~FrameInput() = default;
};
class IVFReader : public FrameInput       // ivf_reader.cc:36-46: the shown frames of a file, one after the other
{
  FilePlayer player_;
public:
  explicit IVFReader( const std::string & filename ) : player_( filename ) {}
  Optional<RasterHandle> get_next_frame() override
  {
    if ( player_.eof() ) return Optional<RasterHandle>();
    return Optional<RasterHandle>( true, player_.advance() );
  }
  uint16_t display_width() override { return player_.width(); }
  uint16_t display_height() override { return player_.height(); }
};

inline std::ostream & operator<<( std::ostream & out, const FramePlayer & player ) { return out << player.current_decoder().get_hash().str(); }   // player.cc:75-78

// ---------------------------------------------------------------- output side of the front-ends (util/file_descriptor.hh, input/yuv4mpeg.{hh,cc})
class FileDescriptor       // the subset vp8decode / xc-decode-bundle use (util/file_descriptor.hh:44-160): an owned fd, unbuffered writes
{
  int fd_ = -1;
  unsigned int write_count_ = 0;
  void write_all( const void * data, size_t n )
  {
    const char * p = static_cast<const char *>( data );
    while ( n ) {
      const ssize_t w = ::write( fd_, p, n );
      if ( w <= 0 ) throw std::runtime_error( "write: failed" );
      p += w; n -= static_cast<size_t>( w ); write_count_++;
    }
  }
public:
  FileDescriptor() = default;
  FileDescriptor( FILE * file ) : fd_( file ? ( std::fseek( file, 0, SEEK_SET ), ::dup( fileno( file ) ) ) : -1 )   // (the FILE keeps its own descriptor)
  { if ( fd_ < 0 ) throw std::runtime_error( "fopen: cannot open file" ); }
  FileDescriptor( const int s_fd ) : fd_( s_fd ) {}
  FileDescriptor( FileDescriptor && o ) noexcept : fd_( o.fd_ ), write_count_( o.write_count_ ) { o.fd_ = -1; }
  FileDescriptor & operator=( FileDescriptor && o ) noexcept { if ( this != &o ) { close(); fd_ = o.fd_; write_count_ = o.write_count_; o.fd_ = -1; } return *this; }
  FileDescriptor( const FileDescriptor & ) = delete;
  FileDescriptor & operator=( const FileDescriptor & ) = delete;
  ~FileDescriptor() { close(); }
  void close() { if ( fd_ > 2 ) (void) ::close( fd_ ); fd_ = -1; }
  const int & fd_num() const { return fd_; }
  bool valid() const { return fd_ >= 0; }
  long tell() const { return static_cast<long>( ::lseek( fd_, 0, SEEK_CUR ) ); }
  unsigned int write_count() const { return write_count_; }
  void write( const std::string & s ) { if ( s.empty() ) throw std::runtime_error( "nothing to write" ); write_all( s.data(), s.size() ); }
  void write( const Chunk & c ) { if ( c.size() ) write_all( c.buffer(), c.size() ); }
};

struct YUV4MPEGHeader      // yuv4mpeg.hh:40-65; header text yuv4mpeg.cc:85-128
{
  enum InterlacingMode { PROGRESSIVE, TOP_FIELD_FIRST, BOTTOM_FIELD_FIRST, MIXED_MODES };
  enum ColorSpace { C420jpeg, C420paldv, C420, C422, C444 };
  uint16_t width = 0, height = 0, fps_numerator = 0, fps_denominator = 0, pixel_aspect_ratio_numerator = 0, pixel_aspect_ratio_denominator = 0;
  InterlacingMode interlacing_mode = PROGRESSIVE;
  ColorSpace color_space = C420;
  YUV4MPEGHeader() = default;
  explicit YUV4MPEGHeader( const VP8Raster & r )      // yuv4mpeg.cc:44-50: display size, 24 fps, square pixels, progressive 4:2:0
    : width( r.display_width() ), height( r.display_height() ), fps_numerator( 24 ), fps_denominator( 1 ),
      pixel_aspect_ratio_numerator( 1 ), pixel_aspect_ratio_denominator( 1 ) {}
  size_t y_plane_length() const { return size_t( width ) * height; }
  size_t uv_plane_length() const { return size_t( width ) * height / 4; }
  size_t frame_length() const { return size_t( width ) * height * 3 / 2; }
  std::string to_string() const
  {
    static const char * const im = "ptbm";
    static const char * const cs[] = { "C420jpeg XYSCSS=420JPEG", "C420paldv XYSCSS=420PALDV", "C420 XYSCSS=420", "C422 XYSCSS=422", "C444 XYSCSS=444" };
    return "YUV4MPEG2 W" + std::to_string( width ) + " H" + std::to_string( height ) + " F" + std::to_string( fps_numerator ) + ":" + std::to_string( fps_denominator )
           + " I" + std::string( 1, im[interlacing_mode] ) + " A" + std::to_string( pixel_aspect_ratio_numerator ) + ":" + std::to_string( pixel_aspect_ratio_denominator )
           + " " + cs[color_space] + "\n";
  }
};
struct YUV4MPEGFrameWriter
{
  static void write( const VP8Raster & r, FileDescriptor & fd )      // yuv4mpeg.cc:309-317
  {
    fd.write( std::string( "FRAME\n" ) );
    for ( const auto & chunk : r.display_rectangle_as_planar() ) fd.write( chunk );
  }
};

inline void print_exception( const char * argv0, const std::exception & e ) { std::fprintf( stderr, "%s: %s\n", argv0, e.what() ); }

} // namespace alfalfa_amd

#ifdef ALFALFA_AMD_GLOBAL_NAMES
using alfalfa_amd::Chunk; using alfalfa_amd::Decoder; using alfalfa_amd::FilePlayer; using alfalfa_amd::FramePlayer;
using alfalfa_amd::Invalid; using alfalfa_amd::IVF; using alfalfa_amd::LogicError; using alfalfa_amd::Optional;
using alfalfa_amd::Player; using alfalfa_amd::RasterHandle; using alfalfa_amd::References; using alfalfa_amd::Unsupported;
using alfalfa_amd::VP8Raster; using alfalfa_amd::print_exception;
using alfalfa_amd::FileDescriptor; using alfalfa_amd::YUV4MPEGHeader; using alfalfa_amd::YUV4MPEGFrameWriter;
using alfalfa_amd::EncoderStateSerializer; using alfalfa_amd::EncoderStateDeserializer;
using alfalfa_amd::DecoderState; using alfalfa_amd::ProbabilityTables; using alfalfa_amd::Segmentation; using alfalfa_amd::FilterAdjustments;
using alfalfa_amd::DecoderHash; using alfalfa_amd::make_optional; using alfalfa_amd::FrameInput; using alfalfa_amd::IVFReader;
using alfalfa_amd::CURRENT_FRAME; using alfalfa_amd::LAST_FRAME; using alfalfa_amd::GOLDEN_FRAME; using alfalfa_amd::ALTREF_FRAME;
using alfalfa_amd::UncompressedChunk; using alfalfa_amd::KeyFrame; using alfalfa_amd::InterFrame; using alfalfa_amd::MutableRasterHandle;
using alfalfa_amd::CorruptionLevel; using alfalfa_amd::NO_CORRUPTION; using alfalfa_amd::CORRUPTED_RESIDUES;
using alfalfa_amd::CORRUPTED_FIRST_PARTITION; using alfalfa_amd::CORRUPTED_FRAME;
#endif
