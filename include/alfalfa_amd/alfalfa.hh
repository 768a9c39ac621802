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
// What is NOT mirrored (host-side plumbing outside the hot path): Frame<> object graphs
// (parse_frame<F>/decode_frame<F> are replaced by get_frame_output), boost-based hash()/minihash.
//   EncoderStateSerializer / EncoderStateDeserializer, Decoder / FramePlayer / FilePlayer ::serialize, ::deserialize
//               -- the reference's `.state` files, byte-compatible    (decoder/enc_state_serializer.hh, decoder.cc:48-81)
//
// Everything lives in namespace alfalfa_amd; define ALFALFA_AMD_GLOBAL_NAMES before including to also export the
// names into the global namespace (drop-in for code written against the reference headers).
#pragma once

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

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

// ---------------------------------------------------------------- Optional
template <class T>
class Optional
{
  bool initialized_ = false;
  T value_ {};
public:
  Optional() = default;
  Optional( const T & v ) : initialized_( true ), value_( v ) {}
  Optional( const bool init, const T & v ) : initialized_( init ), value_( init ? v : T() ) {}
  bool initialized() const { return initialized_; }
  const T & get() const { if ( !initialized_ ) throw std::runtime_error( "attempt to get uninitialized Optional" ); return value_; }
  const T & get_or( const T & fallback ) const { return initialized_ ? value_ : fallback; }
  void clear() { initialized_ = false; value_ = T(); }
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
  explicit GpuContext( const int device ) { check( aa_ctx_create( device, &ctx_ ) ); }
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
  int frame_index;                         // -1: the blank initial reference
  std::unique_ptr<VP8Raster> host;         // filled on first get()
};
}

class RasterHandle
{
  std::shared_ptr<detail::RasterState> state_;
public:
  RasterHandle() = default;
  RasterHandle( std::shared_ptr<detail::StreamOwner> owner, const int frame_index )
    : state_( std::make_shared<detail::RasterState>() ) { state_->owner = std::move( owner ); state_->frame_index = frame_index; }
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
  bool operator==( const RasterHandle & o ) const { return get() == o.get(); }
  bool operator!=( const RasterHandle & o ) const { return !operator==( o ); }
};

struct References
{
  RasterHandle last, golden, alternative;
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

class Decoder
{
  std::shared_ptr<detail::StreamOwner> owner_;
  std::vector<RasterHandle> handles_;      // frame index -> handle (so References can name earlier frames)
  RasterHandle handle_for( const int frame_index ) const
  {
    if ( frame_index < 0 ) return RasterHandle( owner_, -1 );
    return handles_.at( frame_index );
  }
public:
  Decoder( const uint16_t width, const uint16_t height ) : Decoder( GpuContext::process_default(), width, height ) {}
  Decoder( std::shared_ptr<GpuContext> ctx, const uint16_t width, const uint16_t height )
    : owner_( std::make_shared<detail::StreamOwner>( std::move( ctx ), width, height ) ) {}

  uint16_t get_width() const { return owner_->width; }
  uint16_t get_height() const { return owner_->height; }

  // Decoder::get_frame_output (decoder.cc:125-135): (shown, raster) -- hidden frames still update the references
  std::pair<bool, RasterHandle> get_frame_output( const Chunk & compressed_frame )
  {
    int index = -1, shown = 0;
    check( aa_stream_decode( owner_->stream, compressed_frame.buffer(), compressed_frame.size(), &index, &shown ) );
    if ( static_cast<int>( handles_.size() ) <= index ) handles_.resize( index + 1 );
    handles_[index] = RasterHandle( owner_, index );
    return std::make_pair( shown != 0, handles_[index] );
  }
  // N independent decoders, one frame each, as ONE batch step on the GPU (aa_decode_batch): what fills the chip when an
  // ExCamera bundle or a set of streams is decoded (the reference loops over its decoders one after the other).
  // All decoders must live on the same GpuContext.  Returns (shown, raster) per decoder.
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
    for ( size_t i = 0; i < decoders.size(); i++ ) {
      Decoder & d = *decoders[i];
      if ( static_cast<int>( d.handles_.size() ) <= index[i] ) d.handles_.resize( index[i] + 1 );
      d.handles_[index[i]] = RasterHandle( d.owner_, index[i] );
      out.emplace_back( shown[i] != 0, d.handles_[index[i]] );
    }
    return out;
  }
  Optional<RasterHandle> parse_and_decode_frame( const Chunk & compressed_frame )   // decoder.cc:137-141
  {
    const std::pair<bool, RasterHandle> out = get_frame_output( compressed_frame );
    return make_optional( out.first, out.second );
  }
  References get_references() const
  {
    int l = -1, g = -1, a = -1;
    check( aa_stream_references( owner_->stream, &l, &g, &a ) );
    return References { handle_for( l ), handle_for( g ), handle_for( a ) };
  }
  const VP8Raster & example_raster() const { example_ = get_references().last; return example_.get(); }
  aa_stream * native_handle() const { return owner_->stream; }

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
    return d;
  }
private:
  mutable RasterHandle example_;
};

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
  References current_references() const { return decoder_.get_references(); }
};

class FilePlayer : public FramePlayer
{
  IVF file_;
  unsigned int frame_no_ = 0;
  std::string filename_;
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
    // the reference also checks Decoder::minihash against the IVF header here; minihash (boost::hash_combine) is not provided
  }
public:
  explicit FilePlayer( const std::string & filename ) : FilePlayer( filename, IVF( filename ) ) {}
  static FilePlayer deserialize( EncoderStateDeserializer & idata, const std::string & filename ) { return FilePlayer( filename, IVF( filename ), idata ); }
  RasterHandle advance()                                  // player.cc:134-144
  {
    while ( !eof() ) {
      Optional<RasterHandle> raster = decode( file_.frame( frame_no_++ ) );
      if ( raster.initialized() ) return raster.get();
    }
    throw Unsupported( "hidden frames at end of file" );
  }
  bool eof() const { return frame_no_ == file_.frame_count(); }
  unsigned int cur_frame_no() const { return frame_no_ - 1; }
  long unsigned int original_size() const { return file_.frame( cur_frame_no() ).size(); }
};

using Player = FilePlayer;

// ---------------------------------------------------------------- output side of the front-ends (util/file_descriptor.hh, input/yuv4mpeg.{hh,cc})
class FileDescriptor       // the subset vp8decode / xc-decode-bundle use: wrap a FILE* or an fd, write strings and chunks
{
  FILE * file_ = nullptr;
  bool owned_ = false;
public:
  FileDescriptor() = default;
  explicit FileDescriptor( FILE * f ) : file_( f ), owned_( true ) { if ( !f ) throw std::runtime_error( "fopen: cannot open file" ); }
  explicit FileDescriptor( const int fd ) : file_( fd == 1 ? stdout : fd == 2 ? stderr : fdopen( fd, "wb" ) ), owned_( fd > 2 ) { if ( !file_ ) throw std::runtime_error( "fdopen failed" ); }
  FileDescriptor( FileDescriptor && o ) noexcept : file_( o.file_ ), owned_( o.owned_ ) { o.file_ = nullptr; o.owned_ = false; }
  FileDescriptor & operator=( FileDescriptor && o ) noexcept { if ( this != &o ) { close(); file_ = o.file_; owned_ = o.owned_; o.file_ = nullptr; o.owned_ = false; } return *this; }
  FileDescriptor( const FileDescriptor & ) = delete;
  FileDescriptor & operator=( const FileDescriptor & ) = delete;
  ~FileDescriptor() { close(); }
  void close() { if ( file_ && owned_ ) std::fclose( file_ ); file_ = nullptr; }
  bool valid() const { return file_ != nullptr; }
  long tell() const { return std::ftell( file_ ); }
  void write( const std::string & s ) { if ( !s.empty() && std::fwrite( s.data(), s.size(), 1, file_ ) != 1 ) throw std::runtime_error( "fwrite returned short write" ); }
  void write( const Chunk & c ) { if ( c.size() && std::fwrite( c.buffer(), c.size(), 1, file_ ) != 1 ) throw std::runtime_error( "fwrite returned short write" ); }
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
#endif
