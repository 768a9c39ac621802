// Host-side VP8 frame parser (see parser.hh for the reference functions it replaces).
#include "parser.hh"

#include <cstring>

#include "bool_reader.hh"
#include "parse_common.hh"

namespace aa {

namespace {

// Block::parse_tokens, tokens.cc:50-135.  Returns true iff a non-zero token was decoded (Q1).
// `out` must be 16 zeroed int16 (de-zigzagged positions are written).
inline bool parse_block( BoolReader & bd, const uint8_t ( *probs )[3][11], int first, int ctx, int16_t * out )
{
  static constexpr uint8_t kCat3[3] = { 173, 148, 140 };
  static constexpr uint8_t kCat4[4] = { 176, 155, 140, 135 };
  static constexpr uint8_t kCat5[5] = { 180, 157, 141, 134, 130 };
  static constexpr uint8_t kCat6[11] = { 254, 254, 243, 230, 196, 177, 153, 140, 133, 130, 129 };
  bool nonzero = false;
  int index = first;
  const uint8_t * p = probs[kBand[index]][ctx];
  if ( !bd.get( p[0] ) ) return false;   // immediate EOB
  for ( ;; ) {
    // here: "not EOB" has been established for position `index`
    while ( !bd.get( p[1] ) ) {           // ZERO token: no EOB check for the next position
      if ( ++index == 16 ) return nonzero;
      p = probs[kBand[index]][0];
    }
    nonzero = true;
    int value;
    int next_ctx;
    if ( !bd.get( p[2] ) ) { value = 1; next_ctx = 1; }
    else {
      next_ctx = 2;
      if ( !bd.get( p[3] ) ) {
        if ( !bd.get( p[4] ) ) value = 2;
        else value = 3 + bd.get( p[5] );
      } else if ( !bd.get( p[6] ) ) {
        if ( !bd.get( p[7] ) ) value = 5 + bd.get( 159 );
        else { value = 7 + 2 * bd.get( 165 ); value += bd.get( 145 ); }
      } else {
        const uint8_t * cp; int n, base;
        if ( !bd.get( p[8] ) ) {
          if ( !bd.get( p[9] ) ) { cp = kCat3; n = 3; base = 11; } else { cp = kCat4; n = 4; base = 19; }
        } else {
          if ( !bd.get( p[10] ) ) { cp = kCat5; n = 5; base = 35; } else { cp = kCat6; n = 11; base = 67; }
        }
        int inc = 0;
        for ( int i = 0; i < n; i++ ) inc = ( inc << 1 ) + bd.get( cp[i] );
        value = base + inc;
      }
    }
    if ( bd.get( 128 ) ) value = -value;
    out[kZigzag[index]] = static_cast<int16_t>( value );
    if ( ++index == 16 ) return true;
    p = probs[kBand[index]][next_ctx];
    if ( !bd.get( p[0] ) ) return true;   // EOB
  }
}

struct Quantizer { uint16_t f[6]; };   // {y_dc, y_ac, y2_dc, y2_ac, uv_dc, uv_ac}

inline int clamp_q( int q ) { return q < 0 ? 0 : ( q > 127 ? 127 : q ); }

// Quantizer::Quantizer, quantization.cc:83-93
Quantizer make_quantizer( int y_ac_qi, const int delta[5] /* y_dc, y2_dc, y2_ac, uv_dc, uv_ac */ )
{
  Quantizer q;
  q.f[1] = k_ac_qlookup[clamp_q( y_ac_qi )];
  q.f[0] = k_dc_qlookup[clamp_q( y_ac_qi + delta[0] )];
  q.f[2] = static_cast<uint16_t>( k_dc_qlookup[clamp_q( y_ac_qi + delta[1] )] * 2 );
  q.f[3] = static_cast<uint16_t>( k_ac_qlookup[clamp_q( y_ac_qi + delta[2] )] * 155 / 100 );
  q.f[4] = k_dc_qlookup[clamp_q( y_ac_qi + delta[3] )];
  q.f[5] = k_ac_qlookup[clamp_q( y_ac_qi + delta[4] )];
  if ( q.f[3] < 8 ) q.f[3] = 8;
  if ( q.f[4] > 132 ) q.f[4] = 132;
  return q;
}

} // namespace

void ProbTables::set_defaults()
{
  std::memcpy( coeff, k_default_coeff_probs, sizeof coeff );
  std::memcpy( y_mode, k_default_y_mode_probs, sizeof y_mode );
  std::memcpy( uv_mode, k_default_uv_mode_probs, sizeof uv_mode );
  std::memcpy( mv, k_default_mv_probs, sizeof mv );
}

Parser::Parser( uint16_t width, uint16_t height )
  : width_( width ), height_( height ), mbw_( ( width + 15u ) / 16u ), mbh_( ( height + 15u ) / 16u ),
    above_nz_( static_cast<size_t>( mbw_ ) * 9 )
{
  probs_.set_defaults();
  seg_.map.assign( static_cast<size_t>( mbw_ ) * mbh_, 3 );
}

size_t Parser::state_size() const { return 4 + 6 + 1101 + 10 + 9 + seg_.map.size(); }

void Parser::export_state( uint8_t * out ) const
{
  std::memcpy( out, "AAST", 4 ); out += 4;
  const uint16_t hdr[3] = { 1, width_, height_ };
  std::memcpy( out, hdr, 6 ); out += 6;
  std::memcpy( out, probs_.coeff, 1056 ); std::memcpy( out + 1056, probs_.y_mode, 4 );
  std::memcpy( out + 1060, probs_.uv_mode, 3 ); std::memcpy( out + 1063, probs_.mv, 38 ); out += 1101;
  out[0] = seg_.enabled; out[1] = seg_.absolute; std::memcpy( out + 2, seg_.quant, 4 ); std::memcpy( out + 6, seg_.lf, 4 ); out += 10;
  out[0] = fadj_.enabled; std::memcpy( out + 1, fadj_.ref, 4 ); std::memcpy( out + 5, fadj_.mode, 4 ); out += 9;
  std::memcpy( out, seg_.map.data(), seg_.map.size() );
}

void Parser::import_state( const uint8_t * in, size_t size )
{
  uint16_t hdr[3];
  if ( size != state_size() || std::memcmp( in, "AAST", 4 ) != 0 ) throw ParseError( AA_ERR_ARGUMENT, "import_state: not a decoder-state blob of this geometry" );
  std::memcpy( hdr, in + 4, 6 );
  if ( hdr[0] != 1 || hdr[1] != width_ || hdr[2] != height_ ) throw ParseError( AA_ERR_ARGUMENT, "import_state: version or frame size mismatch" );
  in += 10;
  std::memcpy( probs_.coeff, in, 1056 ); std::memcpy( probs_.y_mode, in + 1056, 4 );
  std::memcpy( probs_.uv_mode, in + 1060, 3 ); std::memcpy( probs_.mv, in + 1063, 38 ); in += 1101;
  seg_.enabled = in[0]; seg_.absolute = in[1]; std::memcpy( seg_.quant, in + 2, 4 ); std::memcpy( seg_.lf, in + 6, 4 ); in += 10;
  fadj_.enabled = in[0]; std::memcpy( fadj_.ref, in + 1, 4 ); std::memcpy( fadj_.mode, in + 5, 4 ); in += 9;
  std::memcpy( seg_.map.data(), in, seg_.map.size() );
}

// ---- the reference's state wire format (enc_state_serializer.hh:43-56,64-86; decoder.cc:283-330,342-376,396-474;
// probability_tables.cc:126-195) ----
namespace {
enum : uint8_t { T_PROB_TABLE, T_FILT_ADJ, T_SEGM_ABS, T_SEGM_REL, T_DECODER_STATE, T_OPT_EMPTY, T_OPT_FULL };
void put32( std::vector<uint8_t> & o, uint32_t v ) { for ( int i = 0; i < 4; i++ ) o.push_back( static_cast<uint8_t>( v >> ( 8 * i ) ) ); }
void put16( std::vector<uint8_t> & o, uint16_t v ) { o.push_back( static_cast<uint8_t>( v ) ); o.push_back( static_cast<uint8_t>( v >> 8 ) ); }
struct Reader
{
  const uint8_t * p; size_t n, at = 0;
  void need( size_t k ) const { if ( at + k > n ) throw ParseError( AA_ERR_INVALID, "invalid decoder state: truncated" ); }
  uint8_t u8() { need( 1 ); return p[at++]; }
  uint16_t u16() { need( 2 ); const uint16_t v = static_cast<uint16_t>( p[at] | ( p[at + 1] << 8 ) ); at += 2; return v; }
  uint32_t u32() { need( 4 ); const uint32_t v = p[at] | ( p[at + 1] << 8 ) | ( p[at + 2] << 16 ) | ( static_cast<uint32_t>( p[at + 3] ) << 24 ); at += 4; return v; }
  void bytes( void * dst, size_t k ) { need( k ); std::memcpy( dst, p + at, k ); at += k; }
  void expect( uint8_t tag, const char * what ) { if ( u8() != tag ) throw ParseError( AA_ERR_INVALID, std::string( "invalid decoder state: expected " ) + what ); }
};
}

void Parser::serialize_reference( std::vector<uint8_t> & o ) const
{
  o.push_back( T_DECODER_STATE );
  const size_t len_at = o.size();
  put32( o, 0 );
  const size_t body = o.size();
  put16( o, width_ ); put16( o, height_ );
  o.push_back( T_PROB_TABLE ); put32( o, 1101 );
  const uint8_t * t = &probs_.coeff[0][0][0][0];
  o.insert( o.end(), t, t + 1056 ); o.insert( o.end(), probs_.y_mode, probs_.y_mode + 4 );
  o.insert( o.end(), probs_.uv_mode, probs_.uv_mode + 3 ); o.insert( o.end(), &probs_.mv[0][0], &probs_.mv[0][0] + 38 );
  if ( seg_.enabled ) {
    o.push_back( T_OPT_FULL );
    o.push_back( seg_.absolute ? T_SEGM_ABS : T_SEGM_REL );
    // The reference sizes its segment map by the frame's PIXEL dimensions (DecoderState hands width/height to
    // Segmentation, decoder.cc:238, decoder_state.hh:170-176); only the top-left mb_width x mb_height corner is ever
    // touched, the rest keeps the initial 3.  The wire format carries the whole thing.
    put32( o, 4u + 4u + 4u + static_cast<uint32_t>( width_ ) * height_ );
    put16( o, width_ ); put16( o, height_ );
    for ( int i = 0; i < 4; i++ ) o.push_back( static_cast<uint8_t>( seg_.quant[i] ) );
    for ( int i = 0; i < 4; i++ ) o.push_back( static_cast<uint8_t>( seg_.lf[i] ) );
    for ( unsigned r = 0; r < height_; r++ )
      for ( unsigned c = 0; c < width_; c++ ) o.push_back( ( r < mbh_ && c < mbw_ ) ? seg_.map[static_cast<size_t>( r ) * mbw_ + c] : 3 );
  } else o.push_back( T_OPT_EMPTY );
  if ( fadj_.enabled ) {
    o.push_back( T_OPT_FULL );
    o.push_back( T_FILT_ADJ ); put32( o, 8 );
    for ( int i = 0; i < 4; i++ ) o.push_back( static_cast<uint8_t>( fadj_.ref[i] ) );
    for ( int i = 0; i < 4; i++ ) o.push_back( static_cast<uint8_t>( fadj_.mode[i] ) );
  } else o.push_back( T_OPT_EMPTY );
  // the reference's length field counts 4 (dims) + each part's len+5 + the two option tags (decoder.cc:287-311)
  const uint32_t len = static_cast<uint32_t>( o.size() - body );
  for ( int i = 0; i < 4; i++ ) o[len_at + i] = static_cast<uint8_t>( len >> ( 8 * i ) );
}

size_t Parser::deserialize_reference( const uint8_t * in, size_t size )
{
  Reader r { in, size };
  r.expect( T_DECODER_STATE, "DECODER_STATE" );
  (void) r.u32();
  const uint16_t w = r.u16(), h = r.u16();
  if ( w != width_ || h != height_ ) throw ParseError( AA_ERR_INVALID, "invalid decoder state: frame size differs from this decoder's" );
  r.expect( T_PROB_TABLE, "PROB_TABLE" );
  if ( r.u32() != 1101 ) throw ParseError( AA_ERR_INVALID, "invalid decoder state: probability table length" );
  ProbTables pt;
  r.bytes( pt.coeff, 1056 ); r.bytes( pt.y_mode, 4 ); r.bytes( pt.uv_mode, 3 ); r.bytes( pt.mv, 38 );
  SegmentationState sg; sg.map.assign( static_cast<size_t>( mbw_ ) * mbh_, 3 );
  uint8_t opt = r.u8();
  if ( opt == T_OPT_FULL ) {
    const uint8_t tag = r.u8();
    if ( tag != T_SEGM_ABS && tag != T_SEGM_REL ) throw ParseError( AA_ERR_INVALID, "invalid decoder state: expected SEGM_ABS/SEGM_REL" );
    sg.enabled = true; sg.absolute = tag == T_SEGM_ABS;
    const uint32_t len = r.u32();
    const uint16_t mw = r.u16(), mh = r.u16();        // pixel dimensions, see serialize_reference
    if ( mw != width_ || mh != height_ || len != 12u + static_cast<uint32_t>( mw ) * mh ) throw ParseError( AA_ERR_INVALID, "invalid decoder state: segmentation map size" );
    r.bytes( sg.quant, 4 ); r.bytes( sg.lf, 4 );
    r.need( static_cast<size_t>( mw ) * mh );
    for ( unsigned row = 0; row < mbh_; row++ ) std::memcpy( &sg.map[static_cast<size_t>( row ) * mbw_], r.p + r.at + static_cast<size_t>( row ) * mw, mbw_ );
    r.at += static_cast<size_t>( mw ) * mh;
  } else if ( opt != T_OPT_EMPTY ) throw ParseError( AA_ERR_INVALID, "invalid decoder state: option tag" );
  FilterAdjustState fa;
  opt = r.u8();
  if ( opt == T_OPT_FULL ) {
    r.expect( T_FILT_ADJ, "FILT_ADJ" );
    if ( r.u32() != 8 ) throw ParseError( AA_ERR_INVALID, "invalid decoder state: filter adjustment length" );
    fa.enabled = true; r.bytes( fa.ref, 4 ); r.bytes( fa.mode, 4 );
  } else if ( opt != T_OPT_EMPTY ) throw ParseError( AA_ERR_INVALID, "invalid decoder state: option tag" );
  probs_ = pt; seg_ = sg; fadj_ = fa;       // nothing is changed unless the whole blob parsed
  return r.at;
}

// UncompressedChunk::UncompressedChunk (uncompressed_chunk.cc:34-130).  With accept_partial (error concealment) two things that
// otherwise reject the frame are survived instead:
//   the first partition reaches to or past the end of the frame -> CORRUPTED_FIRST_PARTITION: it is what is left of the frame,
//     there is no DCT data (:82-95);
//   the frame is too short for the fields being read (the reference's Chunk throws out_of_range) -> CORRUPTED_FRAME: an inter
//     frame, version 0, with an EMPTY first partition and no DCT data (:116-127); `show` is whatever byte 0 said, if there is one.
FrameTag parse_frame_tag( const uint8_t * data, size_t size, uint16_t width, uint16_t height, bool accept_partial )
{
  FrameTag t;
  t.rest_at = size;                            // where the DCT partitions start (size: there are none)
  bool corrupted_frame = false;
  if ( size >= 1 ) {
    t.key = !( data[0] & 1 ); t.show = ( data[0] >> 4 ) & 1;
    const unsigned version = ( data[0] >> 1 ) & 7;
    if ( version == 4 || version == 6 ) t.experimental = true;
    else if ( version != 0 ) throw ParseError( AA_ERR_UNSUPPORTED, "unsupported bitstream: VP8 version of " + std::to_string( version ) );
  }
  if ( size < 3 ) {
    if ( !accept_partial ) throw ParseError( AA_ERR_INVALID, "invalid bitstream: VP8 frame truncated" );
    corrupted_frame = true;
  } else {
    const uint32_t tag = data[0] | ( data[1] << 8 ) | ( static_cast<uint32_t>( data[2] ) << 16 );
    t.first_len = ( tag >> 5 ) & 0x7FFFF;
    t.first_off = t.key ? 10 : 3;
    if ( size <= static_cast<size_t>( t.first_off ) + t.first_len ) {
      if ( !accept_partial ) throw ParseError( AA_ERR_INVALID, "invalid bitstream: invalid VP8 first partition length" );
      if ( size < t.first_off ) corrupted_frame = true;          // (frame( offset, size - offset ) is out of range)
      else { t.first_len = static_cast<uint32_t>( size - t.first_off ); t.corruption = 2; }
    } else t.rest_at = static_cast<size_t>( t.first_off ) + t.first_len;
  }
  if ( corrupted_frame ) { t.key = false; t.experimental = false; t.first_off = 0; t.first_len = 0; t.rest_at = size; t.corruption = 3; }
  if ( t.key ) {
    if ( data[3] != 0x9d || data[4] != 0x01 || data[5] != 0x2a )
      throw ParseError( AA_ERR_INVALID, "invalid bitstream: did not find key-frame start code" );
    const unsigned fw = ( data[6] | ( data[7] << 8 ) ) & 0x3FFF, hscale = data[7] >> 6;
    const unsigned fh = ( data[8] | ( data[9] << 8 ) ) & 0x3FFF, vscale = data[9] >> 6;
    if ( fw != width || fh != height || hscale || vscale )
      throw ParseError( AA_ERR_UNSUPPORTED, "unsupported bitstream: VP8 upscaling not supported" );
  }
  return t;
}

void Parser::parse_header( const uint8_t * data, size_t size, aa_frame_header & hdr, FrameParams & fp )
{
  BoolReader bd;
  parse_header_impl( data, size, hdr, fp, bd );
}

void Parser::parse_header_impl( const uint8_t * data, size_t size, aa_frame_header & hdr, FrameParams & fp, BoolReader & bd )
{
  // ---- frame tag + partition split: uncompressed_chunk.cc:34-130 ----
  const FrameTag t = parse_frame_tag( data, size, width_, height_, conceal_ );
  const bool key = t.key, show = t.show;
  if ( !key && t.experimental ) throw ParseError( AA_ERR_UNSUPPORTED, "unsupported bitstream: experimental" );    // decoder.cc:131-133
  if ( key && t.experimental ) throw ParseError( AA_ERR_INVALID, "invalid bitstream: experimental key frame" );   // decoder_state.hh:81-83
  const uint32_t first_off = t.first_off, first_len = t.first_len;
  const uint8_t * rest = data + t.rest_at;
  const size_t rest_len = size - t.rest_at;

  bd.reset( data + first_off, first_len );

  // ---- frame header: frame_header.hh:194-295 ----
  std::memset( &hdr, 0, sizeof hdr );
  hdr.key_frame = key; hdr.show_frame = show;
  hdr.mb_width = static_cast<uint16_t>( mbw_ ); hdr.mb_height = static_cast<uint16_t>( mbh_ );
  hdr.width = width_; hdr.height = height_;
  hdr.num_macroblocks = mbw_ * mbh_;
  hdr.compressed_size = static_cast<uint32_t>( size );

  bool color_space = false, clamping_type = false;
  if ( key ) { color_space = bd.flag(); clamping_type = bd.flag(); }

  // Flagged<UpdateSegmentation>
  const bool seg_enabled = bd.flag();
  bool seg_update_map = false, seg_update_data = false, seg_abs = false;
  int seg_quant[4] = { 0, 0, 0, 0 }, seg_lf[4] = { 0, 0, 0, 0 };
  uint8_t seg_tree_probs[3] = { 255, 255, 255 };
  if ( seg_enabled ) {
    seg_update_map = bd.flag();
    seg_update_data = bd.flag();
    if ( seg_update_data ) {
      seg_abs = bd.flag();
      for ( int i = 0; i < 4; i++ ) seg_quant[i] = bd.flag() ? bd.signed_literal( 7 ) : 0;
      for ( int i = 0; i < 4; i++ ) seg_lf[i] = bd.flag() ? bd.signed_literal( 6 ) : 0;
    }
    if ( seg_update_map ) for ( int i = 0; i < 3; i++ ) seg_tree_probs[i] = bd.flag() ? static_cast<uint8_t>( bd.literal( 8 ) ) : 255;
  }
  const bool filter_type = bd.flag();
  hdr.loop_filter_level = static_cast<uint8_t>( bd.literal( 6 ) );
  hdr.sharpness_level = static_cast<uint8_t>( bd.literal( 3 ) );
  // Flagged<Flagged<ModeRefLFDeltaUpdate>>
  const bool lf_adj_enabled = bd.flag();
  bool lf_delta_update = false;
  int ref_delta[4] = { 0, 0, 0, 0 }, mode_delta[4] = { 0, 0, 0, 0 };
  if ( lf_adj_enabled ) {
    lf_delta_update = bd.flag();
    if ( lf_delta_update ) {
      for ( int i = 0; i < 4; i++ ) ref_delta[i] = bd.flag() ? bd.signed_literal( 6 ) : 0;
      for ( int i = 0; i < 4; i++ ) mode_delta[i] = bd.flag() ? bd.signed_literal( 6 ) : 0;
    }
  }
  const int log2_parts = bd.literal( 2 );
  hdr.num_dct_partitions = static_cast<uint8_t>( 1 << log2_parts );
  const int y_ac_qi = bd.literal( 7 );
  hdr.q_index = static_cast<uint8_t>( y_ac_qi );
  int qdelta[5];   // y_dc, y2_dc, y2_ac, uv_dc, uv_ac (bitstream order, frame_header.hh:40-44)
  for ( int i = 0; i < 5; i++ ) qdelta[i] = bd.flag() ? bd.signed_literal( 4 ) : 0;

  bool refresh_entropy;
  bool sign_bias_golden = false, sign_bias_alt = false;
  if ( key ) {
    refresh_entropy = bd.flag();
    hdr.refresh_last = hdr.refresh_golden = hdr.refresh_alternate = 1;
  } else {
    hdr.refresh_golden = static_cast<uint8_t>( bd.flag() );
    hdr.refresh_alternate = static_cast<uint8_t>( bd.flag() );
    hdr.copy_buffer_to_golden = hdr.refresh_golden ? 0 : static_cast<uint8_t>( bd.literal( 2 ) );
    hdr.copy_buffer_to_alternate = hdr.refresh_alternate ? 0 : static_cast<uint8_t>( bd.literal( 2 ) );
    sign_bias_golden = bd.flag(); sign_bias_alt = bd.flag();
    refresh_entropy = bd.flag();
    hdr.refresh_last = static_cast<uint8_t>( bd.flag() );
  }
  hdr.sign_bias_golden = sign_bias_golden; hdr.sign_bias_alternate = sign_bias_alt;

  // ---- state transition: decoder_state.hh:72-167 ----
  ProbTables ft;            // this frame's tables
  if ( key ) ft.set_defaults();   // key frames reset the persistent state (decoder.cc:234-240), applied below once the header is valid
  else ft = probs_;
  for ( int i = 0; i < 4; i++ ) for ( int j = 0; j < 8; j++ ) for ( int k = 0; k < 3; k++ ) for ( int l = 0; l < 11; l++ )
    if ( bd.get( k_coeff_update_probs[( ( i * 8 + j ) * 3 + k ) * 11 + l] ) ) ft.coeff[i][j][k][l] = static_cast<uint8_t>( bd.literal( 8 ) );
  const bool skip_enabled = bd.flag();
  const int prob_skip = skip_enabled ? bd.literal( 8 ) : 0;
  int prob_inter = 0, prob_last = 0, prob_golden = 0;
  if ( !key ) {
    prob_inter = bd.literal( 8 ); prob_last = bd.literal( 8 ); prob_golden = bd.literal( 8 );
    if ( bd.flag() ) for ( int i = 0; i < 4; i++ ) ft.y_mode[i] = static_cast<uint8_t>( bd.literal( 8 ) );
    if ( bd.flag() ) for ( int i = 0; i < 3; i++ ) ft.uv_mode[i] = static_cast<uint8_t>( bd.literal( 8 ) );
    for ( int i = 0; i < 2; i++ ) for ( int j = 0; j < 19; j++ )
      if ( bd.get( k_mv_update_probs[i * 19 + j] ) ) { const int x = bd.literal( 7 ); ft.mv[i][j] = static_cast<uint8_t>( x ? x << 1 : 1 ); }
  }
  if ( color_space || clamping_type ) throw ParseError( AA_ERR_UNSUPPORTED, "unsupported bitstream: VP8 color_space and clamping_type bits" );
  if ( filter_type ) throw ParseError( AA_ERR_UNSUPPORTED, "unsupported bitstream: VP8 'simple' in-loop deblocking filter" );

  bool seg_reset = false;       // the persistent segment map restarts at all-3 with this frame
  if ( key ) {
    probs_.set_defaults();
    seg_.enabled = seg_enabled; seg_.absolute = false;
    std::memset( seg_.quant, 0, 4 ); std::memset( seg_.lf, 0, 4 );
    if ( seg_enabled ) { std::memset( seg_.map.data(), 3, seg_.map.size() ); seg_reset = true; }    // Segmentation ctor: map(width, height, 3)
    fadj_.enabled = lf_adj_enabled; std::memset( fadj_.ref, 0, 4 ); std::memset( fadj_.mode, 0, 4 );
  } else {
    if ( lf_adj_enabled ) { if ( !fadj_.enabled ) { fadj_.enabled = true; std::memset( fadj_.ref, 0, 4 ); std::memset( fadj_.mode, 0, 4 ); } }
    else fadj_.enabled = false;
    if ( seg_enabled ) {
      if ( !seg_.enabled ) {
        seg_.enabled = true; seg_.absolute = false; std::memset( seg_.quant, 0, 4 ); std::memset( seg_.lf, 0, 4 );
        std::memset( seg_.map.data(), 3, seg_.map.size() ); seg_reset = true;
      }
    } else seg_.enabled = false;
  }
  if ( refresh_entropy ) probs_ = ft;
  if ( lf_adj_enabled && lf_delta_update )
    for ( int i = 0; i < 4; i++ ) { fadj_.ref[i] = static_cast<int8_t>( ref_delta[i] ); fadj_.mode[i] = static_cast<int8_t>( mode_delta[i] ); }
  if ( seg_enabled && seg_update_data ) {
    seg_.absolute = seg_abs;
    for ( int i = 0; i < 4; i++ ) { seg_.quant[i] = static_cast<int8_t>( seg_quant[i] ); seg_.lf[i] = static_cast<int8_t>( seg_lf[i] ); }
  }
  hdr.segmentation_enabled = seg_.enabled; hdr.filter_adjustments_enabled = fadj_.enabled;

  // ---- per-frame constants for the device ----
  for ( int s = 0; s < 4; s++ ) {
    int qi = y_ac_qi;
    if ( seg_.enabled ) qi = static_cast<uint8_t>( seg_.quant[s] + ( seg_.absolute ? 0 : y_ac_qi ) );   // Q2: wraps as uint8 before clamp
    const Quantizer q = make_quantizer( qi, qdelta );
    std::memcpy( hdr.quant[s], q.f, sizeof q.f );
  }
  std::memset( &fp, 0, sizeof fp );
  for ( int s = 0; s < 4; s++ )
    fp.seg_level[s] = static_cast<int16_t>( seg_.enabled ? seg_.lf[s] + ( seg_.absolute ? 0 : hdr.loop_filter_level ) : hdr.loop_filter_level );   // Q3: unclamped

  // ---- DCT partitions: uncompressed_chunk.cc:132-155 ----
  const int nparts = hdr.num_dct_partitions;
  {
    if ( rest_len < static_cast<size_t>( 3 * ( nparts - 1 ) ) ) throw ParseError( AA_ERR_OUT_OF_RANGE, "attempted to read past end of chunk" );
    const uint8_t * p = rest + 3 * ( nparts - 1 );
    size_t left = rest_len - 3 * ( nparts - 1 );
    for ( int i = 0; i < nparts; i++ ) {
      size_t len = left;
      if ( i < nparts - 1 ) {
        len = rest[3 * i] | ( rest[3 * i + 1] << 8 ) | ( static_cast<size_t>( rest[3 * i + 2] ) << 16 );
        if ( len > left ) throw ParseError( AA_ERR_OUT_OF_RANGE, "attempted to read past end of chunk" );
      }
      fp.part_off[i] = static_cast<uint32_t>( p - data ); fp.part_size[i] = static_cast<uint32_t>( len );
      p += len; left -= len;
    }
  }

  // ---- what the macroblock loop (host: Parser::parse below; device: parse_kernels.hip) needs ----
  fp.first_off = first_off; fp.first_size = first_len;
  const BoolState bs = bd.state();
  fp.bd_bitpos = bs.bitpos; fp.bd_range = bs.range; fp.bd_active = bs.active;
  fp.key = key; fp.nparts = static_cast<uint8_t>( nparts );
  fp.mbw = static_cast<uint16_t>( mbw_ ); fp.mbh = static_cast<uint16_t>( mbh_ );
  fp.seg_enabled = seg_enabled; fp.seg_update_map = seg_update_map;
  std::memcpy( fp.seg_tree_probs, seg_tree_probs, 3 );
  fp.skip_enabled = skip_enabled; fp.prob_skip = static_cast<uint8_t>( prob_skip );
  fp.prob_inter = static_cast<uint8_t>( prob_inter ); fp.prob_last = static_cast<uint8_t>( prob_last ); fp.prob_golden = static_cast<uint8_t>( prob_golden );
  fp.sign_bias_golden = sign_bias_golden; fp.sign_bias_alt = sign_bias_alt;
  fp.loop_filter_level = hdr.loop_filter_level; fp.fadj_enabled = fadj_.enabled;
  std::memcpy( fp.fadj_ref, fadj_.ref, 4 ); std::memcpy( fp.fadj_mode, fadj_.mode, 4 );
  std::memcpy( fp.y_mode_probs, ft.y_mode, 4 ); std::memcpy( fp.uv_mode_probs, ft.uv_mode, 3 ); std::memcpy( fp.mv_probs, ft.mv, 38 );
  std::memcpy( fp.coeff_probs, ft.coeff, sizeof ft.coeff );
  seg_map_reset_ = seg_reset;
}

// Everything of a frame behind its header: macroblock headers (first partition, `bd` continuing behind the frame header) and
// tokens (DCT partitions), Frame::parse_macroblock_headers / parse_tokens (frame.cc:95-137).  Needs nothing of the stream's state
// but what the header pre-pass put into `fp` -- and, for streams that use segmentation, the persistent segment map (`segmap`,
// else null).  `above_nz`: scratch, 9 bytes per macroblock column.
template <class BD>
static void parse_body( BD & bd, const uint8_t * data, const FrameParams & fp, aa_mb_info * mbs, int16_t * coeff_out, uint8_t * segmap, uint8_t * above_nz,
                        uint32_t & coeff_blocks_out, uint32_t & intra_mbs_out )
{
  const int nparts = fp.nparts;
  BoolReader parts[8];
  for ( int i = 0; i < nparts; i++ ) parts[i].reset( data + fp.part_off[i], fp.part_size[i] );

  // ---- macroblock headers + tokens ----
  const unsigned mbw = fp.mbw, mbh = fp.mbh;
  std::memset( above_nz, 0, size_t( mbw ) * 9 );
  uint32_t coeff_blocks = 0, intra_mbs = 0;

  for ( unsigned row = 0; row < mbh; row++ ) {
    uint8_t left_nz[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    BoolReader & tok = parts[row % nparts];
    for ( unsigned col = 0; col < mbw; col++ ) {
      const unsigned mi = row * mbw + col;
      aa_mb_info & mb = mbs[mi];
      const uint8_t flags = parse_mb_header( bd, fp, kHeaderTables, mbs, mi, col, row, segmap );
      const bool skip = flags & AA_MB_SKIP, has_y2 = flags & AA_MB_HAS_Y2;
      if ( !( flags & AA_MB_INTER ) ) intra_mbs++;

      // ---- tokens: Macroblock::parse_tokens (macroblock.cc:475-502); storage order = parse order: Y2, Y0..15, U, V ----
      uint8_t * anz = &above_nz[static_cast<size_t>( col ) * 9];
      mb.coeff_index = coeff_blocks;
      bool any = false;
      if ( skip ) {
        std::memset( anz, 0, 8 ); std::memset( left_nz, 0, 8 );
        if ( has_y2 ) { anz[8] = 0; left_nz[8] = 0; }   // a non-coded Y2 leaves the chain untouched (frame.cc:255-269)
      } else {
        uint32_t mask = 0;
        if ( has_y2 ) {                        // parsed first and stored first
          int16_t * slot = coeff_out + static_cast<size_t>( coeff_blocks ) * 16;
          std::memset( slot, 0, 32 );
          const bool nz = parse_block( tok, fp.coeff_probs[Y2], 0, anz[8] + left_nz[8], slot );
          anz[8] = left_nz[8] = nz;
          if ( nz ) { mask |= 1u << 24; coeff_blocks++; }
        }
        const int ytype = has_y2 ? Y_AFTER_Y2 : Y_WITHOUT_Y2;
        const int yfirst = has_y2 ? 1 : 0;
        for ( int b = 0; b < 16; b++ ) {
          int16_t * slot = coeff_out + static_cast<size_t>( coeff_blocks ) * 16;
          std::memset( slot, 0, 32 );
          const int bx = b & 3, by = b >> 2;
          const bool nz = parse_block( tok, fp.coeff_probs[ytype], yfirst, anz[bx] + left_nz[by], slot );
          anz[bx] = left_nz[by] = nz;
          if ( nz ) { mask |= 1u << b; coeff_blocks++; }
        }
        for ( int pl = 0; pl < 2; pl++ ) for ( int b = 0; b < 4; b++ ) {
          int16_t * slot = coeff_out + static_cast<size_t>( coeff_blocks ) * 16;
          std::memset( slot, 0, 32 );
          uint8_t & a = anz[4 + pl * 2 + ( b & 1 )]; uint8_t & l = left_nz[4 + pl * 2 + ( b >> 1 )];
          const bool nz = parse_block( tok, fp.coeff_probs[UV], 0, a + l, slot );
          a = l = nz;
          if ( nz ) { mask |= 1u << ( 16 + pl * 4 + b ); coeff_blocks++; }
        }
        mb.nz_mask = mask;
        any = mask != 0;
      }
      if ( any ) mb.flags |= AA_MB_HAS_NONZERO;
      if ( has_y2 && !any ) mb.flags |= AA_MB_LF_SKIP_INNER;
    }
  }
  coeff_blocks_out = coeff_blocks; intra_mbs_out = intra_mbs;
}

void Parser::parse( const uint8_t * data, size_t size, aa_frame_header & hdr, aa_mb_info * mbs, int16_t * coeff_out )
{
  FrameParams fp;
  BoolReader bd;                 // the first partition's decoder continues right behind the frame header
  parse_header_impl( data, size, hdr, fp, bd );
  uint32_t coeff_blocks = 0, intra_mbs = 0;
  parse_body( bd, data, fp, mbs, coeff_out, seg_.enabled ? seg_.map.data() : nullptr, above_nz_.data(), coeff_blocks, intra_mbs );
  hdr.num_coeff_blocks = coeff_blocks;
  hdr.num_intra_mbs = intra_mbs;
  hdr.has_intra_mb = intra_mbs != 0;
}

// ... from what the header pre-pass (Parser::parse_header) left in `fp`, the way a GPU lane takes a frame over: the first
// partition's decoder resumed from the exported state.  For frames of streams WITHOUT segmentation (the persistent map is the
// one piece of macroblock data that outlives a frame); any thread, no Parser.
void parse_frame_body( const uint8_t * data, const FrameParams & fp, aa_mb_info * mbs, int16_t * coeff_out, uint8_t * above_nz, uint32_t * coeff_blocks, uint32_t * intra_mbs )
{
  BoolReader32 bd;
  BoolState st; st.bitpos = fp.bd_bitpos; st.range = fp.bd_range; st.active = fp.bd_active;
  bd.resume( data + fp.first_off, fp.first_size, st );
  parse_body( bd, data, fp, mbs, coeff_out, nullptr, above_nz, *coeff_blocks, *intra_mbs );
}

} // namespace aa
