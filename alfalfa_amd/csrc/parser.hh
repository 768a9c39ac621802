// Host-side VP8 frame parser: compressed frame -> aa_frame_header + aa_mb_info[] + compact coefficient blocks.
//
// Implements, from the algorithm, what the reference does in
//   UncompressedChunk (uncompressed_chunk.cc:34-155), DecoderState::parse_and_apply<F> (decoder_state.hh:72-167),
//   Frame::parse_macroblock_headers / parse_tokens (frame.cc:95-137), Macroblock ctor + decode_prediction_modes
//   (macroblock.cc:43-111,342-456), Block::parse_tokens (tokens.cc:50-135), plus the per-frame constants the device
//   needs (quantisers frame.cc:185-206 / quantization.cc:83-93, per-MB loop-filter level frame.cc:144-166 /
//   macroblock.cc:611-623 / loopfilter.cc:59-79).
// It does NOT build the reference's object graph (TwoD<Macroblock>, Optional contexts): records go straight into
// caller-provided (normally pinned) arrays in the layout the kernels read.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/alfalfa_amd.h"
#include "parse_common.hh"

namespace aa {

struct ProbTables {            // ProbabilityTables, decoder.hh:57-92
  uint8_t coeff[4][8][3][11];
  uint8_t y_mode[4];
  uint8_t uv_mode[3];
  uint8_t mv[2][19];
  void set_defaults();
};

struct SegmentationState {     // Optional<Segmentation>, decoder.hh:151-188
  bool enabled = false;
  bool absolute = false;
  int8_t quant[4] = { 0, 0, 0, 0 };
  int8_t lf[4] = { 0, 0, 0, 0 };
  std::vector<uint8_t> map;    // mb_width * mb_height
};

struct FilterAdjustState {     // Optional<FilterAdjustments>, decoder.hh:94-121
  bool enabled = false;
  int8_t ref[4] = { 0, 0, 0, 0 };
  int8_t mode[4] = { 0, 0, 0, 0 };
};

class ParseError
{
public:
  aa_status code;
  std::string message;
  ParseError( aa_status c, std::string m ) : code( c ), message( std::move( m ) ) {}
};

// What the first bytes of a frame say (UncompressedChunk, uncompressed_chunk.cc:34-130)
struct FrameTag {
  bool key = false, show = false, experimental = false;
  int corruption = 0;                 // CorruptionLevel (uncompressed_chunk.hh:40-46): 0 none, 2 CORRUPTED_FIRST_PARTITION, 3 CORRUPTED_FRAME
  uint32_t first_off = 3, first_len = 0;
  size_t rest_at = 0;                 // where the DCT partitions start
};
FrameTag parse_frame_tag( const uint8_t * data, size_t size, uint16_t width, uint16_t height, bool accept_partial );   // throws ParseError

class Parser
{
public:
  Parser( uint16_t width, uint16_t height );

  // Throws ParseError.  mb_out: mb_width*mb_height records.  coeff_out: worst case 25*16 int16 per MB.
  void parse( const uint8_t * data, size_t size, aa_frame_header & hdr, aa_mb_info * mb_out, int16_t * coeff_out );

  // Header pre-pass only: frame tag, partition split, frame header, state transition (everything of `parse` that is serial
  // across the frames of a stream), no macroblock data.  `fp` receives what the macroblock loop needs, including the boolean
  // decoder's state at the first macroblock header; the device parser (parse_kernels.hip) takes it from there.  The
  // persistent segment MAP is not maintained by this call (it is macroblock data): see segment_map_reset().
  void parse_header( const uint8_t * data, size_t size, aa_frame_header & hdr, FrameParams & fp );
  // whether the last header restarted the persistent segment map at all-3 (key frame / segmentation switched on)
  bool segment_map_reset() const { return seg_map_reset_; }
  std::vector<uint8_t> & segment_map() { return seg_.map; }

  // Decoder::set_error_concealment (decoder.hh:298): frames that end early are ACCEPTED instead of rejected
  // (UncompressedChunk's accept_partial, uncompressed_chunk.cc:34-130): a frame cut inside its first partition keeps what is
  // there of it and has no DCT data; a frame too short even for its tag becomes an inter frame of no bytes at all; every
  // decoder reads zeros past the end of what it has.  (That is ALL the reference does: the macroblock-level branches of
  // macroblock.cc:53-70,345-352,386 depend on BoolDecoder::valid(), and decoder_state.hh:79,120 builds the first partition's
  // decoder with complete_chunk = "the frame is corrupted", so valid() never turns false on a corrupted frame.)
  void set_error_concealment( bool on ) { conceal_ = on; }
  bool error_concealment() const { return conceal_; }

  uint16_t width() const { return width_; }
  uint16_t height() const { return height_; }
  unsigned mb_width() const { return mbw_; }
  unsigned mb_height() const { return mbh_; }

  // DecoderState as a flat blob (the hand-off `Decoder( DecoderState, References )` needs, decoder.cc:43-46):
  // "AAST" u16 version, u16 width, u16 height, probs[1101], seg{enabled,absolute,quant[4],lf[4]}, fadj{enabled,ref[4],mode[4]},
  // segmentation map[mb_width*mb_height]
  size_t state_size() const;
  void export_state( uint8_t * out ) const;
  void import_state( const uint8_t * in, size_t size );   // throws ParseError(AA_ERR_ARGUMENT) on a foreign / mismatching blob

  // The same state in the REFERENCE's wire format (DecoderState::serialize / deserialize, decoder.cc:283-330; tags and
  // little-endian integers of enc_state_serializer.hh:43-56): [DECODER_STATE][u32 len][u16 w][u16 h][PROB_TABLE ...]
  // [OPT_FULL SEGM_ABS|SEGM_REL ... | OPT_EMPTY][OPT_FULL FILT_ADJ ... | OPT_EMPTY].
  void serialize_reference( std::vector<uint8_t> & out ) const;
  size_t deserialize_reference( const uint8_t * in, size_t size );   // returns the bytes consumed; throws ParseError(AA_ERR_INVALID)

  const ProbTables & probs() const { return probs_; }
  const SegmentationState & segmentation() const { return seg_; }
  const FilterAdjustState & filter_adjustments() const { return fadj_; }

private:
  void parse_header_impl( const uint8_t * data, size_t size, aa_frame_header & hdr, FrameParams & fp, class BoolReader & bd );
  bool seg_map_reset_ = false;
  bool conceal_ = false;
  uint16_t width_, height_;
  unsigned mbw_, mbh_;
  ProbTables probs_;
  SegmentationState seg_;
  FilterAdjustState fadj_;
  std::vector<uint8_t> above_nz_;   // per MB column: 4 Y, 2 U, 2 V, 1 Y2
};

// A frame's macroblock headers and tokens from the header pre-pass's FrameParams alone (no Parser: any thread) -- what a GPU lane
// does with a ParseJob, on a host core.  Streams without segmentation only (fp.seg_enabled == 0).  above_nz: 9 * fp.mbw bytes of
// scratch; mbs: fp.mbw * fp.mbh records; coeff_out: worst case 25 * 16 int16 per macroblock.
void parse_frame_body( const uint8_t * data, const FrameParams & fp, aa_mb_info * mbs, int16_t * coeff_out, uint8_t * above_nz, uint32_t * coeff_blocks, uint32_t * intra_mbs );

} // namespace aa
