/* alfalfa_amd.h -- C ABI of the MI355X-native VP8 decode hot path (drop-in for excamera/alfalfa src/decoder).
 *
 * The reference has no FFI layer: its seam is the C++ class API Decoder / DecoderState / References /
 * RasterHandle / VP8Raster (decoder.hh:123-300, raster_handle.hh:77-123, vp8_raster.hh:53-316).  This
 * header is the plain-C boundary underneath our C++ mirror of those classes (include/alfalfa_amd/ C++ headers)
 * and underneath any other binding (ctypes in alfalfa_amd/capi.py).  Plain pointers
 * and sizes only; every function returns an aa_status and never throws.  No torch types.
 *
 * Pipeline (reference call stack SURVEY.md 3.1):
 *   host   aa_parser_*      decompress_frame + parse_frame  (uncompressed_chunk.cc:34-155, decoder_state.hh:72-167,
 *                           frame.cc:95-137, macroblock.cc:43-502, tokens.cc:50-135) -> macroblock records + coefficient blocks
 *   device aa_stream_*      decode_frame: Frame::decode + Frame::loopfilter + Frame::copy_to (decoder.cc:101-118,
 *                           frame.cc:139-307) as HIP kernels over device-resident rasters
 *   batch  aa_decode_batch  one pass of the hot path over frame f_i of N independent streams (ExCamera chunks / GOPs)
 *
 * There is NO CPU fallback for the device half: without a HIP device every aa_ctx_* / aa_stream_* call fails
 * with AA_ERR_NO_DEVICE.  The parser half is host code by design (serial BoolDecoder) and runs anywhere.
 */
#ifndef ALFALFA_AMD_H
#define ALFALFA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4 (round 5): aa_ctx_info and aa_kernel_stats grew in round 4 (a caller built against version 3 passes smaller structs: check
 * aa_abi_version() against the header you compiled with before calling aa_ctx_get_info / aa_ctx_kernel_stats), packed coefficient
 * storage became the default, AA_SUBMIT_HOST on a big call no longer waits for the parse (host lanes), hand-overs are refused at the memory limit. */
#define AA_ABI_VERSION 4

typedef enum aa_status {
  AA_OK = 0,
  AA_ERR_INVALID = -1,      /* reference: Invalid("invalid bitstream: ...")       exception.hh:76-82  */
  AA_ERR_UNSUPPORTED = -2,  /* reference: Unsupported("unsupported bitstream: ...") exception.hh:84-90 */
  AA_ERR_LOGIC = -3,        /* reference: LogicError                                exception.hh:92-98  */
  AA_ERR_OUT_OF_RANGE = -4, /* reference: std::out_of_range from Chunk bounds       chunk.hh:54-59     */
  AA_ERR_HIP = -5,          /* HIP runtime failure (message has hipGetErrorString)                      */
  AA_ERR_NO_DEVICE = -6,    /* no HIP device / kernels missing: the device path NEVER falls back to CPU */
  AA_ERR_ARGUMENT = -7,
  AA_ERR_NO_MEMORY = -8     /* HBM: the memory limit of the context leaves no room (the message says what to release); the call can be repeated */
} aa_status;

/* Message of the last failing call on this thread ("" if none). */
const char * aa_last_error( void );
int aa_abi_version( void );
/* The HIP runtime reads GPU_MAX_HW_QUEUES when it starts; this library runs 16 HIP streams side by side (long-lived entropy-decode
 * grids beside short reconstruction kernels) and wants 16 hardware queues (the default is 4).  Call this before the process makes
 * its first HIP call -- the bindings call it before their first aa_ctx_create -- to set the variable if the environment does not
 * (a value that is set stands).  -> 1 if it was set already, 0 if this call set it.  Nothing else of the host process is touched;
 * a context checks what it really got (aa_ctx_info::stream_concurrency). */
int aa_runtime_prepare( void );
/* Cores this process can really use: hardware threads and affinity mask bounded by the cgroup CPU quota (a container may show 256
 * and grant 16).  The host workers of aa_submit_frames are never more than twice this, whatever `threads` says. */
int aa_host_cpus( void );
/* Number of visible HIP devices (0 when there is none; never fails). */
int aa_device_count( void );

/* ------------------------------------------------------------------------------------------------
 * Parsed-frame records (what Appendix B of SURVEY.md says the device needs).  Layout is ABI.
 * ---------------------------------------------------------------------------------------------- */

/* Per-macroblock record, 80 bytes.  Modes use the reference's enums (modemv_data.hh:37-49). */
typedef struct aa_mb_info {
  uint8_t y_mode;        /* mbmode: DC_PRED,V_PRED,H_PRED,TM_PRED,B_PRED,NEARESTMV,NEARMV,ZEROMV,NEWMV,SPLITMV */
  uint8_t uv_mode;       /* DC_PRED..TM_PRED (intra MBs) */
  uint8_t ref_frame;     /* reference_frame: 0 CURRENT (intra), 1 LAST, 2 GOLDEN, 3 ALTREF */
  uint8_t segment_id;    /* 0..3 (persistent map value; 0 when segmentation is off) */
  uint8_t flags;         /* AA_MB_* */
  uint8_t lf_level;      /* final loop-filter level of this MB, 0 = not filtered, else 1..63
                            (segment + ref + mode adjustments applied, loopfilter.cc:43-79, macroblock.cc:611-623) */
  uint8_t split_partition; /* SPLITMV partition id (mv_partitions index) */
  uint8_t reserved;
  uint32_t nz_mask;      /* bit b set: block b has stored coefficients. b: 0..15 Y (raster), 16..19 U, 20..23 V, 24 Y2.
                            Stored blocks follow each other in PARSE order: Y2 (if bit 24), then bits 0..23 ascending */
  uint32_t coeff_index;  /* index (in 16-coefficient blocks) of this MB's first stored block in the frame's coefficient array */
  union {
    uint8_t b_mode[16];  /* B_PRED: bmode of each 4x4 (intra MBs) */
    int16_t mv[16][2];   /* inter MBs: {x, y} of each Y sub-block, quarter-pel (all equal unless SPLITMV) */
  } u;
} aa_mb_info;

#define AA_MB_HAS_NONZERO 1u   /* Macroblock::has_nonzero_: residual path runs (macroblock.cc:531,547,579,593) */
#define AA_MB_HAS_Y2 2u        /* Y2 coded: mode is neither B_PRED nor SPLITMV (block.hh:183-190) */
#define AA_MB_INTER 4u         /* inter_coded() */
#define AA_MB_SKIP 8u          /* mb_skip_coeff */
#define AA_MB_LF_SKIP_INNER 16u /* loop filter skips sub-block edges: Y2 coded && !has_nonzero (macroblock.cc:607) */

/* Per-frame constants. */
typedef struct aa_frame_header {
  uint8_t key_frame, show_frame;
  uint8_t loop_filter_level;   /* header value; 0 => Frame::loopfilter is skipped entirely (frame.cc:143) */
  uint8_t sharpness_level;
  uint8_t num_dct_partitions;
  uint8_t segmentation_enabled, filter_adjustments_enabled;
  uint8_t refresh_last, refresh_golden, refresh_alternate;  /* key frames: all 1 */
  uint8_t copy_buffer_to_golden, copy_buffer_to_alternate;  /* 0 none, 1 last, 2 the other (frame.cc:278-292) */
  uint8_t sign_bias_golden, sign_bias_alternate;
  uint8_t q_index;
  uint8_t has_intra_mb;        /* any intra MB in this frame (always 1 for key frames) */
  uint16_t mb_width, mb_height;
  uint16_t width, height;      /* display size */
  /* dequantisation factors per segment (index 0 used when segmentation is off; all 4 filled):
     {y_dc, y_ac, y2_dc, y2_ac, uv_dc, uv_ac}  quantization.cc:83-93, frame.cc:185-206 */
  uint16_t quant[4][6];
  uint32_t num_macroblocks;
  uint32_t num_coeff_blocks;   /* stored 16-coefficient blocks (32 bytes each) */
  uint32_t num_intra_mbs;
  uint32_t compressed_size;
} aa_frame_header;

/* ------------------------------------------------------------------------------------------------
 * Host parser: the reference's DecoderState + parse_frame (decoder_state.hh:72-167).  Host only.
 * ---------------------------------------------------------------------------------------------- */
typedef struct aa_parser aa_parser;

aa_status aa_parser_create( uint16_t width, uint16_t height, aa_parser ** out );
void aa_parser_destroy( aa_parser * p );

/* Parse one compressed frame (Chunk = {data,size}, chunk.hh:38-73) and apply it to the persistent state.
 * mb_out must hold mb_width*mb_height records; coeff_out must hold 25*16*mb_width*mb_height int16 (worst case).
 * On error the persistent state is left as the reference leaves it (it may already be modified). */
aa_status aa_parser_parse( aa_parser * p, const uint8_t * data, size_t size,
                           aa_frame_header * hdr_out, aa_mb_info * mb_out, int16_t * coeff_out );

/* Decoder::set_error_concealment (decoder.hh:298; FramePlayer::set_error_concealment player.hh:72; salsify-receiver.cc:192): a frame
 * that ends inside its first partition, or before its tag is complete, is ACCEPTED (UncompressedChunk accept_partial,
 * uncompressed_chunk.cc:34-130) -- what is there of the first partition is used, decoders read zeros past the end, a frame
 * without a complete tag becomes an inter frame of no bytes -- instead of AA_ERR_INVALID.  Off by default, as in the reference. */
aa_status aa_parser_set_error_concealment( aa_parser * p, int on );
/* UncompressedChunk( frame, expected_width, expected_height, accept_partial ) (uncompressed_chunk.cc:34-130): what the frame tag
 * says, with the reference's checks and error classes.  corruption_level: 0 NO_CORRUPTION, 2 CORRUPTED_FIRST_PARTITION,
 * 3 CORRUPTED_FRAME (uncompressed_chunk.hh:40-46).  Any out pointer may be NULL. */
aa_status aa_parse_frame_tag( const uint8_t * data, size_t size, uint16_t width, uint16_t height, int accept_partial,
                              int * key_frame, int * show_frame, int * experimental, int * corruption_level );

/* Persistent state (DecoderState, decoder.hh:190-225), flat export for tests / serialisation:
 * probs[1101] = 1056 coefficient, 4 y-mode, 3 uv-mode, 38 mv probabilities. */
aa_status aa_parser_get_probs( const aa_parser * p, uint8_t probs[1101] );
/* segmentation: enabled, absolute, quant[4], lf[4]; map (mb_width*mb_height bytes) may be NULL */
aa_status aa_parser_get_segmentation( const aa_parser * p, int * enabled, int * absolute, int8_t quant[4], int8_t lf[4], uint8_t * map );
aa_status aa_parser_get_filter_adjustments( const aa_parser * p, int * enabled, int8_t ref[4], int8_t mode[4] );

/* DecoderState as one flat blob: the host half of the entry-state hand-off `Decoder( DecoderState, References )`
 * (decoder.cc:43-46; the device half is aa_stream_import_reference).  Size depends only on the frame size. */
size_t aa_parser_state_size( const aa_parser * p );
aa_status aa_parser_export_state( const aa_parser * p, uint8_t * buf, size_t capacity );
aa_status aa_parser_import_state( aa_parser * p, const uint8_t * buf, size_t size );

/* The same state in the REFERENCE's wire format: DecoderState::serialize / DecoderState::deserialize
 * (decoder.cc:283-330; tags and little-endian integers of enc_state_serializer.hh:43-86).  buf == NULL: size query. */
aa_status aa_parser_serialize_state( const aa_parser * p, uint8_t * buf, size_t capacity, size_t * size );
aa_status aa_parser_deserialize_state( aa_parser * p, const uint8_t * buf, size_t size );

/* ------------------------------------------------------------------------------------------------
 * Device context: one per GPU (one process per GPU in multi-GPU runs).
 * ---------------------------------------------------------------------------------------------- */
typedef struct aa_ctx aa_ctx;
typedef struct aa_stream aa_stream;

aa_status aa_ctx_create( int device, aa_ctx ** out );
void aa_ctx_destroy( aa_ctx * ctx );
aa_status aa_ctx_sync( aa_ctx * ctx );               /* waits for copy + compute streams */
/* HBM of the context's device: bytes free / total right now (hipMemGetInfo).  Either pointer may be NULL. */
aa_status aa_ctx_memory( aa_ctx * ctx, size_t * free_bytes, size_t * total_bytes );
/* Launch schedule of the dependency-ordered kernels (intra prediction, loop filter):
 *   AA_SCHEDULE_ROWS (default)  row-pipelined persistent kernels, rows ordered in-launch by ticket + progress words
 *   AA_SCHEDULE_DIAGONAL        one launch per 2:1 anti-diagonal (kernel boundary = synchronisation); for A/B runs
 * Also selectable with the environment variable ALFALFA_AMD_SCHEDULE=rows|diagonal at context creation. */
#define AA_SCHEDULE_ROWS 0
#define AA_SCHEDULE_DIAGONAL 1
aa_status aa_ctx_set_schedule( aa_ctx * ctx, int schedule );
/* The row-pipelined kernels keep a sticky error word (a bounded in-launch wait expired, a wave migrated between XCDs, a
 * ticket queue was not drained): once set, aa_ctx_sync / downloads report AA_ERR_HIP and later launches give up early.
 * After the caller has dealt with it (e.g. switched to AA_SCHEDULE_DIAGONAL), this clears the word. */
aa_status aa_ctx_clear_error( aa_ctx * ctx );
/* HBM the context may take for its pools (frame records, rasters, the coefficient heap of the device parser): by default 7/8 of
 * what was free when it was created; a caller that shares the GPU sets less.  Memory is taken as frames need it, up to this.
 * Round 5: the limit bounds what the context ACCEPTS.  aa_submit_frames stops a thirty-second short of it: a hand-over whose
 * arena would take pool + mapped coefficient heap beyond that first waits for pieces released behind queued kernels, and when
 * none are left is refused with AA_ERR_NO_MEMORY (nothing is appended; repeatable once frames have been reconstructed and
 * released) -- a pipelining caller treats that as "not now".  The coefficient heap, which never unmaps, leaves a sixteenth of
 * the limit to the pool.  Reconstruction (rasters, transient dense blocks) is never refused: it lives on what is kept back, and
 * goes past the limit only if the caller holds more decoded frames at once than that covers. */
aa_status aa_ctx_set_memory_limit( aa_ctx * ctx, size_t bytes );
/* What the context holds right now (the memory budget of a deployment: one context per GPU, one process per GPU). */
typedef struct aa_ctx_info {
  uint64_t memory_limit_bytes;       /* aa_ctx_set_memory_limit */
  uint64_t pool_bytes;               /* HBM taken for frame records, rasters, batch arenas (slabs, recycled) */
  uint64_t pool_free_bytes;          /* ... of which in the free lists right now */
  uint64_t pool_pending_bytes;       /* ... of which released but possibly still read by queued kernels */
  uint64_t heap_mapped_bytes;        /* HBM mapped into the coefficient heap of the device parser */
  uint64_t heap_limit_bytes;         /* its virtual size */
  uint64_t heap_used_bytes;          /* chunks frames hold or are expected to take */
  uint64_t pinned_host_bytes;        /* pinned host memory (batch arenas, staging chunks) */
  uint32_t heap_is_virtual;          /* 1: grown on demand (hipMemMap); 0: one fixed allocation (no virtual memory management) */
  uint32_t token_lanes_per_workgroup, token_workgroups_capacity, token_workgroups_alive;
  uint32_t token_lane_lds_bytes, token_workgroup_lds_bytes;
  uint32_t jobs_waiting;             /* frames in the token workers' queue that no lane has taken */
  uint32_t compute_units;
  int32_t heap_free_chunks;          /* 64-KB chunks in the coefficient pool */
  uint32_t lanes_starved;            /* times a token lane found the pool empty (since the context was created) */
  /* diagnostics, ALFALFA_AMD_TOKEN_PROFILE=1 (else zero): where the token workers' waves spent their time, summed over the waves
   * that have left, in 10-ns ticks: [0] macroblock-boundary passes [1] how many [2] decode steps of waves [3] looking for /
   * starting frames [4] ring top-ups [5] periods (steps + boundary passes) [6] lane-periods that had a frame [7] periods */
  uint64_t token_profile[8];
  uint32_t packed_coefficients;      /* 1: device-parsed frames store packed coefficients (aa_ctx_set_packed_coefficients) */
  uint32_t lane_per_partition;       /* 1: frames with several DCT partitions may get a token lane per partition (aa_ctx_set_lane_per_partition) */
  uint32_t clock_mhz;                /* the device's shader clock (hipDeviceAttributeClockRate) */
  uint32_t host_share_ms;            /* aa_ctx_set_host_share_ms */
  uint32_t host_rate_kb_per_ms;      /* what the host lanes get through, all workers together (KB of compressed data per ms: measured on the frames they have parsed,
                                        parse time only; 0: none yet).  The share of later calls is planned with it: visible cores and usable cores differ under a CPU quota */
  uint32_t reserved1;
  uint32_t stream_concurrency;       /* how many of the context's HIP streams were seen running side by side (probed at the first aa_submit_frames; 0: not yet) */
  uint32_t streams_needed;           /* ... of how many (16): fewer means GPU_MAX_HW_QUEUES was not in effect, see aa_runtime_prepare */
  uint32_t host_waited_parse_ms;     /* aa_decode_batch waiting for the device parser, since the last aa_ctx_kernel_stats reset (= its parse_wait_ms, without its synchronisation) */
  uint32_t host_waited_compute_ms;   /* ... for a raster-binding buffer: the compute stream was 16 calls behind (= bind_wait_ms) */
} aa_ctx_info;
aa_status aa_ctx_get_info( aa_ctx * ctx, aa_ctx_info * out );
/* How the device parser stores a frame's coefficients until the frame is reconstructed.  1 (default): packed -- per macroblock 25
 * mask slots + the non-zero coefficients in parse order (about a third of the memory on video content, and fewer stores for the
 * token lanes); the reconstruction kernels read that form themselves (since round 6: no dense copy is made).
 * 0: dense, 32 bytes per non-zero 4x4 block.  Results are identical.  The choice is per context and can only be made before
 * the context's first aa_submit_frames call (AA_ERR_LOGIC afterwards); the environment variable ALFALFA_AMD_PACKED=0 makes
 * dense the default of every context. */
aa_status aa_ctx_set_packed_coefficients( aa_ctx * ctx, int on );
/* Key frames of big calls on the host's cores.  aa_submit_frames hands a call with many streams to the GPU's token lanes; a KEY
 * frame's chain is the longest there is (seconds on a lane, ~35 ms on a core) and its group cannot be reconstructed before it
 * is parsed, so the streams of such a call whose frames are all key frames go to the context's host lanes instead (see
 * AA_SUBMIT_HOST: the same records, and the call does not wait) -- biggest first, while what the host lanes have been given and
 * not finished stays within `ms` milliseconds of their work at the rate they have really achieved.  Default 80 (environment:
 * ALFALFA_AMD_HOST_SHARE_MS); 0: every frame of a big call goes to the GPU's lanes.  AA_SUBMIT_DEVICE overrides it per call. */
aa_status aa_ctx_set_host_share_ms( aa_ctx * ctx, double ms );
/* One token lane per DCT partition.  A frame with 2, 4 or 8 partitions (frame.cc:119-137: macroblock row r is coded in
 * partition r % P) is then decoded by that many lanes of one wave -- rows handed from lane to lane through the above-row
 * flags in LDS --, whenever the wave that draws it has the lanes idle; its entropy-decode latency falls towards 1 / P of the
 * single-lane figure.  Single-partition frames, records and rasters are unaffected.  Per context, before its first
 * aa_submit_frames call (AA_ERR_LOGIC afterwards); ALFALFA_AMD_LANE_PER_PARTITION=1 makes it the default.  Off by default:
 * simulated on the host lane by lane and wave by wave (tests/test_wave_sim.py) and run on the GPU (tests/test_gpu_lane_per_partition.py). */
aa_status aa_ctx_set_lane_per_partition( aa_ctx * ctx, int on );
/* hipStream_t handles as opaque pointers (to order foreign work, e.g. an RCCL broadcast, against ours) */
void * aa_ctx_compute_stream( aa_ctx * ctx );
void * aa_ctx_copy_stream( aa_ctx * ctx );

/* ------------------------------------------------------------------------------------------------
 * Stream decoder: the reference's Decoder (decoder.hh:244-300) with device-resident References.
 * Frames are appended in bitstream order; rasters live in HBM and are fetched on demand.
 * ---------------------------------------------------------------------------------------------- */
aa_status aa_stream_create( aa_ctx * ctx, uint16_t width, uint16_t height, aa_stream ** out );
void aa_stream_destroy( aa_stream * s );

/* Host half: parse frame into pinned staging (no GPU work). Returns the frame's index in *frame_index. */
aa_status aa_stream_parse( aa_stream * s, const uint8_t * data, size_t size, int * frame_index, aa_frame_header * hdr_out );
/* Append a frame given as RECORDS -- header (quantiser factors, loop-filter level, reference update flags ...), macroblock records
 * with their final loop-filter levels, coefficient blocks in parse order -- instead of as a bitstream: the reference update of
 * Encoder::write_frame (encoder.cc:146-160: frame.decode + frame.loopfilter + copy_to on a Frame the encoder holds) and the replay
 * of xc-enc -r (frontend/xc-enc.cc:286-300) need no serialise -> parse round trip.  The records are what aa_parser_parse /
 * aa_stream_read_records produce.  The stream's DecoderState is NOT advanced (it belongs to whoever made the records); the frame
 * is decoded by aa_decode_batch like any other. */
aa_status aa_stream_append_records( aa_stream * s, const aa_frame_header * hdr, const aa_mb_info * mbs, const int16_t * coeffs, int * frame_index );
/* hipMemcpyAsync (copy stream) of every parsed, not yet uploaded frame's records into HBM. */
aa_status aa_stream_upload( aa_stream * s );
/* Give the pinned host staging of everything uploaded so far back to the system (the device copy is what decode reads).
 * Waits for this context's copy stream.  Later aa_stream_parse calls stage into fresh pinned memory.  Optional: a
 * long-lived bundle decoder that parses ahead of decoding calls it to bound pinned memory. */
aa_status aa_stream_release_staging( aa_stream * s );
/* Device half for frame `frame_index` of each of n streams (all on the same ctx): reconstruct, loop-filter,
 * update references.  Frames of one stream must be submitted in order.  Asynchronous. */
aa_status aa_decode_batch( aa_ctx * ctx, aa_stream * const * streams, int n, const int * frame_index );
/* Convenience = parse + upload + decode_batch(n=1): Decoder::get_frame_output (decoder.cc:125-135). */
aa_status aa_stream_decode( aa_stream * s, const uint8_t * data, size_t size, int * frame_index, int * shown );

/* ---- Device-side entropy decode (reference HOT LOOP #1: Frame::parse_macroblock_headers / parse_tokens,
 * frame.cc:95-137, tokens.cc:50-135, over BoolDecoder bool_decoder.hh:45-120).
 * aa_stream_parse runs the serial BoolDecoder on ONE host core per stream; a GPU consumes the output of ~1000 such cores.
 * aa_submit_frames instead does only the frame-header pre-pass on the host (decoder_state.hh:72-167: the part that is serial
 * across the frames of a stream) and hands the compressed frames themselves to the GPU, where every (stream, frame) is an
 * independent lane: macroblock headers and tokens are parsed in HBM, into the same records aa_stream_parse would have
 * produced.  Frames may be many per stream (in order) and of many streams; streams are processed in parallel by `threads`
 * host workers (0: one per core).  Asynchronous; aa_decode_batch of those frames orders itself behind the parse.
 * frame_index_out[i] (may be NULL) = index of frame i in its stream, -1 if it was not appended.  On a bitstream error the
 * failing frame and the later frames of ITS stream in this call are not appended, everything else is; the first error is
 * returned (same classes as aa_stream_parse). */
typedef struct aa_frame_in { aa_stream * stream; const uint8_t * data; size_t size; } aa_frame_in;
aa_status aa_submit_frames( aa_ctx * ctx, const aa_frame_in * frames, int n, int * frame_index_out, int threads );
/* Two-phase form, for callers that pipeline many batches and are bound by HBM: with AA_SUBMIT_DEFER_TOKENS the call does the
 * header pre-pass, the upload and the macroblock-header kernel (first partition) and allocates only the macroblock records;
 * the token kernel (DCT partitions) and the coefficient blocks -- 9/10 of a frame's records, worst-case sized -- wait for
 * aa_launch_tokens, which takes the oldest `max_batches` deferred batches (<= 0: all).  Anything that needs a deferred
 * frame's records (aa_decode_batch, aa_stream_frame_header, aa_stream_read_records) launches its batch's tokens itself. */
#define AA_SUBMIT_DEFER_TOKENS 1u
/* Routing.  By default a call whose streams are fewer than the host workers it may use (and at most 24) is parsed on the HOST --
 * one worker per stream, Parser::parse, records uploaded: the same records, sooner, because a frame on a GPU lane is a chain of
 * seconds and one core is worth ~75 lanes (an 8-chunk bundle: 4x the rate of the GPU parser) -- and everything larger on the GPU.
 * AA_SUBMIT_DEVICE / AA_SUBMIT_HOST force one or the other (so does the environment variable ALFALFA_AMD_ROUTE=device|host).
 * AA_SUBMIT_HOST on a call with MORE streams than that (frames that are needed at once: the key frames of the groups of pictures a
 * pipeline starts with -- 35 ms on a core, 2 s as a chain on a GPU lane) hands the frames to the context's HOST LANES: they take the
 * device route's header pre-pass and arena (frame indices at once; later frames of the same streams can be submitted immediately, on
 * any route), worker threads of the context -- as many as aa_host_cpus() -- parse macroblock headers and tokens the way a GPU lane
 * does and finish each frame with the same completion word.  THE CALL DOES NOT WAIT FOR THEM; aa_decode_batch,
 * aa_stream_frame_header, aa_stream_read_records and the release calls do, frame by frame.  Frames of a stream that uses
 * segmentation (its persistent map lives on the device) take the GPU's lanes all the same. */
#define AA_SUBMIT_DEVICE 2u
#define AA_SUBMIT_HOST 4u
aa_status aa_submit_frames_ex( aa_ctx * ctx, const aa_frame_in * frames, int n, int * frame_index_out, int threads, unsigned flags );
aa_status aa_launch_tokens( aa_ctx * ctx, int max_batches, int * launched_out );
/* Header of an appended frame.  For device-parsed frames the counts (num_coeff_blocks, num_intra_mbs, has_intra_mb) are known
 * only once the parse has run: this call waits for it. */
aa_status aa_stream_frame_header( aa_stream * s, int frame_index, aa_frame_header * out );
/* Test / debug view: a frame's parsed records as they sit in HBM (host- or device-parsed), copied back.  mb_out:
 * mb_width*mb_height records; coeff_out: up to coeff_capacity_blocks blocks of 16 (either may be NULL). */
aa_status aa_stream_read_records( aa_stream * s, int frame_index, aa_mb_info * mb_out, int16_t * coeff_out, size_t coeff_capacity_blocks );

/* Decoder::set_error_concealment for a stream decoder: applies to aa_stream_parse / aa_stream_decode AND to aa_submit_frames (the
 * frame tag is read by the host pre-pass; the GPU lanes read zeros past the end of what a frame has, like the host's). */
aa_status aa_stream_set_error_concealment( aa_stream * s, int on );
int aa_stream_error_concealment( const aa_stream * s );

/* Number of frames appended so far.  aa_stream_release_before: the caller is done with frames < first_kept -- their raster
 * handles are dropped (a raster lives on while a reference points at it: RasterHandle semantics, raster_handle.cc:113-122)
 * and the parsed records of the decoded ones go back to the context's pools (reused once queued kernels have run). */
int aa_stream_frame_count( const aa_stream * s );
aa_status aa_stream_release_before( aa_stream * s, int first_kept );
/* Forget all frames but keep decoder state? No: rewind the DEVICE half to frame 0 so the same resident
 * records can be decoded again (used by bench.py's steps; parser state is untouched). */
aa_status aa_stream_rewind( aa_stream * s );
/* Same, to frame `frame_index` (a key frame, or a frame whose reference rasters are all still held). */
aa_status aa_stream_rewind_to( aa_stream * s, int frame_index );

/* Output raster of a decoded frame (VP8Raster: three padded planes, stride = padded width, raster.hh:54-56).
 * Synchronises with the compute stream, then D2H.  Any pointer may be NULL. */
aa_status aa_stream_download( aa_stream * s, int frame_index, uint8_t * y, uint8_t * u, uint8_t * v );
/* The same without stalling the decoder (what a player that shows frames while decoding the next ones wants, player.cc:134-144):
 * the copy is queued on the context's COPY stream behind the work the compute stream holds at the time of the call and the
 * call returns; the planes -- pinned memory, aa_pinned_alloc -- are valid after aa_stream_download_wait (or aa_ctx_sync). */
aa_status aa_pinned_alloc( aa_ctx * ctx, size_t bytes, void ** out );
void aa_pinned_free( void * p );
aa_status aa_stream_download_async( aa_stream * s, int frame_index, uint8_t * y, uint8_t * u, uint8_t * v );
aa_status aa_stream_download_wait( aa_stream * s );
/* A whole batch at once -- what frontend/vp8decode.cc:78-93 / decode-bundle.cc:92-99 do with every shown frame, for n decoders in
 * lock step: frame frame_index[i] of streams[i] -> dst + i * stride (pinned memory, aa_pinned_alloc; stride a multiple of 16 and at
 * least a raster: Y, U, V planes back to back as VP8Raster pads them).  One gather kernel behind the reconstruction on the
 * compute stream and ONE copy on the copy stream, instead of 3 n plane copies; valid after aa_ctx_download_wait / aa_ctx_sync. */
aa_status aa_download_batch_async( aa_ctx * ctx, aa_stream * const * streams, int n, const int * frame_index, uint8_t * dst, size_t stride );
aa_status aa_ctx_download_wait( aa_ctx * ctx );
/* ... or only until at most max_in_flight of the batches queued by aa_download_batch_async are still on their way (oldest first): a
 * player that writes into a ring of r destination buffers calls this with r - 1 before it reuses one -- it waits for the copy that used
 * the buffer, not for the one queued a moment ago (what vp8decode.cc's display loop gets from double-buffered rasters). */
aa_status aa_ctx_download_wait_until( aa_ctx * ctx, int max_in_flight );
/* Device pointers of a frame's planes (valid while the frame's raster is alive). */
aa_status aa_stream_raster_device( aa_stream * s, int frame_index, void ** y, void ** u, void ** v );
/* References::last/golden/alternative after the most recently SUBMITTED frame: frame indices (-1 = initial blank). */
aa_status aa_stream_references( const aa_stream * s, int * last, int * golden, int * alternate );
/* Identity of the three reference rasters as they stand now (all frames handed to aa_decode_batch so far applied): equal
 * numbers = the same raster; a number stays the same for as long as that raster is a reference.  Lets a binding keep ONE
 * handle per raster across frames, also for rasters no frame produced (the blank initial one, imported ones). */
aa_status aa_stream_reference_slots( const aa_stream * s, int slots[3] );
/* Replace all three references by a raster given as device planes (e.g. received by an RCCL broadcast over xGMI):
 * the entry-state hand-off of xc-decode-bundle (decoder.cc:171-175, References(EncoderStateDeserializer&)). */
aa_status aa_stream_import_reference( aa_stream * s, const void * y_dev, const void * u_dev, const void * v_dev );
/* Same from host planes. */
aa_status aa_stream_import_reference_host( aa_stream * s, const uint8_t * y, const uint8_t * u, const uint8_t * v );

/* Copy a decoded frame's raster into caller-owned DEVICE planes (e.g. a torch tensor used as RCCL send buffer);
 * asynchronous on the compute stream. */
aa_status aa_stream_export_raster( aa_stream * s, int frame_index, void * y_dev, void * u_dev, void * v_dev );
/* The stream's DecoderState (its parser), same blob as aa_parser_export_state / aa_parser_import_state. */
size_t aa_stream_state_size( const aa_stream * s );
aa_status aa_stream_export_state( const aa_stream * s, uint8_t * buf, size_t capacity );
aa_status aa_stream_import_state( aa_stream * s, const uint8_t * buf, size_t size );

/* The whole decoder as the reference writes it to a `.state` file: Decoder::serialize / Decoder::deserialize
 * (decoder.cc:54-81) = DecoderState + References (the LAST raster only; golden and alternative alias it after loading,
 * decoder.cc:171-197).  Files written by the reference's xc-enc -O / read by vp8decode -s and xc-decode-bundle load here
 * and vice versa.  Everything parsed must have been submitted; serialize waits for the device.  buf == NULL: size query. */
aa_status aa_stream_serialize( aa_stream * s, uint8_t * buf, size_t capacity, size_t * size );
aa_status aa_stream_deserialize( aa_stream * s, const uint8_t * buf, size_t size );

/* Hashes as the reference computes them (boost::hash_combine / hash_range, the pre-1.81 formula, over 64-bit size_t):
 *   DecoderState::hash (decoder.cc:266-281), BaseRaster::raw_hash of a decoded frame (raster.cc:52-61; computed on the host
 *   from a copy of the planes -- the recurrence is serial -- and cached per raster like HashCachedRaster),
 *   DecoderHash = {state, last, golden, alternative} and its hash (decoder.cc:143-153,482-490),
 *   Decoder::minihash = low 32 bits (decoder.cc:516-529): what xc-decode-bundle checks against IVF header bytes 28-31. */
aa_status aa_parser_state_hash( const aa_parser * p, uint64_t * out );
aa_status aa_stream_state_hash( aa_stream * s, uint64_t * out );
aa_status aa_stream_raster_hash( aa_stream * s, int frame_index, uint64_t * out );
aa_status aa_stream_decoder_hash( aa_stream * s, uint64_t parts[4], uint64_t * whole );
aa_status aa_stream_minihash( aa_stream * s, uint32_t * out );
/* Give back ONE frame: its raster handle (the raster lives on while a reference points at it) and, once decoded, its
 * parsed records -- what the destructor of the last RasterHandle of a frame does in the reference (raster_handle.cc:113-122). */
aa_status aa_stream_release_frame( aa_stream * s, int frame_index );
/* References{ last, golden, alternative } (Decoder( DecoderState, References ), decoder.cc:43-46): planes[i] = Y, U, V of
 * reference i, in HBM (is_host[i] == 0) or in host memory; references given by the same Y pointer become one raster.
 * aa_stream_reference_device / _download: the current References of a stream (which: 0 last, 1 golden, 2 alternative),
 * also before any frame (the blank raster) or after an import. */
aa_status aa_stream_set_references( aa_stream * s, const void * const planes[3][3], const int is_host[3] );
aa_status aa_stream_reference_device( aa_stream * s, int which, void ** y, void ** u, void ** v );
aa_status aa_stream_reference_download( aa_stream * s, int which, uint8_t * y, uint8_t * u, uint8_t * v );

/* Padded plane geometry for a display size (VP8Raster ctor, prediction.cc:94-97). */
void aa_raster_geometry( uint16_t width, uint16_t height, uint32_t * padded_width, uint32_t * padded_height );

/* Encoder feedback.  The reference update of Encoder::write_frame (encoder.cc:146-170: frame.decode +
 * frame.loopfilter + copy_to) IS aa_stream_decode of the frame it has just serialised.  What the encoder does BEFORE it knows
 * the loop-filter level -- Encoder::apply_best_loopfilter_settings (encoder.cc:459-516): filter a copy of the reconstruction
 * with every candidate level, score it against the original, keep the best -- is this call: `data` = the frame serialised
 * with any provisional level, `original_luma` = the original's padded luma plane (padded width x padded height, stride =
 * padded width: BaseRaster::Y(), util/raster.hh:54-60), candidates level_lo..level_hi (0..63; the reference tries 0..63 on
 * the first frame and last-1..last+1 afterwards, encoder.cc:477-487).  The stream's own state and references are not touched.
 * -> best_level / best_ssim by the reference's rule (ascending levels, stop at the first that does not improve);
 * ssim_out[level_hi - level_lo + 1] (optional): every candidate's score; rasters_out (optional, host): every candidate's
 * filtered raster, Y U V planes back to back. */
aa_status aa_stream_lf_search( aa_stream * s, const uint8_t * data, size_t size, const uint8_t * original_luma,
                               int level_lo, int level_hi, int * best_level, double * best_ssim, double * ssim_out, uint8_t * rasters_out );

/* BaseRaster::quality (util/raster.cc:63-66: x264's SSIM of two planes, stride = width) for planes in HOST memory. */
aa_status aa_ssim_host( const uint8_t * a, const uint8_t * b, int width, int height, double * out );

/* Per-kernel timing of the device half, measured with HIP events on the compute stream.
 * enable=1 brackets every kernel launch with events (serialises nothing beyond event records). */
typedef struct aa_kernel_stats {
  double recon_inter_ms, recon_intra_ms, loopfilter_ms;      /* summed launch durations */
  uint64_t recon_inter_launches, recon_intra_launches, loopfilter_launches;
  uint64_t macroblocks;                                      /* MBs processed by decode_batch calls */
  double parse_headers_ms, parse_tokens_ms;                  /* device-side entropy decode: k_parse_mb_headers, k_parse_tokens */
  uint64_t parse_launches;                                   /* aa_submit_frames calls timed */
  uint64_t parsed_macroblocks;                               /* MBs handed to the device parser */
  double recon_split_ms;                                     /* k_recon_inter on SPLITMV macroblocks (recon_inter_ms: k_recon_inter4 only) */
  uint64_t recon_split_launches;
  /* where the host stood still (always counted, profile on or off): waiting for released HBM pieces to come back because
   * hipMalloc was out of memory; waiting in aa_decode_batch for the device parser to finish the frames asked for */
  uint64_t pool_waits;
  double pool_wait_ms, parse_wait_ms;
  double bind_wait_ms;      /* aa_decode_batch waiting for one of its (16) raster-binding buffers: the compute stream is that far behind */
  double alloc_ms;          /* host time inside the device pool allocator (hipMalloc of new slabs included) */
  uint64_t slab_mallocs;    /* hipMalloc calls of the pool */
  /* token workers (device-side entropy decode) */
  uint64_t token_steps;     /* decode steps of the frames whose parse has been waited for (an upper bound on the bools decoded) */
  uint64_t token_frames;
  uint64_t worker_launches, worker_wgs;   /* worker grids launched / workgroups in them */
  uint64_t worker_retires;  /* grids told to finish so that their stream could take a new grid */
  uint64_t heap_grows;      /* pieces of memory mapped into the coefficient heap */
  uint64_t heap_mapped_bytes;
  uint64_t nomem_retries;   /* frames a lane handed back because the coefficient pool was empty, run again */
  uint64_t frames_evicted;  /* frames parsed ahead of their turn whose chunks were taken back for a frame needed now (parsed again later) */
  uint64_t host_routed_frames; /* frames of aa_submit_frames calls that were parsed by host cores (few streams: one worker per stream; else host lanes) */
  /* packed coefficient storage (aa_ctx_set_packed_coefficients) */
  /* (these two took the place of expand_ms / expand_launches: no expansion pass since round 6) */
  uint64_t row_handoff_stale_polls; /* ... of row_handoff_rereads: waits whose poll, repeated BEHIND the reads that saw the value, still returned the old one */
  uint64_t row_handoff_rereads;   /* waits of the row-pipelined kernels for the row above that the slow path's second look ended (kernels.hip, reread_progress):
                                     since the context was created, not reset; counts, not errors */
  uint64_t packed_frames, packed_words, packed_blocks;   /* frames stored packed, the 16-bit words they took, the dense blocks they stand for */
  /* host lanes: host_batch_parse_cpu_ms = the workers' summed parse time (CPU seconds x 1000; a diagnostic sum); host_batch_ms,
   * host_batch_parse_wall_ms, host_batch_arena_ms: 0 since round 5 (nothing of a submit call waits for host parsing any more) */
  double host_batch_ms, host_batch_parse_wall_ms, host_batch_parse_cpu_ms, host_batch_arena_ms;
  uint64_t pinned_allocs;
} aa_kernel_stats;
aa_status aa_ctx_profile( aa_ctx * ctx, int enable );
aa_status aa_ctx_kernel_stats( aa_ctx * ctx, aa_kernel_stats * out, int reset );

#ifdef __cplusplus
}
#endif
#endif /* ALFALFA_AMD_H */
