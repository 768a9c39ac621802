// Records the HIP kernels read, as laid out in HBM.  Shared by runtime.cpp (host) and kernels.hip (device).
#pragma once
#include <stdint.h>

#include "../../include/alfalfa_amd.h"

// One decoded-frame job, resident in HBM (one per (stream, frame)); built by the host when the frame is
// parsed: raster slots are assigned then (References bookkeeping = Frame::copy_to, frame.cc:271-307), so a
// job is self-contained and can be (re)played without touching the host.
struct aa_dev_frame {
  uint8_t * cur[3];            // output raster planes Y,U,V (padded, stride = padded width; raster.hh:54-56)
  const uint8_t * ref[4][3];   // [1] last, [2] golden, [3] altref planes ([0] unused)
  const aa_mb_info * mbs;      // mb_width*mb_height records
  const int16_t * coeffs;      // compact 16-coefficient blocks
  const unsigned long long * intra_rows;   // per MB row: ceil(mbw/64) words, bit c set = MB (c,row) is intra
  uint16_t quant[4][6];        // per segment {y_dc,y_ac,y2_dc,y2_ac,uv_dc,uv_ac}
  uint16_t mbw, mbh;
  uint8_t key_frame;
  uint8_t loop_filter_level;   // header value (0: no filtering at all)
  uint8_t sharpness;
  uint8_t has_intra;
  uint32_t packed;             // 1: packed coefficient storage (coeff_pack.hh) -- coeffs = the coefficient heap's base and a macroblock's record
                               // holds the offset of its words (reserved << 32 | coeff_index, in 16-bit words); 0: dense 16-coefficient blocks
};

// The raster planes of one frame job.  A job's records are final when the frame is parsed, but WHICH rasters it writes and
// predicts from is decided when it is handed to reconstruction (References bookkeeping, frame.cc:271-307): frames that are
// parsed far ahead of their decode -- the device parser keeps several batches in flight -- hold no raster until then.
struct aa_raster_binding {
  aa_dev_frame * job;
  uint8_t * cur[3];
  const uint8_t * ref[3][3];     // last, golden, altref
};

// One raster on its way out (aa_download_batch_async): `bytes` from `src` into piece i of the staging area
struct aa_gather_job {
  const uint8_t * src;
  size_t bytes;
};

#define AA_MAX_XCD 16

#define AA_SYNC_WS_DUMP 140
#define AA_SYNC_WS_RESCUES 132
// In-launch ordering state of the row-pipelined kernels; zeroed (hipMemsetAsync) before every such launch.
struct aa_sync_ws {
  int error;                   // != 0: a bounded spin expired (sticky; reported as AA_ERR_HIP by the host).  NOT zeroed per launch.
  int where[3];                // diagnostics of the FIRST expired wait: unit, row, need << 16 | seen
  int dump[AA_SYNC_WS_DUMP];   // ... and what that wave saw when it gave up: progress[] of every row of its unit (128), then polls, clock ticks waited, rows;
                               // [AA_SYNC_WS_RESCUES .. +2]: waits that only the slow path's second look ended (see reread_progress), by which of its reads
  int ticket[AA_MAX_XCD];      // per-XCD queue: next (unit,row) to hand out  -- zeroed from here on before every launch
  int progress[1];             // [unit in launch][mbh_max]: macroblock columns of that row that are final
};
#define AA_SYNC_WS_ZERO_FROM ( 16 + 4 * AA_SYNC_WS_DUMP )   // byte offset of `ticket`

#define AA_MAX_BATCH 480       // frames per launch and kind: kernel argument = 480 pointers (3840 B) passed by value (limit 4 KB)

struct aa_frame_list {
  const aa_dev_frame * f[AA_MAX_BATCH];
};

// one stream's share of a segment-map pass: its frames are order[first .. first+count) (bit 31 of an entry: the map restarts
// at all-3 with that frame), map = the stream's persistent segment map in HBM (mb_width*mb_height bytes)
struct aa_seg_stream { uint8_t * map; uint32_t first, count; };

// Counters of the device-side job queue / coefficient pool / worker grids as one kernel thread saw them, in pinned host memory
#define AA_MAX_WORKER_GRIDS 16
struct aa_tok_mirror {
  uint32_t seq;                // the mirror kernel's number (written last)
  uint32_t q_head, q_publish, q_reserve;
  int32_t pool_avail;
  uint32_t pool_starving;
  uint32_t exited[AA_MAX_WORKER_GRIDS];
  // diagnostics (ALFALFA_AMD_TOKEN_PROFILE=1), summed over the waves that have left, 100 MHz ticks: [0] boundary passes [1] their
  // number [2] wave steps [3] looking for / starting frames [4] ring top-ups [5] periods [6] lane-periods with a frame [7] periods
  unsigned long long prof[8];
};

// kernels.hip / parse_kernels.hip launchers (plain C++ signatures so runtime.cpp needs no HIP kernel syntax)
namespace aa {
struct ParseJob;   // tok_fsm.hh
struct TokQueue; struct CoeffPool; struct Heap;
// device-side entropy decode of n frames (jobs resident in HBM): macroblock headers, then (streams with segmentation only)
// the segment-map pass, then tokens
// order[slot] = job index: the host sorts the frames of a batch by chain length so that the lanes of a wave finish together
int launch_parse_mb_headers( const ParseJob * jobs, const uint32_t * order, int n, void * stream );
int launch_segment_fixup( const ParseJob * jobs, const aa_seg_stream * streams, int n_streams, const uint32_t * order, void * stream );
// token workers (tok_fsm.hh, parse_kernels.hip): lanes that take frames from a queue in HBM
void token_worker_shape( uint32_t lane_bytes, int n_cus, int * lanes_out, uint32_t * lds_out, int * wgs_per_cu_out );
int launch_token_workers( TokQueue * q, unsigned long long * slots, const Heap & heap, uint32_t * exited, const uint32_t * retire, uint32_t gen, uint32_t spread,
                          unsigned long long * prof, unsigned long long linger_ticks, int wgs, int lanes, uint32_t lane_bytes, uint32_t lds, bool packed, uint32_t mp_hint, void * stream,
                          uint32_t * cu_slots = nullptr, uint32_t cu_cap = 0 );
// (cu_slots: AA_CU_SLOTS counters in HBM, zero at start -- worker workgroups resident per CU, for the per-CU admission of k_token_workers)
#define AA_CU_SLOTS ( 16 * 256 )       /* (the upper half, but for its last word, is the token lanes' store sink: 8 KB nobody reads) */
int launch_enqueue_jobs( TokQueue * q, unsigned long long * slots, const ParseJob * jobs, const uint32_t * order, int n, void * stream );
int launch_pool_push_range( const Heap & heap, uint32_t first, uint32_t count, void * stream );
int launch_pool_free_lists( const Heap & heap, const uint32_t * const * lists, int n, void * stream );
int launch_mirror_counters( const TokQueue * q, const CoeffPool * pool, const uint32_t * exited, int n_grids, const unsigned long long * prof, aa_tok_mirror * out, uint32_t seq, void * stream );
// whole-vector inter macroblocks, four per wave
int launch_recon_inter4( const aa_frame_list & list, int n, unsigned max_mbs, void * stream );
// one inter macroblock per wave; split_only: only SPLITMV macroblocks (the rest is launch_recon_inter4's)
int launch_recon_inter( const aa_frame_list & list, int n, unsigned max_mbs, bool split_only, void * stream );
int launch_recon_intra_diagonal( const aa_frame_list & list, int n, int diagonal, int row_lo, int rows, void * stream );
int launch_loopfilter_diagonal( const aa_frame_list & list, int n, int diagonal, int row_lo, int rows, void * stream );
// Row kernels are XCD-affine: unit u (frame / group) is processed by workgroups that run on XCD u % n_xcd.
// list = groups of four frames of one geometry (slot 0 never null, missing frames null)
// list = groups of four frames (any geometry; slot 0 never null, missing frames null)
int launch_recon_intra4( const aa_frame_list & list, int n_groups, int mbh_max, aa_sync_ws * ws, int n_xcd, void * stream );
// boundary: 128 bytes per (frame slot of the list, MB row, MB column) = 4 * n_groups * mbh_max * mbw_max lines; transient
int launch_loopfilter_rows4( const aa_frame_list & list, int n_groups, int mbh_max, int mbw_max, aa_sync_ws * ws, uint8_t * boundary, int n_xcd, void * stream );
// out16[x] += number of workgroups (of `blocks`) that ran on XCD x
int launch_probe_xcds( int * out16, int blocks, void * stream );
// seen[slot] = how many of the n probe kernels (one per stream) had arrived when this one gave up waiting or all were there
int launch_probe_concurrency( uint32_t * counter, uint32_t * seen, uint32_t n, unsigned long long timeout_ticks, int slot, void * stream );
int launch_bind_rasters( const aa_raster_binding * b, int n, void * stream );
// raster i (jobs[i]) -> staging + i * stride, one launch for the lot
int launch_gather_rasters( const aa_gather_job * jobs, int n, uint8_t * staging, size_t stride, size_t max_bytes, void * stream );
// per-window SSIM terms of two planes (stride = width; width a multiple of 8): (height/4 - 1) x (width/4 - 1) floats
// dst = src with lf_level := byte `segment_id` of `levels` (records are 80 bytes, 16-byte aligned)
int launch_lf_relevel( const aa_mb_info * src, aa_mb_info * dst, unsigned nmb, uint32_t levels, void * stream );
int launch_ssim_windows( const uint8_t * a, const uint8_t * b, int width, int height, float * out, void * stream );
}
