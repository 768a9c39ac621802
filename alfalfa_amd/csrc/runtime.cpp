// Device runtime behind the C ABI (include/alfalfa_amd.h): HIP context, per-stream frame store (pinned host
// arena mirrored 1:1 in HBM), raster slots with the reference's References bookkeeping, batched launches.
//
// Memory plan (MI355X: 288 GB HBM3E per GPU):
//   * every parsed frame's records (aa_dev_frame + aa_mb_info[] + compact coefficient blocks) are appended to a
//     per-stream arena: a pinned host chunk and a device chunk with IDENTICAL offsets, so an upload is one large
//     hipMemcpyAsync of the not-yet-uploaded byte range on the copy stream (side stream, event-ordered);
//   * rasters are slots of 3 tightly packed padded planes (VP8Raster, raster.hh:54-56); slots are assigned when a
//     frame is PARSED by replaying Frame::copy_to (frame.cc:271-307) on slot ids, so each job is self-contained
//     and device execution never consults the host;
//   * decode = k_recon_inter4 + k_recon_intra4 + k_loopfilter_rows4: three launches per batch step (the two row-pipelined
//     kernels order macroblock rows in-launch by per-XCD tickets + progress words); ALFALFA_AMD_SCHEDULE=diagonal selects
//     the launch-per-anti-diagonal schedule instead (kept for A/B measurement);
//   * pinned staging is pooled per context and can be given back once uploaded (aa_stream_release_staging).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <sched.h>
#include <sys/resource.h>
#include <sys/syscall.h>
#include <unistd.h>

#include "../../include/alfalfa_amd.h"
#include "device_types.h"
#include "parser.hh"
#include "tok_fsm.hh"
#include "coeff_pack.hh"

namespace {

thread_local std::string g_last_error;

// The entropy-decode kernels are long (a chain per lane, seconds) and must run BESIDE reconstruction and beside each other.
// HIP spreads streams over few hardware queues by default (4), and commands of one queue run in order: with more streams
// than queues a decode launch can end up queued behind a parse kernel.  The number of queues is read from the environment
// (GPU_MAX_HW_QUEUES) when the HIP runtime starts.  The library does not touch the host process's environment behind its back:
// aa_runtime_prepare() -- called by the bindings before their first aa_ctx_create, or by the host program before ITS first HIP
// call -- sets the variable if it is unset; a context then CHECKS how many of its streams really run side by side
// (probe_stream_concurrency) and says so: aa_ctx_info::stream_concurrency, a warning on stderr below 8 -- an error only with
// ALFALFA_AMD_REQUIRE_QUEUES=1 (several contexts of one process share the queues and a probe can arrive late behind another's grid).

aa_status fail( aa_status code, const std::string & msg ) { g_last_error = msg; return code; }

// Cores this process can really use: what the OS shows (hardware_concurrency, the affinity mask) bounded by the cgroup CPU quota.
// Round 4's GPU box shows 256 hardware threads under a quota of 16 CPUs (cpu.max = "1600000 100000"): 256 parser threads there get
// through LESS than 32 do (tools/host_parallelism.py: 25x one thread at 16-64 threads, 22x at 256), and a plan that counts on 256
// cores is off by 16x.
int effective_cpus()
{
  static const int n = [] {
    int cpus = static_cast<int>( std::thread::hardware_concurrency() );
    if ( cpus < 1 ) cpus = 1;
    cpu_set_t set;
    if ( sched_getaffinity( 0, sizeof set, &set ) == 0 ) { const int a = CPU_COUNT( &set ); if ( a > 0 && a < cpus ) cpus = a; }
    auto read_two = []( const char * path, double * a, double * b ) -> int {
      FILE * f = std::fopen( path, "r" );
      if ( !f ) return 0;
      char first[64] = { 0 };
      int got = std::fscanf( f, "%63s %lf", first, b );
      std::fclose( f );
      if ( got < 1 ) return 0;
      if ( std::strcmp( first, "max" ) == 0 ) { *a = -1; return got; }
      *a = atof( first );
      return got;
    };
    double quota = -1, period = 100000;
    if ( read_two( "/sys/fs/cgroup/cpu.max", &quota, &period ) < 1 ) {                 // cgroup v2; else v1
      double q = -1, dummy = 0;
      if ( read_two( "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", &q, &dummy ) >= 1 ) { quota = q; double pp = 100000; if ( read_two( "/sys/fs/cgroup/cpu/cpu.cfs_period_us", &pp, &dummy ) >= 1 && pp > 0 ) period = pp; }
    }
    if ( quota > 0 && period > 0 ) { const int q = std::max( 1, static_cast<int>( quota / period + 0.5 ) ); if ( q < cpus ) cpus = q; }
    return cpus;
  }();
  return n;
}
// host workers of one call: what the caller asked for (0: all), never more than twice the cores the process really gets
int worker_threads( int asked )
{
  const int cap = std::max( 1, std::min( 256, 2 * effective_cpus() ) );
  return std::max( 1, std::min( asked > 0 ? asked : cap, cap ) );
}
aa_status hip_fail( hipError_t e, const char * what )
{
  return fail( e == hipErrorNoDevice || e == hipErrorInvalidDevice ? AA_ERR_NO_DEVICE : AA_ERR_HIP,
               std::string( what ) + ": " + hipGetErrorString( e ) );
}
#define HIP_TRY( expr ) do { hipError_t e__ = ( expr ); if ( e__ != hipSuccess ) return hip_fail( e__, #expr ); } while ( 0 )

constexpr size_t kChunkBytes = size_t( 64 ) << 20;
constexpr size_t kAlign = 256;
inline size_t align_up( size_t v, size_t a = kAlign ) { return ( v + a - 1 ) / a * a; }

struct Chunk {
  uint8_t * host = nullptr;   // pinned
  uint8_t * dev = nullptr;
  size_t capacity = 0, used = 0, uploaded = 0;
  size_t pinned_bytes = 0;    // size of the host allocation (capacity shrinks to `used` when the chunk is sealed)
  size_t dev_bytes = 0;       // size of the device piece
  int live_frames = 0;        // frames whose records live here and have not been released
};

// One aa_submit_frames call: a pinned arena mirrored in HBM holding the parse jobs, the reconstruction job records, the
// result summaries and the compressed frames themselves; the device parser writes each frame's records into that frame's own
// record block.  Lives until the last of its frames is released.
struct Batch {
  uint8_t * host = nullptr, * dev = nullptr;
  size_t host_bytes = 0, dev_bytes = 0;
  int n = 0, live = 0;
  hipEvent_t hdr_done = nullptr;             // macroblock-header kernel (+ segment pass, + the hand-over to the job queue) finished
  size_t summaries_off = 0;
  uint8_t * host_dev = nullptr;              // `host` as the device sees it (the parse kernels write the summaries there)
  // Two-phase form (AA_SUBMIT_DEFER_TOKENS): the macroblock-header kernel has been queued, the token kernel has not -- the
  // coefficient blocks (9/10 of a frame's records) are only allocated when it is (aa_launch_tokens, or the first call that
  // needs the frame's records).
  bool tokens_pending = false;
  bool patch_jobs = false;                   // the jobs in HBM lack the coefficient pointers (two-phase form)
  size_t head_bytes = 0;                     // parse jobs + reconstruction job records at the start of the arena
  const uint32_t * launch_order_dev = nullptr;
  uint32_t * launch_order_host = nullptr;    // ... in the pinned half: the LIVE items only, longest chains first
  int n_order = 0;                           // entries of the launch order (frames the pre-pass rejected are not in it)
  int max_mbw = 0, max_nparts = 1;
  hipStream_t ps = nullptr;
  int parse_stream_index = 0;
  // on_host: the frame was handed to a HOST LANE, not to the GPU's job queue (host_lane_run): its dense coefficient blocks are a
  // pool piece of its own (host_dense: written by the worker before the frame's `done` word, read after it)
  struct Item { aa_stream * s; int frame; bool live; bool on_host = false; size_t data_off = 0; uint8_t * host_dense = nullptr; size_t host_dense_bytes = 0; };
  std::vector<Item> items;                   // [n]; live: accepted and not released since
};

// A raster: three padded planes in one piece of the context's device pool.  The piece goes back to the pool when the last
// holder lets go (RasterHandle semantics, raster_handle.cc:113-122) -- rasters are pooled per GPU, not per stream, so a stream
// that is idle between two groups of pictures holds one raster (its last reference), not a private stock.
struct Slot {
  uint8_t * dev = nullptr;    // null: a free entry of the stream's slot table
  bool shared = false;        // the context's blank raster of this size (what a new decoder's references point at): not this stream's to free
  int refs = 0;               // References + frame handles holding this raster
  bool hash_valid = false;    // HashCachedRaster: the hash is computed once per raster (raster_handle.cc:196-206)
  uint64_t hash = 0;
};

struct FrameRec {
  aa_frame_header hdr;
  aa_dev_frame * host_job = nullptr;       // in pinned chunk
  const aa_dev_frame * dev_job = nullptr;  // same offset in device chunk
  int out_slot = -1;
  int ref_after[3] = { -1, -1, -1 };       // References (last, golden, alt) after this frame, as frame indices
  std::vector<uint8_t> intra_diagonals;    // [d] != 0: diagonal d holds an intra MB
  bool has_split = false;                  // some macroblock is SPLITMV (handled by the one-macroblock-per-wave kernel)
  bool handle_held = true;
  int chunk = -1;                          // host-parsed frame: index of the frame-store chunk holding its records
  Batch * batch = nullptr;                 // device-parsed frame: its submit call ...
  int batch_item = -1;                     // ... and its index there
  bool summary_pending = false;            // counts (intra macroblocks, coefficient blocks, SPLITMV) not yet read back from the device parser
  uint8_t * rec_block = nullptr;           // device-parsed frame: macroblock records, intra row masks, flags, chunk list in HBM
  size_t rec_bytes = 0;
  bool rec_in_arena = false;               // ... inside the device half of its batch's arena (freed with the batch)
  const uint32_t * chunk_list = nullptr;   // ... the list of coefficient chunks its token lane took (in rec_block; [0] = count)
  aa_mb_info * dev_mbs = nullptr;          // ... its macroblock records (in rec_block)
  const uint32_t * packed_pos = nullptr;   // ... packed coefficient storage: where every macroblock's words start (in rec_block); null: dense blocks
  volatile aa::FrameSummary * summary = nullptr;   // in the batch's pinned arena: the token lane's last word lands here
  const aa::ParseJob * parse_job = nullptr;        // in the batch's device arena
  bool enqueued = false;                   // handed to the job queue (a token lane may be writing its records)
  uint32_t est_chunks = 0;                 // coefficient chunks accounted for this frame (an estimate until the parse is over)
  bool chunks_returned = false;            // the frame was handed back for lack of memory and its chunks are in the pool again
  bool records_released = false;
  bool placed = false;                     // raster slot + References bookkeeping done (at the first decode submission)
};

} // namespace

struct aa_parser { aa::Parser impl; aa_parser( uint16_t w, uint16_t h ) : impl( w, h ) {} };

struct aa_ctx {
  int device = 0;
  hipStream_t compute = nullptr, copy = nullptr;
  hipEvent_t upload_done = nullptr;
  std::atomic<int> refs { 1 };   // the context handle + one per stream: freed by whoever drops the last (bindings may finalise in any order)
  // device-side entropy decode: submit calls rotate over a few HIP streams so that the parse of one batch runs beside the
  // parse of the next and beside reconstruction (a parse is a few thousand latency-bound chains, not a chip-filling kernel)
  struct BindBuf { aa_raster_binding * host = nullptr, * dev = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool busy = false; };
  static constexpr int kBindBufs = 16;      // aa_decode_batch calls the host may run ahead of the compute stream
  BindBuf bind_bufs[kBindBufs];
  int next_bind_buf = 0;
  // HOST LANES: host cores in the role of token lanes.  A frame handed to them (AA_SUBMIT_HOST on a call with many streams: the key
  // frames a pipeline needs at once -- 35 ms on a core, 2 s as a chain on a GPU lane) has had its header pre-pass like every frame
  // of the device route; a worker thread parses macroblock headers and tokens from the batch arena's pinned copy (aa::parse_frame_body),
  // uploads the records to where the kernels expect them and then writes the SAME `done` word a GPU lane writes -- the submit call does
  // not wait, and everything that waits for a frame's parse (aa_decode_batch, aa_stream_frame_header, release) waits for that word.
  struct HostLanes {
    struct Task { Batch * b; int item; };
    std::mutex mu; std::condition_variable cv;
    std::deque<Task> q;
    std::vector<std::thread> threads;
    bool stop = false;
    // what the workers have really achieved (parse time only: no allocation, no upload) and what they have been given and not finished
    std::atomic<uint64_t> parsed_bytes { 0 }, parse_us { 0 }, backlog_bytes { 0 };
    uint64_t parse_us_mark = 0;          // parse_us at the last aa_ctx_kernel_stats reset
  } host_lanes;
  // the raster list of a batched download (aa_download_batch_async), read by k_gather_rasters over the bus
  struct GatherBuf { aa_gather_job * host = nullptr, * dev = nullptr; size_t cap = 0; hipEvent_t done = nullptr; bool busy = false; };
  GatherBuf gather_bufs[kBindBufs];
  int next_gather_buf = 0;
  // A parse batch holds its stream for as long as its longest chain (seconds for a key frame): a batch queued behind another
  // one on the same stream starts that much later.  Hence one stream per batch that can be in flight, and a batch goes to a
  // stream that has nothing queued (pick_parse_stream).
  static constexpr int kMaxParseStreams = 20;
  // Hardware queues are few (16 asked for above) and a queue runs its commands in order: a stream that shares a queue with a
  // worker grid -- which stays for as long as there is work -- would not get a kernel started until that grid leaves.  So the
  // context keeps to 16 streams: compute, copy, utility, expansion, 4 header-parse streams (short kernels now), 8 worker streams.
  int n_parse_streams = 4;
  std::vector<hipStream_t> parse_streams;
  std::vector<hipEvent_t> parse_idle;    // recorded behind the last operation queued on the stream
  int prio_low = 0;
  int next_parse_stream = 0;
  hipEvent_t last_seg_batch = nullptr;  // segment-map passes of consecutive batches must run in order
  std::vector<Batch *> deferred;        // batches whose frames have not been handed to the job queue yet, oldest first
  // ---- token workers: job queue, coefficient heap, worker grids (tok_fsm.hh) ----
  struct Tok {
    bool ready = false;
    hipStream_t util = nullptr;          // mirror kernel, heap pushes: never behind anything long
    hipStream_t host_up = nullptr;       // the HOST LANES' uploads (records, dense blocks, the job-record patch of a frame a host core parsed): a stream of their
                                         // own since round 6 (the hardware queue the expansion kernels of rounds 4-5 had).  On the copy stream -- which is also the
                                         // download stream, whose copies wait for compute-stream events -- a key frame's `done` word waited behind every queued
                                         // raster download and the reconstruction backlog in front of it
    // job queue
    aa::TokQueue * q = nullptr;
    unsigned long long * slots = nullptr;
    uint32_t q_slots = 1u << 18;
    uint64_t jobs_enqueued = 0;          // tickets the host has scheduled (a ticket = one frame, rejected ones included)
    std::vector<Batch *> inflight;       // batches with frames handed to the queue that may not be through yet (a batch leaves when it dies)
    std::vector<Batch *> batches;        // every live batch whose frames went to the queue, oldest first (eviction walks it from the back)
    uint32_t * one_dev = nullptr;        // a zero in HBM: the `order` of a one-job hand-over
    // coefficient heap: ONE virtual range, physical memory mapped as the frames need it
    uint8_t * heap = nullptr;
    size_t heap_va = 0, heap_mapped = 0, heap_limit = 0, grow_bytes = size_t( 1 ) << 30;
    bool vmm = false;
    std::vector<hipMemGenericAllocationHandle_t> handles;
    aa::CoeffPool * pool = nullptr;
    uint32_t * ring = nullptr;
    int64_t chunks_committed = 0;        // chunks frames hold (parsed: what they took) or are expected to take (in flight: estimate)
    double blocks_per_byte = 1.0;        // running estimate: coefficient blocks a frame stores per byte of its compressed size (never more than 25 per macroblock)
    // Packed coefficient storage (tok_fsm.hh): the lanes write a mask word + the non-zero coefficients of a block instead of 16
    // coefficients; frames are expanded into a transient dense array when they are handed to reconstruction.  One format per
    // context, fixed when the first frame is submitted.  The default since round 4 (ALFALFA_AMD_PACKED=0 /
    // aa_ctx_set_packed_coefficients( ctx, 0 ): dense blocks).
    bool packed = true;
    // One lane per DCT partition (tok_fsm.hh): frames with 2 / 4 / 8 partitions may be decoded by as many lanes of one wave.
    // Per context, fixed at the first submit (ALFALFA_AMD_LANE_PER_PARTITION=1 / aa_ctx_set_lane_per_partition).
    bool lane_per_partition = false;
    uint32_t mp_hint = 1;                // most partitions a frame submitted so far had (workgroups leave that many lanes per ticket)
    double words_per_byte = 4.0;         // running estimate for packed frames: 16-bit words stored per compressed byte
    uint32_t seen_starving = 0;
    std::vector<const uint32_t *> pending_lists;   // chunk lists of released frames, not yet handed to k_pool_free_lists
    // worker grids: slot g = worker stream g; counters are cumulative over the grids a slot has run
    static constexpr int kSlots = 8;
    struct Slot { hipStream_t st = nullptr; uint32_t launched = 0, gen = 0;
                  bool queued_behind_retiring = false;      // a grid was told to finish and a new one waits behind it on the stream ...
                  uint32_t retiring_until = 0; } slot[kSlots];  // ... until the slot's exit count reaches this (the old grid is gone)
    uint32_t * exited_dev = nullptr;     // [AA_MAX_WORKER_GRIDS] in HBM
    unsigned long long * prof_dev = nullptr;   // diagnostics counters (ALFALFA_AMD_TOKEN_PROFILE=1), else null
    uint32_t * retire_host = nullptr, * retire_dev = nullptr;     // [AA_MAX_WORKER_GRIDS]: grids of generation <= this take no more jobs
    aa_tok_mirror * mirror_host = nullptr, * mirror_dev = nullptr;
    uint32_t mirror_seq = 0;
    hipEvent_t mirror_ev = nullptr;      // tok_peek_mirror: behind the refresh that is under way
    bool mirror_inflight = false;
    uint32_t lane_bytes = 0, lds = 0;
    int lanes = 0, cap_wgs = 0, n_cus = 0;
    // An idle wave stays this long before it leaves (100 MHz ticks; ALFALFA_AMD_WORKER_LINGER_MS).  100 ms: a wave that stays holds its
    // workgroup's 41 KB of LDS and its SIMD's wave slot against the reconstruction kernels for nothing; a wave that left costs a grid
    // launch when frames come again.  Measured on MI355X, round 5 (profiles/r05_bench_sessions.md, sessions 13/14, the driver's command):
    // 2 s -> 97.6-103.1 M macroblocks/s; 200 ms -> 114.6; 100 ms -> 117.7; 50 ms -> 118.4; 20 ms -> 116.3 (5-7 grids, 1 700-2 400
    // workgroups launched in the 20 steps instead of 1 grid of ~450) -- the drain above all: the last four steps take 0.6 s instead of 1.3
    unsigned long long linger_ticks = 10000000ull;
    // A top-up grid is launched when jobs wait and at least cap_wgs / this many workgroups are gone (ALFALFA_AMD_TOPUP_DIV).  An eighth:
    // with a quarter a 40-step run sat at 578 of 768 workgroups (just above the threshold) for twelve steps while the queue grew to
    // 6 700 jobs, and its drain then waited 1.5 s for the chains that started late (session 17); over 20 steps a quarter and an eighth
    // measure the same (117.8 M both, session 16), a sixteenth spends the worker streams on small grids.
    int topup_div = 8;
  } tok;
  // Device pieces given back while kernels that read them may still be queued: they become reusable once an event recorded
  // on the compute stream after the release has fired ("epochs": one event per group of releases, recorded lazily).
  struct PendingFree { uint8_t * p; size_t bytes; uint64_t epoch; };
  std::vector<PendingFree> pending_free;
  struct Epoch { uint64_t id; hipEvent_t compute, copy; };   // copy: null unless downloads were queued on the copy stream
  std::vector<Epoch> epoch_events;      // closed epochs, oldest first
  uint64_t open_epoch = 1;
  bool open_epoch_used = false;
  // Between the first raster binding of an aa_decode_batch call and its last launch no epoch may be closed: a raster that
  // bind_frame releases (an old reference, an output nobody holds) is still read or written by kernels that are not queued
  // yet, and an event recorded now would sit in FRONT of them on the compute stream.
  int binding_depth = 0;
  bool copy_reads_rasters = false;      // aa_stream_download_async queued copies since the last epoch was closed
  uint32_t stream_concurrency = 0, streams_needed = 0;    // probe_stream_concurrency (at the first submit)
  uint32_t clock_mhz = 0;                                 // hipDeviceAttributeClockRate, asked once
  bool profile = false;
  double host_share_ms = 80.0;
                                 // aa_submit_frames: a big call's key frames go to the host lanes while their backlog stays within this (0: never)
  int schedule = 0;            // 0: row-pipelined persistent kernels (default), 1: one launch per 2:1 anti-diagonal
  int n_xcd = 1;               // XCDs workgroups land on (probed at creation); row kernels keep a unit on one XCD
  int xcd_share[AA_MAX_XCD] = {};
  std::mutex pool_mu;           // pinned staging chunks given back by aa_stream_release_staging, reused by later parses
  std::vector<std::pair<uint8_t *, size_t>> pinned_pool;
  // Device memory of the streams (frame-store chunks, raster blocks) is carved out of 256-MiB slabs: allocation calls
  // serialise in the driver and hundreds of parser threads make them at once.  Freed pieces go to per-size free lists
  // (streams of one frame size all ask for the same two sizes); slabs are returned when the context dies.
  std::vector<uint8_t *> dev_slabs;
  uint8_t * cur_slab = nullptr;
  size_t slab_used = 0;
  size_t pool_bytes = 0;                  // HBM the pool has taken from HIP so far
  size_t pool_soft_limit = ~size_t( 0 );  // beyond this (slabs + mapped coefficient heap) the pool waits for released pieces rather than grow
                                          // (aa_ctx_create: 7/8 of what was free; aa_ctx_set_memory_limit)
  size_t pinned_bytes = 0;                // pinned host memory the context has taken (arenas, staging chunks, binding buffers)
  std::map<size_t, std::vector<uint8_t *>> dev_free;
  // Pieces whose every use -- by the old owner and by the next one -- is a kernel (or copy) queued on the COMPUTE stream: output
  // rasters and the transient dense coefficient blocks of a reconstruction call.  Stream order alone makes them reusable the
  // moment they are released (whatever still reads the old content was queued earlier on the same stream), so they do not wait
  // for a release epoch: a step releases ~45 GB of them at 1080p x 480 streams (a raster per frame, a dense piece per call),
  // which the epochs kept out of reach for one step -- 45 GB of pool that the coefficient heap could not have.  Only handed
  // out by dev_alloc_compute.  (A raster that a plain aa_stream_download_async may still be reading on the COPY stream takes the
  // epoch route: raster_download_pending.)
  std::map<size_t, std::vector<uint8_t *>> compute_free;
  size_t compute_free_bytes = 0;
  // ... released WHILE an aa_decode_batch call binds rasters (binding_depth > 0): the frame being bound still predicts from the
  // reference it has just let go of, and its kernels are not queued yet -- such a piece must not become another frame's output of
  // the same call.  It is held until the call's launches are queued (tests/test_gpu_parity.py::test_rasters_released_while_binding...).
  std::vector<std::pair<uint8_t *, size_t>> compute_hold;
  hipEvent_t last_raster_download = nullptr;   // recorded on the copy stream behind the last aa_stream_download_async
  bool raster_download_pending = false;
  uint64_t downloads_queued = 0, downloads_recorded = 0;   // aa_stream_download_async calls begun / whose event has been recorded (under pool_mu)
  std::mutex blank_mu;
  std::map<size_t, uint8_t *> blank;    // References( width, height ): one all-zero raster per raster size, shared by every new decoder (never written)
  uint8_t * boundary = nullptr; // loop filter: hand-off lines between macroblock rows (transient within a launch)
  size_t boundary_bytes = 0;
  aa_sync_ws * ws = nullptr;   // in-launch ordering state of the row-pipelined kernels
  size_t ws_bytes = 0;
  aa_kernel_stats stats {};
  struct Timed { hipEvent_t a, b; int kind; };
  std::vector<Timed> pending;
  std::vector<hipEvent_t> free_events;
};

namespace { void host_lanes_start( aa_ctx * ctx ); void host_lanes_stop( aa_ctx * ctx ); double host_lanes_rate( aa_ctx * ctx ); }

struct aa_stream {
  aa_ctx * ctx;
  aa::Parser parser;
  uint8_t * dev_segmap = nullptr;   // persistent segment map in HBM (device-parsed frames), mb_width*mb_height bytes
  bool segmap_on_device = false;    // which copy is current: the parser's (host) or dev_segmap
  uint32_t pw, ph;
  size_t plane_bytes[3], slot_bytes;
  std::vector<Chunk> chunks;
  std::vector<Slot> slots;
  std::vector<FrameRec> frames;
  int cur_ref_slot[3];     // slot ids of last/golden/alt at PARSE time
  int cur_ref_frame[3];    // frame indices (for aa_stream_references)
  int next_submit = 0;     // device half progress
  int first_live = 0;      // frames below this index are fully released (aa_stream_release_before scans from here)
  aa_stream( aa_ctx * c, uint16_t w, uint16_t h ) : ctx( c ), parser( w, h ) {}
};

namespace {

aa_status set_device( aa_ctx * ctx ) { HIP_TRY( hipSetDevice( ctx->device ) ); return AA_OK; }

constexpr size_t kSlabBytes = size_t( 256 ) << 20;

// pool_mu held.  Close the open epoch if anything was released in it (one event on the compute stream covers every launch made
// before now, hence every launch made before those releases), then hand pieces of fired epochs to the free lists.
void flush_chunk_frees( aa_ctx * ctx );     // (below)
void collect_pending( aa_ctx * ctx, bool wait_oldest )
{
  if ( ctx->open_epoch_used && ctx->binding_depth == 0 ) {
    flush_chunk_frees( ctx );           // (coefficient chunks of released frames go back behind the kernels that read them)
    hipEvent_t e = nullptr, c = nullptr;
    bool ok = hipEventCreateWithFlags( &e, hipEventDisableTiming ) == hipSuccess && hipEventRecord( e, ctx->compute ) == hipSuccess;
    // asynchronous downloads read rasters on the COPY stream: a released raster is reusable only when those copies are done too
    if ( ok && ctx->copy_reads_rasters )
      ok = hipEventCreateWithFlags( &c, hipEventDisableTiming ) == hipSuccess && hipEventRecord( c, ctx->copy ) == hipSuccess;
    if ( ok ) {
      ctx->epoch_events.push_back( { ctx->open_epoch, e, c } );
      ctx->open_epoch++; ctx->open_epoch_used = false; ctx->copy_reads_rasters = false;
    } else {
      // no event to be had: the epoch is closed by waiting for both streams instead (never left open for good)
      if ( e ) (void) hipEventDestroy( e );
      if ( c ) (void) hipEventDestroy( c );
      (void) hipGetLastError();
      if ( hipStreamSynchronize( ctx->compute ) == hipSuccess && hipStreamSynchronize( ctx->copy ) == hipSuccess ) {
        ctx->epoch_events.push_back( { ctx->open_epoch, nullptr, nullptr } );
        ctx->open_epoch++; ctx->open_epoch_used = false; ctx->copy_reads_rasters = false;
      }
    }
  }
  uint64_t fired = 0;
  while ( !ctx->epoch_events.empty() ) {
    aa_ctx::Epoch & ep = ctx->epoch_events.front();
    bool done = true;
    for ( hipEvent_t * e : { &ep.compute, &ep.copy } ) {
      if ( !*e ) continue;
      hipError_t q = hipEventQuery( *e );
      if ( q != hipSuccess && wait_oldest ) q = hipEventSynchronize( *e );
      if ( q != hipSuccess ) { done = false; break; }
      (void) hipEventDestroy( *e ); *e = nullptr;
    }
    wait_oldest = false;
    if ( !done ) { (void) hipGetLastError(); break; }
    fired = ep.id;
    ctx->epoch_events.erase( ctx->epoch_events.begin() );
  }
  if ( !fired ) return;
  size_t keep = 0;
  for ( auto & pf : ctx->pending_free ) {
    if ( pf.epoch <= fired ) ctx->dev_free[pf.bytes].push_back( pf.p );
    else ctx->pending_free[keep++] = pf;
  }
  ctx->pending_free.resize( keep );
}

// Pieces are recycled through free lists keyed by their exact size.  Streams of one frame size ask for the same few sizes --
// but a batch arena's size follows the compressed sizes of its frames, and a freed arena of a size nobody asks for again would
// sit in its list for good (measured, round 3: 130 GB of pool for 45 GB of live pieces).  So pieces that get an allocation
// of their own (more than half a slab) come in size CLASSES: a sixteenth of the next power of two (at most 6 % over).
inline size_t pool_size_class( size_t bytes )
{
  bytes = align_up( bytes );
  if ( bytes <= kSlabBytes / 2 ) return bytes;
  size_t p2 = kSlabBytes;
  while ( p2 < bytes ) p2 <<= 1;
  const size_t step = p2 / 16;
  return ( bytes + step - 1 ) / step * step;
}

// `refusable`: the piece is for frames being HANDED OVER (a batch arena): such an allocation stops a thirty-second of the limit short
// of it, so that what the caller needs to get frames OUT again -- rasters, the dense blocks of a reconstruction call -- still finds
// room; a caller that pipelines treats AA_ERR_NO_MEMORY from aa_submit_frames as "not now" and reconstructs / releases first.
aa_status dev_alloc( aa_ctx * ctx, size_t bytes, uint8_t ** out, bool refusable = false )
{
  bytes = pool_size_class( bytes );
  std::lock_guard<std::mutex> g( ctx->pool_mu );
  struct Clock { aa_ctx * c; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                 ~Clock() { c->stats.alloc_ms += std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - t0 ).count(); } } clock { ctx };
  int soft_waits = 0;
  for ( int attempt = 0; attempt < 3; attempt++ ) {
    auto it = ctx->dev_free.find( bytes );
    if ( it != ctx->dev_free.end() && !it->second.empty() ) { *out = it->second.back(); it->second.pop_back(); return AA_OK; }
    if ( attempt == 0 && !ctx->pending_free.empty() ) { collect_pending( ctx, false ); continue; }
    const bool big = bytes > kSlabBytes / 2;         // big pieces get their own allocation (still recycled through the free list)
    if ( !big && ctx->cur_slab && ctx->slab_used + bytes <= kSlabBytes ) { *out = ctx->cur_slab + ctx->slab_used; ctx->slab_used += bytes; return AA_OK; }
    // The pool is about to grow.  Past the soft limit, pieces that were released but may still be read by queued kernels are
    // waited for instead (they come back as the compute stream advances): the pool must not creep up to the last byte of HBM.
    const size_t grow = big ? bytes : kSlabBytes;
    if ( ctx->pool_bytes + ctx->tok.heap_mapped + grow > ctx->pool_soft_limit && !ctx->compute_free.empty() ) {
      // Past the limit: what is parked for the compute stream's own reuse (rasters and dense transients of a geometry the caller may
      // have left behind) goes back to the general lists -- through an epoch, since queued kernels may still read it.
      for ( auto & kv : ctx->compute_free ) for ( uint8_t * p : kv.second ) ctx->pending_free.push_back( { p, kv.first, ctx->open_epoch } );
      ctx->compute_free.clear(); ctx->compute_free_bytes = 0; ctx->open_epoch_used = true;
    }
    if ( ctx->pool_bytes + ctx->tok.heap_mapped + grow + ( refusable && ctx->pool_soft_limit != ~size_t( 0 ) ? ctx->pool_soft_limit / 32 : 0 ) > ctx->pool_soft_limit
         && !ctx->pending_free.empty() && soft_waits < 64 ) {
      const auto t0 = std::chrono::steady_clock::now();
      collect_pending( ctx, true );
      ctx->stats.pool_waits++;
      ctx->stats.pool_wait_ms += std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - t0 ).count();
      soft_waits++; attempt = 0;
      continue;
    }
    // ... and a limit that limits what the context ACCEPTS: with nothing left to wait for, the arena of a hand-over is refused
    // (repeatable: the caller reconstructs and releases frames first, or raises aa_ctx_set_memory_limit) rather than take the
    // context past what it was told it may hold.  What gets frames OUT again -- rasters, the dense blocks of a reconstruction call --
    // is never refused: it lives on the thirty-second kept back from the hand-overs, and goes past the limit only when the caller
    // holds more decoded frames at once than that covers (the pool does not re-split the free pieces of other sizes it holds).
    // (What counts is what the context HOLDS -- live pieces + mapped heap: the pool keeps freed pieces in per-size lists and does not
    // re-split them, so after a change of geometry -- or of call sizes -- it may sit on gigabytes that fit nobody; refusing by what it
    // has TAKEN would then refuse every hand-over for good.  In a run of one geometry the free lists are what the next hand-over is
    // served from, and taken ~ held.)
    const size_t keep_back = ctx->pool_soft_limit != ~size_t( 0 ) ? ctx->pool_soft_limit / 32 : 0;
    size_t idle = ctx->compute_free_bytes + ( ctx->cur_slab ? kSlabBytes - ctx->slab_used : 0 );
    for ( auto & kv : ctx->dev_free ) idle += kv.first * kv.second.size();
    const size_t held = ctx->pool_bytes > idle ? ctx->pool_bytes - idle : 0;
    if ( refusable && held + ctx->tok.heap_mapped + grow + keep_back > ctx->pool_soft_limit && ctx->pool_bytes > 0 )
      return fail( AA_ERR_NO_MEMORY, "device pool: the context's memory limit (" + std::to_string( ctx->pool_soft_limit >> 20 ) + " MiB: pool "
                                     + std::to_string( ctx->pool_bytes >> 20 ) + " + coefficient heap " + std::to_string( ctx->tok.heap_mapped >> 20 )
                                     + ") does not allow another " + std::to_string( grow >> 20 ) + " MiB: release decoded frames or raise aa_ctx_set_memory_limit" );
    hipError_t e;
    uint8_t * piece = nullptr;
    e = hipMalloc( reinterpret_cast<void **>( &piece ), grow );
    ctx->stats.slab_mallocs++;
    if ( e == hipSuccess ) {
      ctx->dev_slabs.push_back( piece ); ctx->pool_bytes += grow;
      if ( !big ) { ctx->cur_slab = piece; ctx->slab_used = bytes; }
      *out = piece;
      return AA_OK;
    }
    // out of HBM: what was released but may still be read by queued kernels comes back once they have run -- the pieces parked for
    // the compute stream's own reuse included (another context or process may hold the rest of the HBM well below this one's limit)
    (void) hipGetLastError();
    if ( !ctx->compute_free.empty() ) {
      for ( auto & kv : ctx->compute_free ) for ( uint8_t * p : kv.second ) ctx->pending_free.push_back( { p, kv.first, ctx->open_epoch } );
      ctx->compute_free.clear(); ctx->compute_free_bytes = 0; ctx->open_epoch_used = true;
    }
    if ( attempt == 2 || ctx->pending_free.empty() ) return hip_fail( e, "hipMalloc (frame store)" );
    const auto t0 = std::chrono::steady_clock::now();
    collect_pending( ctx, true );
    ctx->stats.pool_waits++;
    ctx->stats.pool_wait_ms += std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - t0 ).count();
  }
  return fail( AA_ERR_HIP, "device allocation failed" );
}
// `deferred`: kernels already queued may still read the piece
void dev_free( aa_ctx * ctx, uint8_t * p, size_t bytes, bool deferred = false )
{
  if ( !p ) return;
  std::lock_guard<std::mutex> g( ctx->pool_mu );
  if ( deferred ) { ctx->pending_free.push_back( { p, pool_size_class( bytes ), ctx->open_epoch } ); ctx->open_epoch_used = true; }
  else ctx->dev_free[pool_size_class( bytes )].push_back( p );
}
// A piece that only the compute stream ever touched (see aa_ctx::compute_free): reusable at once by dev_alloc_compute
void dev_free_compute( aa_ctx * ctx, uint8_t * p, size_t bytes )
{
  if ( !p ) return;
  {
    std::lock_guard<std::mutex> g( ctx->pool_mu );
    if ( ctx->raster_download_pending && ctx->last_raster_download && ctx->downloads_recorded == ctx->downloads_queued
         && hipEventQuery( ctx->last_raster_download ) == hipSuccess ) ctx->raster_download_pending = false;
    (void) hipGetLastError();
    if ( !ctx->raster_download_pending ) {
      const size_t cls = pool_size_class( bytes );
      if ( ctx->binding_depth > 0 ) ctx->compute_hold.emplace_back( p, cls );
      else { ctx->compute_free[cls].push_back( p ); ctx->compute_free_bytes += cls; }
      return;
    }
  }
  dev_free( ctx, p, bytes, true );
}
// ... and an allocation whose first use is queued on the compute stream
aa_status dev_alloc_compute( aa_ctx * ctx, size_t bytes, uint8_t ** out )
{
  {
    const size_t cls = pool_size_class( bytes );
    std::lock_guard<std::mutex> g( ctx->pool_mu );
    auto it = ctx->compute_free.find( cls );
    if ( it != ctx->compute_free.end() && !it->second.empty() ) { *out = it->second.back(); it->second.pop_back(); ctx->compute_free_bytes -= cls; return AA_OK; }
  }
  return dev_alloc( ctx, bytes, out );
}

aa_status alloc_slot( aa_stream * s, int * out )
{
  uint8_t * piece = nullptr;
  if ( aa_status st = dev_alloc_compute( s->ctx, s->slot_bytes, &piece ) ) return st;     // (written by kernels / copies queued on the compute stream)
  for ( size_t i = 0; i < s->slots.size(); i++ ) if ( !s->slots[i].dev ) { s->slots[i] = Slot(); s->slots[i].dev = piece; *out = static_cast<int>( i ); return AA_OK; }
  Slot sl; sl.dev = piece;
  *out = static_cast<int>( s->slots.size() );
  s->slots.push_back( sl );
  return AA_OK;
}
void retain( aa_stream * s, int slot ) { if ( slot >= 0 ) s->slots[slot].refs++; }
void release( aa_stream * s, int slot )
{
  if ( slot < 0 ) return;
  Slot & sl = s->slots[slot];
  if ( --sl.refs == 0 ) { if ( !sl.shared ) dev_free_compute( s->ctx, sl.dev, s->slot_bytes ); sl.dev = nullptr; sl.shared = false; }   // (kernels already queued may still read it: they are ahead of the next owner's on the compute stream)
}
void set_ref( aa_stream * s, int which, int slot, int frame )
{
  retain( s, slot ); release( s, s->cur_ref_slot[which] );
  s->cur_ref_slot[which] = slot; s->cur_ref_frame[which] = frame;
}

aa_status reserve( aa_stream * s, size_t bytes, Chunk ** out )
{
  if ( s->chunks.empty() || !s->chunks.back().dev || s->chunks.back().used + bytes > s->chunks.back().capacity ) {
    if ( !s->chunks.empty() && s->chunks.back().dev && s->chunks.back().live_frames == 0 ) {   // every frame of the old tail is gone already
      Chunk & t = s->chunks.back();
      if ( t.host ) { std::lock_guard<std::mutex> g( s->ctx->pool_mu ); s->ctx->pinned_pool.emplace_back( t.host, t.pinned_bytes ); t.host = nullptr; }
      dev_free( s->ctx, t.dev, t.dev_bytes, true ); t.dev = nullptr;
    }
    Chunk c;
    c.capacity = std::max( kChunkBytes, align_up( bytes ) );
    {
      std::lock_guard<std::mutex> g( s->ctx->pool_mu );
      auto & pool = s->ctx->pinned_pool;
      for ( size_t i = 0; i < pool.size(); i++ ) if ( pool[i].second == c.capacity ) { c.host = pool[i].first; pool[i] = pool.back(); pool.pop_back(); break; }
    }
    if ( !c.host ) { HIP_TRY( hipHostMalloc( reinterpret_cast<void **>( &c.host ), c.capacity, hipHostMallocDefault ) ); std::lock_guard<std::mutex> g( s->ctx->pool_mu ); s->ctx->pinned_bytes += c.capacity; }
    c.pinned_bytes = c.dev_bytes = c.capacity;
    if ( aa_status st = dev_alloc( s->ctx, c.capacity, &c.dev ) ) { std::lock_guard<std::mutex> g( s->ctx->pool_mu ); s->ctx->pinned_pool.emplace_back( c.host, c.pinned_bytes ); return st; }
    s->chunks.push_back( c );
  }
  *out = &s->chunks.back();
  return AA_OK;
}

hipEvent_t get_event( aa_ctx * ctx )
{
  if ( !ctx->free_events.empty() ) { hipEvent_t e = ctx->free_events.back(); ctx->free_events.pop_back(); return e; }
  hipEvent_t e = nullptr; (void) hipEventCreate( &e ); return e;
}
void drain_profile( aa_ctx * ctx )
{
  for ( auto & t : ctx->pending ) {
    float ms = 0;
    if ( hipEventSynchronize( t.b ) == hipSuccess && hipEventElapsedTime( &ms, t.a, t.b ) == hipSuccess ) {
      if ( t.kind == 0 ) { ctx->stats.recon_inter_ms += ms; ctx->stats.recon_inter_launches++; }
      else if ( t.kind == 1 ) { ctx->stats.recon_intra_ms += ms; ctx->stats.recon_intra_launches++; }
      else if ( t.kind == 2 ) { ctx->stats.loopfilter_ms += ms; ctx->stats.loopfilter_launches++; }
      else if ( t.kind == 3 ) { ctx->stats.parse_headers_ms += ms; ctx->stats.parse_launches++; }
      else if ( t.kind == 4 ) ctx->stats.parse_tokens_ms += ms;
      else if ( t.kind == 6 ) { ctx->stats.expand_ms += ms; ctx->stats.expand_launches++; }
      else { ctx->stats.recon_split_ms += ms; ctx->stats.recon_split_launches++; }
    }
    ctx->free_events.push_back( t.a ); ctx->free_events.push_back( t.b );
  }
  ctx->pending.clear();
}

aa_status ensure_ws( aa_ctx * ctx, aa_sync_ws ** ws, size_t * have, int max_mbh )
{
  const size_t need = sizeof( aa_sync_ws ) + sizeof( int ) * size_t( AA_MAX_BATCH ) * max_mbh;
  if ( need <= *have ) return AA_OK;
  HIP_TRY( hipStreamSynchronize( ctx->compute ) );
  int err = 0;
  if ( *ws ) { (void) hipMemcpy( &err, &( *ws )->error, sizeof err, hipMemcpyDeviceToHost ); (void) hipFree( *ws ); }
  *ws = nullptr; *have = 0;
  HIP_TRY( hipMalloc( reinterpret_cast<void **>( ws ), need ) );
  // (hipMemset of device memory is not ordered against the context's NON-BLOCKING streams: everything that prepares memory a
  // kernel on one of them will use is either queued on that stream or followed by a device-wide wait)
  HIP_TRY( hipMemsetAsync( *ws, 0, need, ctx->compute ) );
  HIP_TRY( hipStreamSynchronize( ctx->compute ) );
  if ( err ) HIP_TRY( hipMemcpy( &( *ws )->error, &err, sizeof err, hipMemcpyHostToDevice ) );   // the error word is sticky
  *have = need;
  return AA_OK;
}
aa_status zero_ws( aa_ctx * ctx, aa_sync_ws * ws, int frames, int max_mbh )
{
  HIP_TRY( hipMemsetAsync( reinterpret_cast<uint8_t *>( ws ) + AA_SYNC_WS_ZERO_FROM, 0,
                           sizeof( aa_sync_ws ) - AA_SYNC_WS_ZERO_FROM + sizeof( int ) * size_t( frames ) * max_mbh, ctx->compute ) );
  return AA_OK;
}
struct LaunchTimer {     // events on the stream the kernel is launched on
  aa_ctx * ctx; int kind; hipEvent_t a = nullptr; hipStream_t st;
  LaunchTimer( aa_ctx * c, int k, hipStream_t on = nullptr ) : ctx( c ), kind( k ), st( on ? on : c->compute ) { if ( ctx->profile ) { a = get_event( ctx ); (void) hipEventRecord( a, st ); } }
  ~LaunchTimer() { if ( ctx->profile ) { hipEvent_t b = get_event( ctx ); (void) hipEventRecord( b, st ); ctx->pending.push_back( { a, b, kind } ); if ( ctx->pending.size() >= 8192 ) drain_profile( ctx ); } }
};


// ---------------- token workers: job queue, coefficient heap, worker grids ----------------
inline double now_ms() { return std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now().time_since_epoch() ).count(); }
inline aa::Heap heap_of( const aa_ctx * ctx )
{
  aa::Heap h;
  h.base = (AA_GLOBAL int16_t *) ctx->tok.heap; h.pool = (AA_GLOBAL aa::CoeffPool *) ctx->tok.pool; h.ring = (AA_GLOBAL uint32_t *) ctx->tok.ring;
  return h;
}
constexpr size_t kChunkBytesHeap = size_t( aa::kChunkBlocks ) * 32;

// pool_mu held.  Chunk lists of released frames -> one kernel on the COMPUTE stream: behind every reconstruction kernel that
// may still read those coefficients, in front of the event that lets the lists' own memory (the frames' record blocks) go.
void flush_chunk_frees( aa_ctx * ctx )
{
  auto & T = ctx->tok;
  if ( T.pending_lists.empty() ) return;
  if ( aa::launch_pool_free_lists( heap_of( ctx ), T.pending_lists.data(), static_cast<int>( T.pending_lists.size() ), ctx->compute ) == 0 ) T.pending_lists.clear();
  else (void) hipGetLastError();          // (kept: tried again at the next epoch)
}

aa_status tok_refresh_mirror( aa_ctx * ctx )
{
  auto & T = ctx->tok;
  T.mirror_seq++;
  if ( int e = aa::launch_mirror_counters( T.q, T.pool, T.exited_dev, aa_ctx::Tok::kSlots, T.prof_dev, T.mirror_dev, T.mirror_seq, T.util ) )
    return hip_fail( static_cast<hipError_t>( e ), "k_mirror_counters" );
  HIP_TRY( hipStreamSynchronize( T.util ) );
  T.mirror_inflight = false;
  return AA_OK;
}
// ... for callers that only LOOK at the counters (aa_ctx_get_info: a pipelining caller plans by them several times per step): ask
// for a refresh if none is under way, wait for it a few hundred microseconds at most, and otherwise make do with the last one.  On
// an idle GPU the kernel is through in tens of microseconds; beside 700 resident worker workgroups and the reconstruction kernels
// it was seen to take 50-100 ms to get its turn -- 295 ms per step of a caller that waited for it (round 5, profiles/r05_bench_sessions.md).
aa_status tok_peek_mirror( aa_ctx * ctx )
{
  auto & T = ctx->tok;
  if ( !T.mirror_ev ) HIP_TRY( hipEventCreateWithFlags( &T.mirror_ev, hipEventDisableTiming ) );
  if ( !T.mirror_inflight ) {
    T.mirror_seq++;
    if ( int e = aa::launch_mirror_counters( T.q, T.pool, T.exited_dev, aa_ctx::Tok::kSlots, T.prof_dev, T.mirror_dev, T.mirror_seq, T.util ) )
      return hip_fail( static_cast<hipError_t>( e ), "k_mirror_counters" );
    HIP_TRY( hipEventRecord( T.mirror_ev, T.util ) );
    T.mirror_inflight = true;
  }
  const double t0 = now_ms();
  for ( ;; ) {
    const hipError_t e = hipEventQuery( T.mirror_ev );
    if ( e == hipSuccess ) { T.mirror_inflight = false; break; }
    if ( e != hipErrorNotReady ) return hip_fail( e, "hipEventQuery (counters)" );
    (void) hipGetLastError();
    if ( now_ms() - t0 > 0.3 ) break;
    usleep( 20 );
  }
  return AA_OK;
}

// Map more of the heap until `want_mapped` bytes are there (or the memory limit / the reserved range says no).
aa_status tok_grow_heap( aa_ctx * ctx, size_t want_mapped )
{
  auto & T = ctx->tok;
  if ( !T.vmm ) return AA_OK;
  want_mapped = std::min( ( want_mapped + T.grow_bytes - 1 ) / T.grow_bytes * T.grow_bytes, T.heap_va );
  while ( T.heap_mapped < want_mapped ) {
    // (the limit is the pool's too, and the heap never unmaps: a sixteenth of it stays out of the heap's reach, for the arenas and
    // rasters of frames handed over after the heap has taken what it could -- round 4's driver run ended with heap + pool 2.4 GB
    // over a limit the pool could only ask about, not keep)
    { std::lock_guard<std::mutex> g( ctx->pool_mu );
      const size_t reserve = ctx->pool_soft_limit == ~size_t( 0 ) ? 0 : ctx->pool_soft_limit / 16;
      if ( ctx->pool_bytes + T.heap_mapped + T.grow_bytes + reserve > ctx->pool_soft_limit && T.heap_mapped >= T.grow_bytes ) break; }
    hipMemAllocationProp prop {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = ctx->device;
    hipMemGenericAllocationHandle_t h;
    if ( hipMemCreate( &h, T.grow_bytes, &prop, 0 ) != hipSuccess ) { (void) hipGetLastError(); break; }
    uint8_t * at = T.heap + T.heap_mapped;
    if ( hipMemMap( at, T.grow_bytes, 0, h, 0 ) != hipSuccess ) { (void) hipGetLastError(); (void) hipMemRelease( h ); break; }
    hipMemAccessDesc acc {};
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = ctx->device; acc.flags = hipMemAccessFlagsProtReadWrite;
    if ( hipMemSetAccess( at, T.grow_bytes, &acc, 1 ) != hipSuccess ) { (void) hipGetLastError(); (void) hipMemUnmap( at, T.grow_bytes ); (void) hipMemRelease( h ); break; }
    T.handles.push_back( h );
    if ( int e = aa::launch_pool_push_range( heap_of( ctx ), static_cast<uint32_t>( T.heap_mapped / kChunkBytesHeap ), static_cast<uint32_t>( T.grow_bytes / kChunkBytesHeap ), T.util ) )
      return hip_fail( static_cast<hipError_t>( e ), "k_pool_push_range" );
    T.heap_mapped += T.grow_bytes;
    ctx->stats.heap_grows++;
  }
  return AA_OK;
}

// One tiny kernel per stream of the context, all waiting for each other (k_probe_concurrency): -> how many ran side by side
aa_status probe_stream_concurrency( aa_ctx * ctx )
{
  auto & T = ctx->tok;
  std::vector<hipStream_t> all { ctx->compute, ctx->copy, T.util, T.host_up };
  for ( auto ps : ctx->parse_streams ) all.push_back( ps );
  for ( auto & sl : T.slot ) all.push_back( sl.st );
  const uint32_t n = static_cast<uint32_t>( all.size() );
  uint32_t * d = nullptr;
  HIP_TRY( hipMalloc( reinterpret_cast<void **>( &d ), sizeof( uint32_t ) * ( n + 1 ) ) );
  HIP_TRY( hipMemset( d, 0, sizeof( uint32_t ) * ( n + 1 ) ) );
  HIP_TRY( hipStreamSynchronize( nullptr ) );        // (the memset runs on the null stream and is not ordered against the non-blocking streams the probe runs on)
  for ( uint32_t i = 0; i < n; i++ )
    if ( int e = aa::launch_probe_concurrency( d, d + 1, n, 300000ull /* 3 ms */, static_cast<int>( i ), all[i] ) ) { (void) hipFree( d ); return hip_fail( static_cast<hipError_t>( e ), "k_probe_concurrency" ); }
  for ( auto st : all ) HIP_TRY( hipStreamSynchronize( st ) );
  std::vector<uint32_t> seen( n + 1 );
  HIP_TRY( hipMemcpy( seen.data(), d, sizeof( uint32_t ) * ( n + 1 ), hipMemcpyDeviceToHost ) );
  (void) hipFree( d );
  uint32_t conc = n;
  for ( uint32_t i = 0; i < n; i++ ) conc = std::min( conc, seen[1 + i] );
  ctx->stream_concurrency = conc; ctx->streams_needed = n;
  if ( conc < n ) {
    const std::string msg = "alfalfa_amd: only " + std::to_string( conc ) + " of this context's " + std::to_string( n ) + " HIP streams run side by side (hardware queues): "
                            "set GPU_MAX_HW_QUEUES=16 in the environment before the process initialises HIP (aa_runtime_prepare() does it when called first); "
                            "long-running entropy-decode grids will otherwise hold up reconstruction kernels that share their queue";
    // (said, not refused: in a process with several contexts the streams of all of them share the hardware queues, and a probe
    // kernel that lands behind another context's lingering worker grid also arrives late -- seen in the GPU test session: 7 of 15)
    const char * strict = std::getenv( "ALFALFA_AMD_REQUIRE_QUEUES" );
    if ( strict && atoi( strict ) ) return fail( AA_ERR_LOGIC, msg );
    static std::atomic<bool> said { false };
    if ( !said.exchange( true ) ) std::fprintf( stderr, "%s\n", msg.c_str() );
  }
  return AA_OK;
}

aa_status tok_init( aa_ctx * ctx )
{
  auto & T = ctx->tok;
  if ( T.ready ) return AA_OK;
  hipDeviceProp_t prop;
  HIP_TRY( hipGetDeviceProperties( &prop, ctx->device ) );
  T.n_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 1;
  HIP_TRY( hipStreamCreateWithFlags( &T.util, hipStreamNonBlocking ) );
  { int lo = 0, hi = 0; (void) hipDeviceGetStreamPriorityRange( &lo, &hi ); HIP_TRY( hipStreamCreateWithPriority( &T.host_up, hipStreamNonBlocking, hi ) ); }
  for ( auto & sl : T.slot ) HIP_TRY( hipStreamCreateWithPriority( &sl.st, hipStreamNonBlocking, ctx->prio_low ) );
  if ( aa_status st = probe_stream_concurrency( ctx ) ) return st;
  // the job queue
  HIP_TRY( hipMalloc( reinterpret_cast<void **>( &T.q ), 256 ) );
  { aa::TokQueue hq {}; hq.mask = T.q_slots - 1; HIP_TRY( hipMemcpy( T.q, &hq, sizeof hq, hipMemcpyHostToDevice ) ); }
  HIP_TRY( hipMalloc( reinterpret_cast<void **>( &T.slots ), size_t( T.q_slots ) * 8 ) );
  HIP_TRY( hipMemset( T.slots, 0, size_t( T.q_slots ) * 8 ) );
  HIP_TRY( hipMalloc( reinterpret_cast<void **>( &T.one_dev ), 256 ) );
  HIP_TRY( hipMemset( T.one_dev, 0, 256 ) );
  HIP_TRY( hipMalloc( reinterpret_cast<void **>( &T.exited_dev ), sizeof( uint32_t ) * AA_MAX_WORKER_GRIDS ) );
  HIP_TRY( hipMemset( T.exited_dev, 0, sizeof( uint32_t ) * AA_MAX_WORKER_GRIDS ) );
  if ( const char * e = std::getenv( "ALFALFA_AMD_WORKER_LINGER_MS" ) ) T.linger_ticks = static_cast<unsigned long long>( std::max( 0, atoi( e ) ) ) * 100000ull;
  if ( const char * e = std::getenv( "ALFALFA_AMD_TOPUP_DIV" ) ) T.topup_div = std::max( 1, std::min( 64, atoi( e ) ) );
  if ( const char * e = std::getenv( "ALFALFA_AMD_TOKEN_PROFILE" ) ) if ( atoi( e ) ) {
    HIP_TRY( hipMalloc( reinterpret_cast<void **>( &T.prof_dev ), 64 ) );
    HIP_TRY( hipMemset( T.prof_dev, 0, 64 ) );
  }
  HIP_TRY( hipHostMalloc( reinterpret_cast<void **>( &T.retire_host ), sizeof( uint32_t ) * AA_MAX_WORKER_GRIDS, hipHostMallocDefault ) );
  std::memset( T.retire_host, 0, sizeof( uint32_t ) * AA_MAX_WORKER_GRIDS );
  HIP_TRY( hipHostGetDevicePointer( reinterpret_cast<void **>( &T.retire_dev ), T.retire_host, 0 ) );
  HIP_TRY( hipHostMalloc( reinterpret_cast<void **>( &T.mirror_host ), sizeof( aa_tok_mirror ), hipHostMallocDefault ) );
  std::memset( T.mirror_host, 0, sizeof( aa_tok_mirror ) );
  HIP_TRY( hipHostGetDevicePointer( reinterpret_cast<void **>( &T.mirror_dev ), T.mirror_host, 0 ) );
  // the coefficient heap: a block index is 32 bits and a block 32 bytes -> at most 128 GiB; by default at most 3/4 of what the
  // context may use.  Virtual range now, memory as the frames need it.
  const size_t cap = size_t( 120 ) << 30;
  if ( const char * e = std::getenv( "ALFALFA_AMD_HEAP_GROW_MB" ) ) T.grow_bytes = std::max<size_t>( 2, static_cast<size_t>( atoi( e ) ) ) << 20;     // (tests: small pieces)
  if ( const char * e = std::getenv( "ALFALFA_AMD_HEAP_LIMIT_MB" ) ) T.heap_limit = std::max<size_t>( 2, static_cast<size_t>( atoi( e ) ) ) << 20;
  if ( !T.heap_limit ) T.heap_limit = ctx->pool_soft_limit == ~size_t( 0 ) ? ( size_t( 16 ) << 30 ) : ctx->pool_soft_limit / 4 * 3;
  T.heap_limit = std::max( T.grow_bytes, std::min( cap, T.heap_limit ) / T.grow_bytes * T.grow_bytes );
  const char * no_vmm = std::getenv( "ALFALFA_AMD_NO_VMM" );
  if ( !( no_vmm && atoi( no_vmm ) ) ) {
    void * va = nullptr;
    size_t va_align = 0;
    if ( const char * e = std::getenv( "ALFALFA_AMD_HEAP_VA_ALIGN_MB" ) ) va_align = static_cast<size_t>( atoi( e ) ) << 20;     // (experiments)
    if ( hipMemAddressReserve( &va, T.heap_limit, va_align, nullptr, 0 ) == hipSuccess && va ) { T.heap = static_cast<uint8_t *>( va ); T.heap_va = T.heap_limit; T.vmm = true; }
    else (void) hipGetLastError();
  }
  if ( !T.vmm ) {
    // no virtual memory management on this runtime: one fixed piece (a quarter of the limit unless the caller set one)
    size_t fixed = std::getenv( "ALFALFA_AMD_HEAP_LIMIT_MB" ) ? T.heap_limit : std::max( T.grow_bytes, T.heap_limit / 4 / T.grow_bytes * T.grow_bytes );
    HIP_TRY( hipMalloc( reinterpret_cast<void **>( &T.heap ), fixed ) );
    T.heap_va = fixed;
  }
  // (the token lanes advance a pointer into a chunk with a 32-bit add: tok_fsm.hh bump_words)
  if ( reinterpret_cast<uintptr_t>( T.heap ) & ( kChunkBytesHeap - 1 ) ) return fail( AA_ERR_HIP, "the coefficient heap's base is not aligned to a chunk (64 KB)" );
  uint32_t entries = 1;
  while ( size_t( entries ) * kChunkBytesHeap < T.heap_va ) entries <<= 1;
  uint8_t * pr = nullptr;
  HIP_TRY( hipMalloc( reinterpret_cast<void **>( &pr ), 256 + size_t( entries ) * 4 ) );
  HIP_TRY( hipMemset( pr, 0, 256 + size_t( entries ) * 4 ) );
  T.pool = reinterpret_cast<aa::CoeffPool *>( pr ); T.ring = reinterpret_cast<uint32_t *>( pr + 256 );
  { aa::CoeffPool hp {}; hp.mask = entries - 1; HIP_TRY( hipMemcpy( T.pool, &hp, sizeof hp, hipMemcpyHostToDevice ) ); }
  HIP_TRY( hipStreamSynchronize( nullptr ) );          // (every memset above -- null stream -- has landed before a kernel on a non-blocking stream looks)
  if ( !T.vmm ) {
    if ( int e = aa::launch_pool_push_range( heap_of( ctx ), 0, static_cast<uint32_t>( T.heap_va / kChunkBytesHeap ), T.util ) ) return hip_fail( static_cast<hipError_t>( e ), "k_pool_push_range" );
    T.heap_mapped = T.heap_va;
    { std::lock_guard<std::mutex> g( ctx->pool_mu ); ctx->pool_bytes += T.heap_va; }      // (counts against the memory limit like a slab)
  }
  T.ready = true;
  return AA_OK;
}

void tok_free( aa_ctx * ctx )
{
  auto & T = ctx->tok;
  if ( T.retire_host ) for ( int g = 0; g < aa_ctx::Tok::kSlots; g++ ) __atomic_store_n( &T.retire_host[g], 0xFFFFFFFFu, __ATOMIC_RELEASE );   // lingering workgroups leave now
  if ( T.util ) { (void) hipStreamSynchronize( T.util ); }
  for ( auto & sl : T.slot ) if ( sl.st ) { (void) hipStreamSynchronize( sl.st ); (void) hipStreamDestroy( sl.st ); sl.st = nullptr; }
  if ( T.util ) { (void) hipStreamDestroy( T.util ); T.util = nullptr; }
  if ( T.host_up ) { (void) hipStreamSynchronize( T.host_up ); (void) hipStreamDestroy( T.host_up ); T.host_up = nullptr; }
  if ( T.mirror_ev ) { (void) hipEventDestroy( T.mirror_ev ); T.mirror_ev = nullptr; }
  if ( T.vmm ) {
    if ( T.heap_mapped ) (void) hipMemUnmap( T.heap, T.heap_mapped );
    for ( auto h : T.handles ) (void) hipMemRelease( h );
    if ( T.heap ) (void) hipMemAddressFree( T.heap, T.heap_va );
  } else if ( T.heap ) (void) hipFree( T.heap );
  if ( T.pool ) (void) hipFree( T.pool );
  if ( T.q ) (void) hipFree( T.q );
  if ( T.slots ) (void) hipFree( T.slots );
  if ( T.one_dev ) (void) hipFree( T.one_dev );
  if ( T.exited_dev ) (void) hipFree( T.exited_dev );
  if ( T.prof_dev ) (void) hipFree( T.prof_dev );
  if ( T.retire_host ) (void) hipHostFree( T.retire_host );
  if ( T.mirror_host ) (void) hipHostFree( T.mirror_host );
  const bool packed = T.packed, lpp = T.lane_per_partition;
  T = aa_ctx::Tok {};
  T.packed = packed; T.lane_per_partition = lpp;
}

// Launch worker workgroups if jobs are waiting and fewer workgroups are alive than the GPU holds (the mirror must be fresh).
// `after`: the grid starts behind this event (the hand-over of the jobs it is launched for).
aa_status tok_launch_workers( aa_ctx * ctx, hipEvent_t after )
{
  auto & T = ctx->tok;
  const aa_tok_mirror & M = *T.mirror_host;
  const int32_t queued = static_cast<int32_t>( static_cast<uint32_t>( T.jobs_enqueued ) - M.q_head );
  if ( queued <= 0 || T.lanes < 1 ) return AA_OK;
  int alive[aa_ctx::Tok::kSlots], alive_total = 0;
  for ( int g = 0; g < aa_ctx::Tok::kSlots; g++ ) {
    alive[g] = static_cast<int>( T.slot[g].launched - M.exited[g] );
    if ( T.slot[g].queued_behind_retiring && static_cast<int32_t>( M.exited[g] - T.slot[g].retiring_until ) >= 0 ) T.slot[g].queued_behind_retiring = false;
    alive_total += alive[g];
  }
  // as many workgroups as there are jobs waiting, up to what the GPU holds: when lanes are plentiful a frame gets a wave of its
  // own (a wave steps faster the fewer lanes it carries); grids already launched for these jobs -- started or not -- count
  int want = std::min( T.cap_wgs, queued ) - alive_total;
  // ... and not in dribs and drabs: a grid takes a worker stream for as long as its last wave lives, so small top-ups use the
  // streams up.  Waves linger when the queue is empty; a top-up is for when a good part of the GPU's lanes is really gone.
  if ( want <= 0 || ( alive_total > 0 && want * T.topup_div < T.cap_wgs && want < queued ) ) return AA_OK;
  int g = -1;
  for ( int k = 0; k < aa_ctx::Tok::kSlots; k++ ) if ( alive[k] == 0 ) { g = k; break; }
  if ( g < 0 ) {
    // every worker stream still has a grid with workgroups alive (remnants that keep taking jobs would hold their stream for
    // good): the smallest one retires -- its lanes finish the frames they have and take no more -- and the new grid queues behind it
    for ( int k = 0; k < aa_ctx::Tok::kSlots; k++ )
      if ( !T.slot[k].queued_behind_retiring && ( g < 0 || alive[k] < alive[g] ) ) g = k;
    // (a retiring grid's lanes stop taking frames and the new grid starts when the old one's last chain has ended: that costs capacity
    // for up to a key frame's chain -- worth it when half the GPU's lanes are gone, or when the grid that goes is small beside what
    // comes.  Without the second case a long run whose waves leave after a short linger would end up with every worker stream held
    // by a remnant and no way to top up before half the lanes are gone.)
    if ( g < 0 ) return AA_OK;
    const bool half_gone = alive[g] * 4 <= T.cap_wgs && alive_total * 2 <= T.cap_wgs;
    const bool small_remnant = alive[g] * 2 <= want;
    if ( !half_gone && !small_remnant ) return AA_OK;
    __atomic_store_n( &T.retire_host[g], T.slot[g].gen, __ATOMIC_RELEASE );
    T.slot[g].queued_behind_retiring = true; T.slot[g].retiring_until = T.slot[g].launched;
    ctx->stats.worker_retires++;
    want = std::min( T.cap_wgs, want + alive[g] );       // (the remnant's workgroups are gone by the time this grid starts)
  }
  auto & sl = T.slot[g];
  sl.gen++;
  if ( after ) HIP_TRY( hipStreamWaitEvent( sl.st, after, 0 ) );
  LaunchTimer timer( ctx, 4, sl.st );
  if ( int e = aa::launch_token_workers( T.q, T.slots, heap_of( ctx ), T.exited_dev + g, T.retire_dev + g, sl.gen, static_cast<uint32_t>( T.cap_wgs ), T.prof_dev, T.linger_ticks, want, T.lanes, T.lane_bytes, T.lds, T.packed, T.lane_per_partition ? T.mp_hint : 0u, sl.st ) )
    return hip_fail( static_cast<hipError_t>( e ), "k_token_workers" );
  sl.launched += static_cast<uint32_t>( want );
  ctx->stats.worker_launches++; ctx->stats.worker_wgs += static_cast<uint64_t>( want );
  return AA_OK;
}

// What a waiting host does every few milliseconds: look at the device's counters, map more heap when lanes starve, make sure
// workgroups exist for the jobs that wait.
aa_status tok_service( aa_ctx * ctx, hipEvent_t after = nullptr )
{
  auto & T = ctx->tok;
  if ( !T.ready ) return AA_OK;
  if ( aa_status st = tok_peek_mirror( ctx ) ) return st;       // (a look, not a wait: this runs inside every wait for a frame's parse)
  if ( T.mirror_host->pool_starving != T.seen_starving ) {
    T.seen_starving = T.mirror_host->pool_starving;
    if ( aa_status st = tok_grow_heap( ctx, T.heap_mapped + T.grow_bytes ) ) return st;
  }
  return tok_launch_workers( ctx, after );
}

inline double parse_timeout_ms()
{
  static const double ms = [] { const char * e = std::getenv( "ALFALFA_AMD_PARSE_TIMEOUT_S" ); const double s = e ? atof( e ) : 0.0; return ( s > 0 ? s : 300.0 ) * 1000.0; }();
  return ms;
}
// the token lane's `done` word of one frame (pinned host memory the lane writes last)
// (the `done` word is the writer's last store behind a release -- a GPU lane's system-scope fence, a host lane's thread fence: read
// it with acquire, so that what the reader looks at next -- the summary's counts, host_dense -- is what the writer left)
inline bool summary_done( volatile aa::FrameSummary * sum ) { return __atomic_load_n( const_cast<const uint32_t *>( &sum->done ), __ATOMIC_ACQUIRE ) != 0; }
aa_status tok_wait_done( aa_ctx * ctx, volatile aa::FrameSummary * sum )
{
  if ( summary_done( sum ) ) return AA_OK;
  const double t0 = now_ms();
  double last = t0 - 1e9;
  int spins = 0;
  while ( !summary_done( sum ) ) {
    const double t = now_ms();
    if ( t - last > 2.0 ) { if ( aa_status st = tok_service( ctx ) ) return st; last = t; }
    if ( t - t0 > parse_timeout_ms() ) return fail( AA_ERR_HIP, "device parser: a frame handed to the token workers was not finished in time (ALFALFA_AMD_PARSE_TIMEOUT_S, default 300)" );
    if ( ++spins > 64 ) usleep( 50 );
  }
  ctx->stats.parse_wait_ms += now_ms() - t0;
  return AA_OK;
}

// every frame handed to the queue so far is through
bool batch_through( const Batch * b )
{
  const volatile aa::FrameSummary * sums = reinterpret_cast<const volatile aa::FrameSummary *>( b->host + b->summaries_off );
  for ( int i = 0; i < b->n; i++ ) if ( b->items[i].live && b->items[i].s->frames[b->items[i].frame].enqueued && !sums[i].done ) return false;
  return true;
}
aa_status tok_quiesce( aa_ctx * ctx )
{
  auto & T = ctx->tok;
  for ( Batch * b : T.inflight ) {
    volatile aa::FrameSummary * sums = reinterpret_cast<volatile aa::FrameSummary *>( b->host + b->summaries_off );
    for ( int i = 0; i < b->n; i++ )
      if ( b->items[i].live && b->items[i].s->frames[b->items[i].frame].enqueued ) if ( aa_status st = tok_wait_done( ctx, &sums[i] ) ) return st;
  }
  T.inflight.clear();
  return AA_OK;
}
void tok_prune_inflight( aa_ctx * ctx )
{
  auto & v = ctx->tok.inflight;
  if ( v.size() < 64 ) return;
  size_t keep = 0;
  for ( Batch * b : v ) if ( !batch_through( b ) ) v[keep++] = b;
  v.resize( keep );
}

// The slice of LDS a token lane needs depends on the widest frame (its above-row flags) and on whether frames have several DCT
// partitions.  Grids with smaller slices cannot run such frames, so the size only ever grows -- and before it does, the queue
// is drained and the grids are gone.
aa_status tok_set_lane_bytes( aa_ctx * ctx, uint32_t need )
{
  auto & T = ctx->tok;
  if ( need <= T.lane_bytes ) return AA_OK;
  if ( T.lane_bytes ) {
    if ( aa_status st = tok_quiesce( ctx ) ) return st;
    // the grids that linger (idle waves stay for `linger`) are told to leave now: nobody should wait that long for them
    for ( int g = 0; g < aa_ctx::Tok::kSlots; g++ ) {
      auto & sl = T.slot[g];
      __atomic_store_n( &T.retire_host[g], sl.gen, __ATOMIC_RELEASE );
      sl.gen++;                                         // (the next grid of this slot is of a generation the word does not cover)
    }
    for ( auto & sl : T.slot ) HIP_TRY( hipStreamSynchronize( sl.st ) );
    for ( auto & sl : T.slot ) { sl.queued_behind_retiring = false; }
  }
  T.lane_bytes = ( need + 15u ) & ~15u;
  int per_cu = 0;
  aa::token_worker_shape( T.lane_bytes, T.n_cus, &T.lanes, &T.lds, &per_cu );
  if ( T.lanes < 1 || per_cu < 1 ) return fail( AA_ERR_UNSUPPORTED, "device parser: a token lane for frames this wide does not fit a workgroup's LDS" );
  T.cap_wgs = per_cu * T.n_cus;
  return AA_OK;
}

} // namespace

static uint8_t * slot_plane( aa_stream * s, int slot, int plane )
{
  uint8_t * p = s->slots[slot].dev;
  if ( plane >= 1 ) p += s->plane_bytes[0];
  if ( plane >= 2 ) p += s->plane_bytes[1];
  return p;
}

namespace {

// Give a frame's records back (device-parsed: its record block and its share of the batch arena; host-parsed: its share of
// a frame-store chunk, freed when the chunk's last frame goes).  `deferred`: queued kernels may still read them.
void release_records( aa_stream * s, FrameRec & f, bool deferred )
{
  if ( f.records_released ) return;
  aa_ctx * ctx = s->ctx;
  // a token lane may still be writing this frame's records (callers release decoded frames: then this is over already)
  bool parsed = false;
  if ( f.enqueued && f.summary ) parsed = tok_wait_done( ctx, f.summary ) == AA_OK;
  if ( !parsed && f.enqueued && f.summary && f.batch && f.batch_item >= 0 && f.batch_item < static_cast<int>( f.batch->items.size() ) && f.batch->items[f.batch_item].on_host ) {
    // A HOST LANE's frame whose wait failed (the parse timeout, or a HIP error out of the service call): a worker thread may be
    // inside host_lane_run for it right now -- reading the pinned arena, writing host_dense, queueing copies into the record
    // block -- or have it in its queue.  Nothing of the batch may be freed under it (use-after-free, another
    // batch's records silently overwritten).  A host lane ends every path with the `done` word, so wait for that word itself,
    // without a bound and without the device.
    while ( !summary_done( f.summary ) ) usleep( 200 );
  }
  f.records_released = true;
  if ( f.rec_block ) {
    // the coefficient chunks the frame took go back to the pool on the device, by a kernel that reads the list out of the
    // record block: the block itself is recycled only behind that kernel (always through an epoch, never at once)
    const bool has_chunks = parsed && f.chunk_list && !f.chunks_returned;
    // (the list is handed to k_pool_free_lists when the open epoch is closed: mark the epoch as used, or -- with every record
    // block inside its batch arena, no deferred free of its own -- nothing would ever close it and the chunks would stay out)
    if ( has_chunks ) { std::lock_guard<std::mutex> g( ctx->pool_mu ); ctx->tok.pending_lists.push_back( f.chunk_list ); ctx->open_epoch_used = true; f.chunks_returned = true; }
    if ( !f.rec_in_arena ) dev_free( ctx, f.rec_block, f.rec_bytes, deferred || has_chunks );
    f.rec_block = nullptr; f.chunk_list = nullptr; f.packed_pos = nullptr;
    ctx->tok.chunks_committed -= f.est_chunks; f.est_chunks = 0;
  }
  if ( Batch * b = f.batch ) {
    if ( f.batch_item >= 0 && f.batch_item < static_cast<int>( b->items.size() ) ) {
      Batch::Item & bi = b->items[f.batch_item];          // (a host lane's frame: its dense blocks; the wait above saw its `done` word)
      if ( bi.host_dense ) { dev_free( ctx, bi.host_dense, bi.host_dense_bytes, true ); bi.host_dense = nullptr; }
    }
    bool last;
    { std::lock_guard<std::mutex> g( ctx->pool_mu ); last = --b->live == 0; }
    if ( f.batch_item >= 0 && f.batch_item < static_cast<int>( b->items.size() ) ) b->items[f.batch_item].live = false;
    if ( last ) {
      if ( b->tokens_pending ) ctx->deferred.erase( std::remove( ctx->deferred.begin(), ctx->deferred.end(), b ), ctx->deferred.end() );
      ctx->tok.inflight.erase( std::remove( ctx->tok.inflight.begin(), ctx->tok.inflight.end(), b ), ctx->tok.inflight.end() );
      ctx->tok.batches.erase( std::remove( ctx->tok.batches.begin(), ctx->tok.batches.end(), b ), ctx->tok.batches.end() );
      if ( b->hdr_done ) { (void) hipEventSynchronize( b->hdr_done ); (void) hipEventDestroy( b->hdr_done ); }   // never forgotten while kernels still read the arena
      { std::lock_guard<std::mutex> g( ctx->pool_mu ); ctx->pinned_pool.emplace_back( b->host, b->host_bytes ); }
      dev_free( ctx, b->dev, b->dev_bytes, true );        // (always through an epoch: the chunk lists of its frames are read out of it on the device)
      delete b;
    }
    f.batch = nullptr; f.summary = nullptr; f.parse_job = nullptr;
  }
  if ( f.chunk >= 0 && f.chunk < static_cast<int>( s->chunks.size() ) ) {
    Chunk & c = s->chunks[f.chunk];
    const bool is_tail = f.chunk + 1 == static_cast<int>( s->chunks.size() ) && c.host != nullptr;   // still being filled
    if ( --c.live_frames == 0 && !is_tail && c.dev ) {
      if ( c.host ) { std::lock_guard<std::mutex> g( ctx->pool_mu ); ctx->pinned_pool.emplace_back( c.host, c.pinned_bytes ); c.host = nullptr; }
      dev_free( ctx, c.dev, c.dev_bytes, deferred ); c.dev = nullptr;
    }
  }
  f.host_job = nullptr; f.dev_job = nullptr;
}

uint8_t * pinned_get( aa_ctx * ctx, size_t bytes, size_t * got )
{
  {
    std::lock_guard<std::mutex> g( ctx->pool_mu );
    auto & pool = ctx->pinned_pool;
    for ( size_t i = 0; i < pool.size(); i++ )
      if ( pool[i].second >= bytes && pool[i].second <= bytes + bytes / 2 ) {
        uint8_t * p = pool[i].first; *got = pool[i].second;
        pool[i] = pool.back(); pool.pop_back();
        return p;
      }
  }
  uint8_t * p = nullptr;
  if ( hipHostMalloc( reinterpret_cast<void **>( &p ), bytes, hipHostMallocDefault ) != hipSuccess ) return nullptr;
  *got = bytes;
  { std::lock_guard<std::mutex> g( ctx->pool_mu ); ctx->pinned_bytes += bytes; ctx->stats.pinned_allocs++; }
  return p;
}

// Per-frame constants of the reconstruction job record (host copy); the raster planes are bound later (bind_frame).
void fill_job( const FrameRec & rec, aa_dev_frame * job )
{
  const aa_frame_header & h = rec.hdr;
  std::memset( job, 0, sizeof *job );
  std::memcpy( job->quant, h.quant, sizeof job->quant );
  job->mbw = h.mb_width; job->mbh = h.mb_height;
  job->key_frame = h.key_frame; job->loop_filter_level = h.loop_filter_level;
  job->sharpness = h.sharpness_level; job->has_intra = h.has_intra_mb;
}

// References bookkeeping of frame `fi`, done when it is first handed to reconstruction (frames of a stream are submitted in
// order): output slot, the rasters it predicts from, then Frame::copy_to on slot ids (frame.cc:271-307).
aa_status bind_frame( aa_stream * s, int fi, aa_raster_binding * out )
{
  FrameRec & rec = s->frames[fi];
  const aa_frame_header & h = rec.hdr;
  int out_slot;
  if ( aa_status st = alloc_slot( s, &out_slot ) ) return st;
  retain( s, out_slot );   // the frame's own handle (RasterHandle returned to the caller)
  rec.out_slot = out_slot;
  out->job = const_cast<aa_dev_frame *>( rec.dev_job );
  for ( int p = 0; p < 3; p++ ) out->cur[p] = slot_plane( s, out_slot, p );
  for ( int r = 0; r < 3; r++ ) for ( int p = 0; p < 3; p++ ) out->ref[r][p] = slot_plane( s, s->cur_ref_slot[r], p );
  enum { LAST = 0, GOLDEN = 1, ALT = 2 };
  if ( h.key_frame ) { for ( int i = 0; i < 3; i++ ) set_ref( s, i, out_slot, fi ); }
  else {
    if ( h.copy_buffer_to_alternate == 1 ) set_ref( s, ALT, s->cur_ref_slot[LAST], s->cur_ref_frame[LAST] );
    else if ( h.copy_buffer_to_alternate == 2 ) set_ref( s, ALT, s->cur_ref_slot[GOLDEN], s->cur_ref_frame[GOLDEN] );
    if ( h.copy_buffer_to_golden == 1 ) set_ref( s, GOLDEN, s->cur_ref_slot[LAST], s->cur_ref_frame[LAST] );
    else if ( h.copy_buffer_to_golden == 2 ) set_ref( s, GOLDEN, s->cur_ref_slot[ALT], s->cur_ref_frame[ALT] );
    if ( h.refresh_golden ) set_ref( s, GOLDEN, out_slot, fi );
    if ( h.refresh_alternate ) set_ref( s, ALT, out_slot, fi );
    if ( h.refresh_last ) set_ref( s, LAST, out_slot, fi );
  }
  for ( int i = 0; i < 3; i++ ) rec.ref_after[i] = s->cur_ref_frame[i];
  rec.placed = true;
  if ( !rec.handle_held ) release( s, out_slot );   // the caller let go of this frame before it was decoded: only references keep its raster
  return AA_OK;
}

// Bind the rasters of the frames of a decode submission that have none yet: one small upload + one tiny kernel, in front of
// the reconstruction kernels on the compute stream.
aa_status bind_batch( aa_ctx * ctx, aa_stream * const * streams, int n, const int * frame_index )
{
  int need = 0;
  for ( int i = 0; i < n; i++ ) if ( !streams[i]->frames[frame_index[i]].placed ) need++;
  if ( !need ) return AA_OK;
  aa_ctx::BindBuf & bb = ctx->bind_bufs[ctx->next_bind_buf];
  ctx->next_bind_buf = ( ctx->next_bind_buf + 1 ) % aa_ctx::kBindBufs;
  if ( bb.busy ) {
    const auto t0 = std::chrono::steady_clock::now();
    HIP_TRY( hipEventSynchronize( bb.done ) );
    bb.busy = false;
    ctx->stats.bind_wait_ms += std::chrono::duration<double, std::milli>( std::chrono::steady_clock::now() - t0 ).count();
  }
  if ( bb.cap < static_cast<size_t>( need ) ) {
    if ( bb.host ) (void) hipHostFree( bb.host );
    bb.host = nullptr; bb.dev = nullptr; bb.cap = 0;
    const size_t cap = std::max<size_t>( 512, size_t( need ) * 2 );
    // pinned and mapped: the patch kernel reads the bindings over the bus.  (No copy engine on the reconstruction path: a small
    // copy queues behind whatever the engine holds, and that can be a copy ordered behind a seconds-long parse kernel.)
    HIP_TRY( hipHostMalloc( reinterpret_cast<void **>( &bb.host ), cap * sizeof( aa_raster_binding ), hipHostMallocDefault ) );
    HIP_TRY( hipHostGetDevicePointer( reinterpret_cast<void **>( &bb.dev ), bb.host, 0 ) );
    bb.cap = cap;
  }
  if ( !bb.done ) HIP_TRY( hipEventCreateWithFlags( &bb.done, hipEventDisableTiming ) );
  int k = 0;
  for ( int i = 0; i < n; i++ ) {
    if ( streams[i]->frames[frame_index[i]].placed ) continue;
    if ( aa_status st = bind_frame( streams[i], frame_index[i], &bb.host[k] ) ) return st;
    k++;
  }
  if ( int e = aa::launch_bind_rasters( bb.dev, k, ctx->compute ) ) return hip_fail( static_cast<hipError_t>( e ), "k_bind_rasters" );
  HIP_TRY( hipEventRecord( bb.done, ctx->compute ) );
  bb.busy = true;
  return AA_OK;
}

// The persistent segment map has two homes: the host parser's (frames parsed by aa_stream_parse) and dev_segmap (frames
// parsed on the device).  Whoever parsed last owns the current copy; the other side is refreshed on demand.
aa_status segmap_to_host( aa_stream * s )
{
  if ( !s->segmap_on_device ) return AA_OK;
  for ( auto ps : s->ctx->parse_streams ) HIP_TRY( hipStreamSynchronize( ps ) );
  std::vector<uint8_t> & m = s->parser.segment_map();
  HIP_TRY( hipMemcpy( m.data(), s->dev_segmap, m.size(), hipMemcpyDeviceToHost ) );
  s->segmap_on_device = false;
  return AA_OK;
}

} // namespace

extern "C" {

const char * aa_last_error( void ) { return g_last_error.c_str(); }
int aa_host_cpus( void ) { return effective_cpus(); }
int aa_runtime_prepare( void )
{
  // (overwrite = 0: a value the host program or its user chose stands)
  const bool was_set = std::getenv( "GPU_MAX_HW_QUEUES" ) != nullptr;
  setenv( "GPU_MAX_HW_QUEUES", "16", 0 );
  return was_set ? 1 : 0;
}
int aa_abi_version( void ) { return AA_ABI_VERSION; }
int aa_device_count( void ) { int n = 0; if ( hipGetDeviceCount( &n ) != hipSuccess ) return 0; return n; }

void aa_raster_geometry( uint16_t width, uint16_t height, uint32_t * pw, uint32_t * ph )
{
  if ( pw ) *pw = 16u * ( ( width + 15u ) / 16u );
  if ( ph ) *ph = 16u * ( ( height + 15u ) / 16u );
}

/* ---------------- parser ---------------- */
aa_status aa_parser_create( uint16_t width, uint16_t height, aa_parser ** out )
{
  if ( !out || !width || !height ) return fail( AA_ERR_ARGUMENT, "aa_parser_create: bad argument" );
  *out = new ( std::nothrow ) aa_parser( width, height );
  return *out ? AA_OK : fail( AA_ERR_ARGUMENT, "out of memory" );
}
void aa_parser_destroy( aa_parser * p ) { delete p; }

aa_status aa_parser_parse( aa_parser * p, const uint8_t * data, size_t size, aa_frame_header * hdr, aa_mb_info * mb, int16_t * coeff )
{
  if ( !p || !data || !hdr || !mb || !coeff ) return fail( AA_ERR_ARGUMENT, "aa_parser_parse: null argument" );
  try { p->impl.parse( data, size, *hdr, mb, coeff ); }
  catch ( const aa::ParseError & e ) { return fail( e.code, e.message ); }
  return AA_OK;
}
aa_status aa_parse_frame_tag( const uint8_t * data, size_t size, uint16_t width, uint16_t height, int accept_partial,
                              int * key_frame, int * show_frame, int * experimental, int * corruption_level )
{
  if ( !data && size ) return fail( AA_ERR_ARGUMENT, "aa_parse_frame_tag: null data" );
  try {
    const aa::FrameTag t = aa::parse_frame_tag( data, size, width, height, accept_partial != 0 );
    if ( key_frame ) *key_frame = t.key; if ( show_frame ) *show_frame = t.show;
    if ( experimental ) *experimental = t.experimental; if ( corruption_level ) *corruption_level = t.corruption;
  } catch ( const aa::ParseError & e ) { return fail( e.code, e.message ); }
  return AA_OK;
}
aa_status aa_parser_set_error_concealment( aa_parser * p, int on )
{
  if ( !p ) return fail( AA_ERR_ARGUMENT, "null parser" );
  p->impl.set_error_concealment( on != 0 );
  return AA_OK;
}
static void export_probs( const aa::Parser & ps, uint8_t * out )
{
  const aa::ProbTables & t = ps.probs();
  std::memcpy( out, t.coeff, 1056 ); std::memcpy( out + 1056, t.y_mode, 4 );
  std::memcpy( out + 1060, t.uv_mode, 3 ); std::memcpy( out + 1063, t.mv, 38 );
}
aa_status aa_parser_get_probs( const aa_parser * p, uint8_t probs[1101] )
{
  if ( !p || !probs ) return fail( AA_ERR_ARGUMENT, "null argument" );
  export_probs( p->impl, probs ); return AA_OK;
}
aa_status aa_parser_get_segmentation( const aa_parser * p, int * enabled, int * absolute, int8_t quant[4], int8_t lf[4], uint8_t * map )
{
  if ( !p ) return fail( AA_ERR_ARGUMENT, "null argument" );
  const aa::SegmentationState & s = p->impl.segmentation();
  if ( enabled ) *enabled = s.enabled;
  if ( absolute ) *absolute = s.absolute;
  if ( quant ) std::memcpy( quant, s.quant, 4 );
  if ( lf ) std::memcpy( lf, s.lf, 4 );
  if ( map ) std::memcpy( map, s.map.data(), s.map.size() );
  return AA_OK;
}
aa_status aa_parser_get_filter_adjustments( const aa_parser * p, int * enabled, int8_t ref[4], int8_t mode[4] )
{
  if ( !p ) return fail( AA_ERR_ARGUMENT, "null argument" );
  const aa::FilterAdjustState & f = p->impl.filter_adjustments();
  if ( enabled ) *enabled = f.enabled;
  if ( ref ) std::memcpy( ref, f.ref, 4 );
  if ( mode ) std::memcpy( mode, f.mode, 4 );
  return AA_OK;
}

static aa_status export_state_common( const aa::Parser & ps, uint8_t * buf, size_t capacity )
{
  if ( !buf || capacity < ps.state_size() ) return fail( AA_ERR_ARGUMENT, "export_state: buffer too small" );
  ps.export_state( buf ); return AA_OK;
}
static aa_status import_state_common( aa::Parser & ps, const uint8_t * buf, size_t size )
{
  if ( !buf ) return fail( AA_ERR_ARGUMENT, "import_state: null buffer" );
  try { ps.import_state( buf, size ); } catch ( const aa::ParseError & e ) { return fail( e.code, e.message ); }
  return AA_OK;
}
size_t aa_parser_state_size( const aa_parser * p ) { return p ? p->impl.state_size() : 0; }
aa_status aa_parser_export_state( const aa_parser * p, uint8_t * buf, size_t capacity )
{ return p ? export_state_common( p->impl, buf, capacity ) : fail( AA_ERR_ARGUMENT, "null parser" ); }
aa_status aa_parser_import_state( aa_parser * p, const uint8_t * buf, size_t size )
{ return p ? import_state_common( p->impl, buf, size ) : fail( AA_ERR_ARGUMENT, "null parser" ); }

/* DecoderState in the reference's wire format (DecoderState::serialize, decoder.cc:283-313) */
static aa_status serialize_out( const std::vector<uint8_t> & blob, uint8_t * buf, size_t capacity, size_t * size )
{
  if ( size ) *size = blob.size();
  if ( !buf ) return AA_OK;                         // size query
  if ( capacity < blob.size() ) return fail( AA_ERR_ARGUMENT, "serialize: buffer too small" );
  std::memcpy( buf, blob.data(), blob.size() );
  return AA_OK;
}
aa_status aa_parser_serialize_state( const aa_parser * p, uint8_t * buf, size_t capacity, size_t * size )
{
  if ( !p ) return fail( AA_ERR_ARGUMENT, "null parser" );
  std::vector<uint8_t> blob;
  p->impl.serialize_reference( blob );
  return serialize_out( blob, buf, capacity, size );
}
aa_status aa_parser_deserialize_state( aa_parser * p, const uint8_t * buf, size_t size )
{
  if ( !p || !buf ) return fail( AA_ERR_ARGUMENT, "null argument" );
  aa::Parser trial = p->impl;                        // the parser changes only if the whole blob is good
  try { if ( trial.deserialize_reference( buf, size ) != size ) return fail( AA_ERR_INVALID, "invalid decoder state: trailing bytes" ); }
  catch ( const aa::ParseError & e ) { return fail( e.code, e.message ); }
  p->impl = trial;
  return AA_OK;
}

/* ---------------- context ---------------- */
aa_status aa_ctx_create( int device, aa_ctx ** out )
{
  if ( !out ) return fail( AA_ERR_ARGUMENT, "aa_ctx_create: null out" );
  int n = 0;
  hipError_t e = hipGetDeviceCount( &n );
  if ( e != hipSuccess || n == 0 )
    return fail( AA_ERR_NO_DEVICE, "no HIP device visible: the decode path has no CPU fallback" );
  if ( device < 0 || device >= n ) return fail( AA_ERR_ARGUMENT, "device index out of range" );
  std::unique_ptr<aa_ctx> ctx( new aa_ctx );
  ctx->device = device;
  HIP_TRY( hipSetDevice( device ) );
  {
    // reconstruction is short and latency-critical, the entropy decode long and patient: the compute stream gets the
    // highest priority the device offers, the parse streams the lowest
    int lo = 0, hi = 0;
    HIP_TRY( hipDeviceGetStreamPriorityRange( &lo, &hi ) );
    ctx->prio_low = lo;
    HIP_TRY( hipStreamCreateWithPriority( &ctx->compute, hipStreamNonBlocking, hi ) );
  }
  HIP_TRY( hipStreamCreateWithFlags( &ctx->copy, hipStreamNonBlocking ) );
  HIP_TRY( hipEventCreateWithFlags( &ctx->upload_done, hipEventDisableTiming ) );
  {
    size_t free_b = 0, total_b = 0;
    if ( hipMemGetInfo( &free_b, &total_b ) == hipSuccess && free_b ) ctx->pool_soft_limit = free_b / 8 * 7;
    (void) hipGetLastError();
  }
  if ( const char * e = getenv( "ALFALFA_AMD_PARSE_STREAMS" ) ) ctx->n_parse_streams = std::max( 1, std::min( aa_ctx::kMaxParseStreams, atoi( e ) ) );
  ctx->parse_streams.assign( ctx->n_parse_streams, nullptr );
  ctx->parse_idle.assign( ctx->n_parse_streams, nullptr );
  for ( auto & ps : ctx->parse_streams ) HIP_TRY( hipStreamCreateWithPriority( &ps, hipStreamNonBlocking, ctx->prio_low ) );
  for ( auto & e : ctx->parse_idle ) HIP_TRY( hipEventCreateWithFlags( &e, hipEventDisableTiming ) );
  if ( const char * e = std::getenv( "ALFALFA_AMD_SCHEDULE" ) ) ctx->schedule = std::string( e ) == "diagonal" ? 1 : 0;
  if ( const char * e = std::getenv( "ALFALFA_AMD_PACKED" ) ) ctx->tok.packed = atoi( e ) != 0;
  if ( const char * e = std::getenv( "ALFALFA_AMD_LANE_PER_PARTITION" ) ) ctx->tok.lane_per_partition = atoi( e ) != 0;
  if ( const char * e = std::getenv( "ALFALFA_AMD_HOST_SHARE_MS" ) ) ctx->host_share_ms = std::max( 0.0, atof( e ) );
  // The row-pipelined kernels keep every unit on one XCD (per-XCD ticket queues indexed by the hardware XCC_ID): find
  // out which XCC ids workgroups of this device really land on.  They must be 0..n-1, each reached by a modest grid.
  {
    int * d = nullptr; int h[AA_MAX_XCD] = {};
    HIP_TRY( hipMalloc( reinterpret_cast<void **>( &d ), sizeof h ) );
    // (on the probe's own stream: a plain hipMemset is not ordered against a non-blocking stream, and under load -- other contexts'
    // worker grids resident -- it was seen to land AFTER the probe kernel: "XCD probe kernel did not run")
    HIP_TRY( hipMemsetAsync( d, 0, sizeof h, ctx->compute ) );
    const int e2 = aa::launch_probe_xcds( d, 2048, ctx->compute );
    if ( e2 ) { (void) hipFree( d ); return hip_fail( static_cast<hipError_t>( e2 ), "k_probe_xcds" ); }
    HIP_TRY( hipStreamSynchronize( ctx->compute ) );
    HIP_TRY( hipMemcpy( h, d, sizeof h, hipMemcpyDeviceToHost ) );
    (void) hipFree( d );
    int n_xcd = 0;
    while ( n_xcd < AA_MAX_XCD && h[n_xcd] > 0 ) n_xcd++;
    for ( int x = n_xcd; x < AA_MAX_XCD; x++ ) if ( h[x] ) return fail( AA_ERR_HIP, "unexpected XCC id layout on this device (ids are not 0..n-1)" );
    if ( n_xcd == 0 ) return fail( AA_ERR_HIP, "XCD probe kernel did not run" );
    for ( int x = 0; x < n_xcd; x++ ) ctx->xcd_share[x] = h[x];
    ctx->n_xcd = n_xcd;
    // The row-pipelined schedule wants every XCD to get its share of a launch's workgroups (the observed round-robin).  If this
    // device spreads them unevenly (CU masks, partition modes), fall back to the schedule whose ordering is kernel boundaries.
    for ( int x = 0; x < n_xcd; x++ ) if ( h[x] * n_xcd * 2 < 2048 ) ctx->schedule = 1;
  }
  *out = ctx.release();
  return AA_OK;
}
static void ctx_free( aa_ctx * ctx );
void aa_ctx_destroy( aa_ctx * ctx )
{
  if ( !ctx ) return;
  if ( --ctx->refs == 0 ) ctx_free( ctx );
}
static void ctx_free( aa_ctx * ctx )
{
  (void) hipSetDevice( ctx->device );
  (void) tok_quiesce( ctx );
  host_lanes_stop( ctx );
  (void) hipStreamSynchronize( ctx->compute ); (void) hipStreamSynchronize( ctx->copy );
  for ( auto ps : ctx->parse_streams ) if ( ps ) { (void) hipStreamSynchronize( ps ); (void) hipStreamDestroy( ps ); }
  for ( auto e : ctx->parse_idle ) if ( e ) (void) hipEventDestroy( e );
  for ( auto & ee : ctx->epoch_events ) { if ( ee.compute ) (void) hipEventDestroy( ee.compute ); if ( ee.copy ) (void) hipEventDestroy( ee.copy ); }
  if ( ctx->last_seg_batch ) (void) hipEventDestroy( ctx->last_seg_batch );
  drain_profile( ctx );
  for ( auto e : ctx->free_events ) (void) hipEventDestroy( e );
  (void) hipEventDestroy( ctx->upload_done );
  if ( ctx->last_raster_download ) (void) hipEventDestroy( ctx->last_raster_download );
  for ( auto & bb : ctx->bind_bufs ) { if ( bb.host ) (void) hipHostFree( bb.host ); if ( bb.done ) (void) hipEventDestroy( bb.done ); }
  for ( auto & gb : ctx->gather_bufs ) { if ( gb.host ) (void) hipHostFree( gb.host ); if ( gb.done ) (void) hipEventDestroy( gb.done ); }
  if ( ctx->ws ) (void) hipFree( ctx->ws );
  if ( ctx->boundary ) (void) hipFree( ctx->boundary );
  tok_free( ctx );
  for ( auto & pc : ctx->pinned_pool ) (void) hipHostFree( pc.first );
  for ( uint8_t * slab : ctx->dev_slabs ) (void) hipFree( slab );
  (void) hipStreamDestroy( ctx->compute ); (void) hipStreamDestroy( ctx->copy );
  delete ctx;
}
static aa_status check_watchdog( aa_ctx * ctx )
{
  if ( !ctx->ws ) return AA_OK;
  int err = 0;
  HIP_TRY( hipMemcpy( &err, &ctx->ws->error, sizeof err, hipMemcpyDeviceToHost ) );
  if ( err == 3 ) return fail( AA_ERR_HIP, "row-pipelined kernel: a workgroup ran on an XCD outside the probed set (output is not valid)" );
  if ( err == 5 ) {
    int hdr[4] = {};
    (void) hipMemcpy( hdr, ctx->ws, sizeof hdr, hipMemcpyDeviceToHost );
    return fail( AA_ERR_HIP, "row-pipelined kernel: XCD " + std::to_string( hdr[1] ) + " handed out " + std::to_string( hdr[2] ) + " of its " + std::to_string( hdr[3] )
                               + " macroblock rows: no workgroup of the launch ran there (output is not valid); use the diagonal schedule on this device" );
  }
  if ( err == 4 ) {
    int hdr[4] = {};
    (void) hipMemcpy( hdr, ctx->ws, sizeof hdr, hipMemcpyDeviceToHost );
    return fail( AA_ERR_HIP, "row-pipelined kernel: a wave of unit " + std::to_string( hdr[1] ) + " row " + std::to_string( hdr[2] ) + " moved from XCD "
                               + std::to_string( hdr[3] >> 16 ) + " to XCD " + std::to_string( hdr[3] & 0xFFFF ) + " (context save/restore under queue oversubscription?): "
                               "the in-launch hand-off is only coherent inside one XCD (output is not valid)" );
  }
  if ( err ) {
    int hdr[4 + AA_MAX_XCD] = {};
    (void) hipMemcpy( hdr, ctx->ws, sizeof hdr, hipMemcpyDeviceToHost );
    std::string tickets;
    for ( int x = 0; x < ctx->n_xcd; x++ ) tickets += ( x ? "," : "" ) + std::to_string( hdr[4 + x] );
    return fail( AA_ERR_HIP, std::string( "row-pipelined kernel " ) + ( err == 1 ? "k_recon_intra4" : "k_loopfilter_rows4" )
                               + ": a bounded wait for the macroblock row above expired (output is not valid); first at unit " + std::to_string( hdr[1] )
                               + " row " + std::to_string( hdr[2] ) + ", needed " + std::to_string( hdr[3] >> 16 ) + " saw " + std::to_string( hdr[3] & 0xFFFF )
                               + "; tickets handed out per XCD (last launch): " + tickets );
  }
  return AA_OK;
}
aa_status aa_ctx_sync( aa_ctx * ctx )
{
  if ( !ctx ) return fail( AA_ERR_ARGUMENT, "null ctx" );
  if ( aa_status st = set_device( ctx ) ) return st;
  HIP_TRY( hipStreamSynchronize( ctx->copy ) );
  for ( auto ps : ctx->parse_streams ) HIP_TRY( hipStreamSynchronize( ps ) );
  if ( aa_status st = tok_quiesce( ctx ) ) return st;       // every frame handed to the token workers is parsed
  { std::lock_guard<std::mutex> g( ctx->pool_mu ); flush_chunk_frees( ctx ); }     // coefficient chunks of released frames are back in the pool
  HIP_TRY( hipStreamSynchronize( ctx->compute ) );
  return check_watchdog( ctx );
}
aa_status aa_ctx_memory( aa_ctx * ctx, size_t * free_bytes, size_t * total_bytes )
{
  if ( !ctx ) return fail( AA_ERR_ARGUMENT, "null ctx" );
  if ( aa_status st = set_device( ctx ) ) return st;
  size_t f = 0, t = 0;
  HIP_TRY( hipMemGetInfo( &f, &t ) );
  if ( free_bytes ) *free_bytes = f;
  if ( total_bytes ) *total_bytes = t;
  return AA_OK;
}
aa_status aa_ctx_set_memory_limit( aa_ctx * ctx, size_t bytes )
{
  if ( !ctx || !bytes ) return fail( AA_ERR_ARGUMENT, "aa_ctx_set_memory_limit: bad argument" );
  std::lock_guard<std::mutex> g( ctx->pool_mu );
  ctx->pool_soft_limit = bytes;
  if ( !ctx->tok.ready ) ctx->tok.heap_limit = 0;      // (the heap's virtual size follows the limit when it is set up)
  return AA_OK;
}
aa_status aa_ctx_get_info( aa_ctx * ctx, aa_ctx_info * out )
{
  if ( !ctx || !out ) return fail( AA_ERR_ARGUMENT, "null argument" );
  if ( aa_status st = set_device( ctx ) ) return st;
  std::memset( out, 0, sizeof *out );
  auto & T = ctx->tok;
  {
    std::lock_guard<std::mutex> g( ctx->pool_mu );
    collect_pending( ctx, false );        // (pieces whose release epoch has fired are free, not pending: a caller that plans by these figures should see that)
    out->memory_limit_bytes = ctx->pool_soft_limit; out->pool_bytes = ctx->pool_bytes; out->pinned_host_bytes = ctx->pinned_bytes;
    for ( auto & kv : ctx->dev_free ) out->pool_free_bytes += kv.first * kv.second.size();
    for ( auto & pf : ctx->pending_free ) out->pool_pending_bytes += pf.bytes;
    if ( ctx->cur_slab ) out->pool_free_bytes += kSlabBytes - ctx->slab_used;
    out->pool_free_bytes += ctx->compute_free_bytes;
    for ( auto & h : ctx->compute_hold ) out->pool_pending_bytes += h.second;
  }
  out->heap_mapped_bytes = T.heap_mapped; out->heap_limit_bytes = T.heap_va;
  out->heap_used_bytes = static_cast<uint64_t>( std::max<int64_t>( 0, T.chunks_committed ) ) * kChunkBytesHeap;
  out->heap_is_virtual = T.vmm ? 1u : 0u;
  out->packed_coefficients = T.packed ? 1u : 0u;
  out->lane_per_partition = T.lane_per_partition ? 1u : 0u;
  out->token_lanes_per_workgroup = static_cast<uint32_t>( T.lanes ); out->token_workgroups_capacity = static_cast<uint32_t>( T.cap_wgs );
  out->token_lane_lds_bytes = T.lane_bytes; out->token_workgroup_lds_bytes = T.lds;
  out->compute_units = static_cast<uint32_t>( T.n_cus );
  out->host_share_ms = static_cast<uint32_t>( ctx->host_share_ms + 0.5 );
  out->host_rate_kb_per_ms = ctx->host_lanes.parse_us.load() ? static_cast<uint32_t>( host_lanes_rate( ctx ) / 1e3 + 0.5 ) : 0u;
  out->stream_concurrency = ctx->stream_concurrency; out->streams_needed = ctx->streams_needed;
  out->host_waited_parse_ms = static_cast<uint32_t>( ctx->stats.parse_wait_ms ); out->host_waited_compute_ms = static_cast<uint32_t>( ctx->stats.bind_wait_ms );
  // (asked once: the attribute query goes to the driver and was seen to take ~100 ms beside a busy GPU -- three looks at the books per
  // step of a pipelining caller were 300 ms of its step)
  if ( !ctx->clock_mhz ) { int khz = 0; if ( hipDeviceGetAttribute( &khz, hipDeviceAttributeClockRate, ctx->device ) == hipSuccess ) ctx->clock_mhz = static_cast<uint32_t>( khz / 1000 ); else (void) hipGetLastError(); }
  out->clock_mhz = ctx->clock_mhz;
  if ( T.ready ) {
    if ( aa_status st = tok_peek_mirror( ctx ) ) return st;
    out->heap_free_chunks = T.mirror_host->pool_avail; out->lanes_starved = T.mirror_host->pool_starving;
    for ( int k = 0; k < 8; k++ ) out->token_profile[k] = T.mirror_host->prof[k];
    uint32_t alive = 0;
    for ( int g = 0; g < aa_ctx::Tok::kSlots; g++ ) alive += T.slot[g].launched - T.mirror_host->exited[g];
    out->token_workgroups_alive = alive;
    const int32_t waiting = static_cast<int32_t>( static_cast<uint32_t>( T.jobs_enqueued ) - T.mirror_host->q_head );
    out->jobs_waiting = waiting > 0 ? static_cast<uint32_t>( waiting ) : 0u;
  }
  return AA_OK;
}
aa_status aa_ctx_set_lane_per_partition( aa_ctx * ctx, int on )
{
  if ( !ctx ) return fail( AA_ERR_ARGUMENT, "null context" );
  if ( ctx->tok.ready && ctx->tok.lane_per_partition != ( on != 0 ) ) return fail( AA_ERR_LOGIC, "aa_ctx_set_lane_per_partition: frames have been submitted to this context already" );
  ctx->tok.lane_per_partition = on != 0;
  return AA_OK;
}
aa_status aa_ctx_set_host_share_ms( aa_ctx * ctx, double ms )
{
  if ( !ctx || !( ms >= 0 ) ) return fail( AA_ERR_ARGUMENT, "aa_ctx_set_host_share_ms: bad argument" );
  ctx->host_share_ms = ms;
  return AA_OK;
}
aa_status aa_ctx_set_packed_coefficients( aa_ctx * ctx, int on )
{
  if ( !ctx ) return fail( AA_ERR_ARGUMENT, "null context" );
  if ( ctx->tok.ready && ctx->tok.packed != ( on != 0 ) ) return fail( AA_ERR_LOGIC, "aa_ctx_set_packed_coefficients: frames have been submitted to this context already" );
  ctx->tok.packed = on != 0;
  return AA_OK;
}
/* The sticky error word of the row-pipelined kernels (a bounded wait expired, a wave found itself on another XCD, a queue
 * was not drained) makes every later row-pipelined launch give up early.  Once it has been reported, this clears it. */
aa_status aa_ctx_clear_error( aa_ctx * ctx )
{
  if ( !ctx ) return fail( AA_ERR_ARGUMENT, "null ctx" );
  if ( aa_status st = set_device( ctx ) ) return st;
  HIP_TRY( hipStreamSynchronize( ctx->compute ) );
  if ( ctx->ws ) { HIP_TRY( hipMemsetAsync( ctx->ws, 0, AA_SYNC_WS_ZERO_FROM, ctx->compute ) ); HIP_TRY( hipStreamSynchronize( ctx->compute ) ); }
  return AA_OK;
}
aa_status aa_ctx_set_schedule( aa_ctx * ctx, int schedule )
{
  if ( !ctx || ( schedule != AA_SCHEDULE_ROWS && schedule != AA_SCHEDULE_DIAGONAL ) ) return fail( AA_ERR_ARGUMENT, "aa_ctx_set_schedule: bad argument" );
  ctx->schedule = schedule;
  return AA_OK;
}
void * aa_ctx_compute_stream( aa_ctx * ctx ) { return ctx ? ctx->compute : nullptr; }
void * aa_ctx_copy_stream( aa_ctx * ctx ) { return ctx ? ctx->copy : nullptr; }
aa_status aa_ctx_profile( aa_ctx * ctx, int enable )
{
  if ( !ctx ) return fail( AA_ERR_ARGUMENT, "null ctx" );
  if ( !enable ) { (void) hipStreamSynchronize( ctx->compute ); for ( auto ps : ctx->parse_streams ) (void) hipStreamSynchronize( ps ); drain_profile( ctx ); }
  ctx->profile = enable != 0;
  return AA_OK;
}
aa_status aa_ctx_kernel_stats( aa_ctx * ctx, aa_kernel_stats * out, int reset )
{
  if ( !ctx || !out ) return fail( AA_ERR_ARGUMENT, "null argument" );
  HIP_TRY( hipStreamSynchronize( ctx->compute ) );
  for ( auto ps : ctx->parse_streams ) HIP_TRY( hipStreamSynchronize( ps ) );
  drain_profile( ctx );
  ctx->stats.heap_mapped_bytes = ctx->tok.heap_mapped;
  // (the host lanes' summed parse time comes from their atomic counter: the workers themselves never write to `stats`)
  const uint64_t lanes_us = ctx->host_lanes.parse_us.load();
  ctx->stats.host_batch_parse_cpu_ms = static_cast<double>( lanes_us - ctx->host_lanes.parse_us_mark ) / 1e3;
  *out = ctx->stats;
  if ( reset ) { ctx->stats = aa_kernel_stats {}; ctx->host_lanes.parse_us_mark = lanes_us; }
  return AA_OK;
}

/* ---------------- stream ---------------- */
aa_status aa_stream_create( aa_ctx * ctx, uint16_t width, uint16_t height, aa_stream ** out )
{
  if ( !ctx || !out || !width || !height ) return fail( AA_ERR_ARGUMENT, "aa_stream_create: bad argument" );
  if ( aa_status st = set_device( ctx ) ) return st;
  std::unique_ptr<aa_stream> s( new aa_stream( ctx, width, height ) );
  aa_raster_geometry( width, height, &s->pw, &s->ph );
  s->plane_bytes[0] = size_t( s->pw ) * s->ph;
  s->plane_bytes[1] = s->plane_bytes[2] = size_t( s->pw / 2 ) * ( s->ph / 2 );
  s->slot_bytes = align_up( s->plane_bytes[0] + 2 * s->plane_bytes[1] );
  // References(width, height): all three references alias one (blank) raster (decoder.cc:161-169).  Rasters are never written
  // once they are references, so every decoder of this raster size points at the SAME blank one: a decoder that is created
  // long before its first key frame is decoded (a chunk waiting in a pipeline) costs no raster.
  uint8_t * blank = nullptr;
  {
    // (decoders are created from many threads at once; the raster is zero before anybody -- compute or copy stream -- can read it)
    std::lock_guard<std::mutex> g( ctx->blank_mu );
    auto it = ctx->blank.find( s->slot_bytes );
    if ( it != ctx->blank.end() ) blank = it->second;
    else {
      if ( aa_status st = dev_alloc( ctx, s->slot_bytes, &blank ) ) return st;
      hipError_t e = hipMemsetAsync( blank, 0, s->slot_bytes, ctx->compute );
      if ( e == hipSuccess ) e = hipStreamSynchronize( ctx->compute );
      if ( e != hipSuccess ) { dev_free( ctx, blank, s->slot_bytes ); return hip_fail( e, "hipMemsetAsync (blank reference raster)" ); }
      ctx->blank[s->slot_bytes] = blank;
    }
  }
  Slot sl; sl.dev = blank; sl.shared = true;
  s->slots.push_back( sl );
  const int slot = 0;
  for ( int i = 0; i < 3; i++ ) { s->cur_ref_slot[i] = -1; s->cur_ref_frame[i] = -1; }
  for ( int i = 0; i < 3; i++ ) set_ref( s.get(), i, slot, -1 );
  ctx->refs++;
  *out = s.release();
  return AA_OK;
}
void aa_stream_destroy( aa_stream * s )
{
  if ( !s ) return;
  aa_ctx * ctx = s->ctx;
  (void) hipSetDevice( ctx->device );
  // Nothing here waits for the reconstruction stream: what queued kernels may still read or write (rasters, records) goes back
  // to the pools through a release epoch, like everything released while the pipeline runs -- a caller that drops a decoder
  // per chunk does not stall.  What IS waited for: uploads out of this stream's pinned staging (it is handed to other parses),
  // token lanes still on its frames (release_records), header kernels of a two-phase batch that was never launched.
  bool staged = false;
  for ( auto & c : s->chunks ) if ( c.host && c.uploaded ) staged = true;
  if ( staged ) (void) hipStreamSynchronize( ctx->copy );
  for ( auto & f : s->frames ) if ( f.batch && f.batch->hdr_done && !f.enqueued && !f.records_released ) (void) hipEventSynchronize( f.batch->hdr_done );
  for ( auto & f : s->frames ) release_records( s, f, true );
  for ( auto & c : s->chunks ) {
    if ( c.host ) { std::lock_guard<std::mutex> g( ctx->pool_mu ); ctx->pinned_pool.emplace_back( c.host, c.pinned_bytes ); c.host = nullptr; }
    if ( c.dev ) { dev_free( ctx, c.dev, c.dev_bytes, true ); c.dev = nullptr; }
  }
  for ( auto & sl : s->slots ) if ( sl.dev && !sl.shared ) dev_free_compute( ctx, sl.dev, s->slot_bytes );
  dev_free( ctx, s->dev_segmap, size_t( s->parser.mb_width() ) * s->parser.mb_height(), true );
  delete s;
  if ( --ctx->refs == 0 ) ctx_free( ctx );
}

} // extern "C"
namespace {
// A frame whose records are produced on the HOST -- by the bitstream parser (aa_stream_parse) or handed in as records
// (aa_stream_append_records) -- is staged in the stream's pinned chunk (mirrored 1:1 in HBM, uploaded by aa_stream_upload):
// job record | macroblock records | intra row masks | coefficient blocks.  `fill` writes the header, the macroblock records and
// the coefficient blocks into the staging area it is given (worst-case sized) or throws aa::ParseError.
template <class Fill>
aa_status append_host_frame( aa_stream * s, int * frame_index, aa_frame_header * hdr_out, Fill && fill )
{
  if ( aa_status st = set_device( s->ctx ) ) return st;
  const size_t nmb = size_t( s->parser.mb_width() ) * s->parser.mb_height();
  const size_t job_bytes = align_up( sizeof( aa_dev_frame ) );
  const size_t mb_bytes = align_up( nmb * sizeof( aa_mb_info ) );
  const size_t words_per_row = ( s->parser.mb_width() + 63 ) / 64;
  const size_t rows_bytes = align_up( words_per_row * s->parser.mb_height() * sizeof( unsigned long long ) );
  const size_t worst = job_bytes + mb_bytes + rows_bytes + align_up( nmb * 25 * 32 );
  Chunk * c;
  if ( aa_status st = reserve( s, worst, &c ) ) return st;
  const size_t off = c->used;
  aa_dev_frame * job = reinterpret_cast<aa_dev_frame *>( c->host + off );
  aa_mb_info * mbs = reinterpret_cast<aa_mb_info *>( c->host + off + job_bytes );
  unsigned long long * intra_rows = reinterpret_cast<unsigned long long *>( c->host + off + job_bytes + mb_bytes );
  int16_t * coeffs = reinterpret_cast<int16_t *>( c->host + off + job_bytes + mb_bytes + rows_bytes );

  FrameRec rec;
  try { fill( rec.hdr, mbs, coeffs ); }
  catch ( const aa::ParseError & e ) { return fail( e.code, e.message ); }
  const aa_frame_header & h = rec.hdr;
  c->used = off + job_bytes + mb_bytes + rows_bytes + align_up( size_t( h.num_coeff_blocks ) * 32 );   // commit what was used

  // which 2:1 anti-diagonals hold intra MBs (launch schedule of k_recon_intra)
  const int mbw = h.mb_width, mbh = h.mb_height;
  rec.intra_diagonals.assign( mbw + 2 * ( mbh - 1 ), 0 );
  std::memset( intra_rows, 0, words_per_row * mbh * sizeof( unsigned long long ) );
  if ( h.has_intra_mb )
    for ( int r = 0; r < mbh; r++ ) for ( int col = 0; col < mbw; col++ )
      if ( !( mbs[r * mbw + col].flags & AA_MB_INTER ) ) {
        rec.intra_diagonals[col + 2 * r] = 1;
        intra_rows[r * words_per_row + ( col >> 6 )] |= 1ull << ( col & 63 );
      }

  if ( !h.key_frame )
    for ( size_t i = 0; i < nmb; i++ ) if ( mbs[i].y_mode == 9 /* SPLITMV */ ) { rec.has_split = true; break; }

  fill_job( rec, job );
  job->mbs = reinterpret_cast<const aa_mb_info *>( c->dev + off + job_bytes );
  job->intra_rows = reinterpret_cast<const unsigned long long *>( c->dev + off + job_bytes + mb_bytes );
  job->coeffs = reinterpret_cast<const int16_t *>( c->dev + off + job_bytes + mb_bytes + rows_bytes );
  rec.host_job = job;
  rec.dev_job = reinterpret_cast<const aa_dev_frame *>( c->dev + off );
  rec.chunk = static_cast<int>( c - s->chunks.data() );
  c->live_frames++;
  const int fi = static_cast<int>( s->frames.size() );
  s->frames.push_back( std::move( rec ) );
  if ( frame_index ) *frame_index = fi;
  if ( hdr_out ) *hdr_out = h;
  return AA_OK;
}
} // namespace
extern "C" {

aa_status aa_stream_parse( aa_stream * s, const uint8_t * data, size_t size, int * frame_index, aa_frame_header * hdr_out )
{
  if ( !s || !data ) return fail( AA_ERR_ARGUMENT, "aa_stream_parse: null argument" );
  if ( aa_status st = set_device( s->ctx ) ) return st;
  if ( aa_status st = segmap_to_host( s ) ) return st;
  return append_host_frame( s, frame_index, hdr_out, [&]( aa_frame_header & hdr, aa_mb_info * mbs, int16_t * coeffs ) { s->parser.parse( data, size, hdr, mbs, coeffs ); } );
}

/* A frame given as RECORDS, no bitstream: what Encoder::write_frame has in hand when it updates its references
 * (encoder.cc:146-160: frame.decode + frame.loopfilter + copy_to on a Frame it holds), and what xc-enc -r replays. */
aa_status aa_stream_append_records( aa_stream * s, const aa_frame_header * hdr, const aa_mb_info * mbs_in, const int16_t * coeffs_in, int * frame_index )
{
  if ( !s || !hdr || !mbs_in ) return fail( AA_ERR_ARGUMENT, "aa_stream_append_records: null argument" );
  if ( hdr->num_coeff_blocks && !coeffs_in ) return fail( AA_ERR_ARGUMENT, "aa_stream_append_records: coefficient blocks missing" );
  const size_t nmb = size_t( s->parser.mb_width() ) * s->parser.mb_height();
  if ( hdr->mb_width != s->parser.mb_width() || hdr->mb_height != s->parser.mb_height() || hdr->num_macroblocks != nmb )
    return fail( AA_ERR_ARGUMENT, "aa_stream_append_records: the header's macroblock dimensions are not this decoder's" );
  if ( hdr->num_coeff_blocks > nmb * 25 ) return fail( AA_ERR_ARGUMENT, "aa_stream_append_records: more coefficient blocks than macroblocks can hold" );
  // every macroblock's blocks must lie inside the array that came with it (the kernels trust coeff_index + popcount( nz_mask ))
  uint32_t intra = 0;
  for ( size_t i = 0; i < nmb; i++ ) {
    const uint32_t nblk = static_cast<uint32_t>( __builtin_popcount( mbs_in[i].nz_mask & 0x1FFFFFFu ) );
    if ( ( mbs_in[i].nz_mask >> 25 ) || ( nblk && size_t( mbs_in[i].coeff_index ) + nblk > hdr->num_coeff_blocks ) )
      return fail( AA_ERR_ARGUMENT, "aa_stream_append_records: macroblock " + std::to_string( i ) + " points outside the coefficient blocks" );
    if ( mbs_in[i].y_mode > 9 || mbs_in[i].uv_mode > 3 || mbs_in[i].ref_frame > 3 || mbs_in[i].segment_id > 3 || mbs_in[i].lf_level > 63 )
      return fail( AA_ERR_ARGUMENT, "aa_stream_append_records: macroblock " + std::to_string( i ) + " has a field out of range" );
    if ( !( mbs_in[i].flags & AA_MB_INTER ) ) intra++;
    else if ( hdr->key_frame ) return fail( AA_ERR_ARGUMENT, "aa_stream_append_records: inter macroblock in a key frame" );
  }
  return append_host_frame( s, frame_index, nullptr, [&]( aa_frame_header & h, aa_mb_info * mbs, int16_t * coeffs ) {
    h = *hdr;
    h.num_intra_mbs = intra; h.has_intra_mb = intra != 0;
    std::memcpy( mbs, mbs_in, nmb * sizeof( aa_mb_info ) );
    if ( hdr->num_coeff_blocks ) std::memcpy( coeffs, coeffs_in, size_t( hdr->num_coeff_blocks ) * 32 );
  } );
}

aa_status aa_stream_upload( aa_stream * s )
{
  if ( !s ) return fail( AA_ERR_ARGUMENT, "null stream" );
  if ( aa_status st = set_device( s->ctx ) ) return st;
  bool any = false;
  for ( auto & c : s->chunks ) {
    if ( c.host && c.uploaded < c.used ) {
      HIP_TRY( hipMemcpyAsync( c.dev + c.uploaded, c.host + c.uploaded, c.used - c.uploaded, hipMemcpyHostToDevice, s->ctx->copy ) );
      c.uploaded = c.used; any = true;
    }
  }
  if ( any ) {
    HIP_TRY( hipEventRecord( s->ctx->upload_done, s->ctx->copy ) );
    HIP_TRY( hipStreamWaitEvent( s->ctx->compute, s->ctx->upload_done, 0 ) );
  }
  return AA_OK;
}

aa_status aa_stream_release_staging( aa_stream * s )
{
  if ( !s ) return fail( AA_ERR_ARGUMENT, "null stream" );
  if ( aa_status st = set_device( s->ctx ) ) return st;
  if ( aa_status st = aa_stream_upload( s ) ) return st;
  HIP_TRY( hipStreamSynchronize( s->ctx->copy ) );
  for ( auto & c : s->chunks ) {
    if ( !c.host ) continue;
    { std::lock_guard<std::mutex> g( s->ctx->pool_mu ); s->ctx->pinned_pool.emplace_back( c.host, c.pinned_bytes ); }
    c.host = nullptr;
    c.capacity = c.used;           // sealed: the next frame is staged in a new chunk
  }
  for ( auto & f : s->frames ) f.host_job = nullptr;
  return AA_OK;
}


/* ---------------- device-side entropy decode ---------------- */
namespace {
struct SubmitItem {
  aa_stream * s; const uint8_t * data; size_t size;
  size_t data_off;          // in the batch arena
  size_t rec_off = 0;       // the frame's record block, in the device half of the arena only
  aa_status status = AA_OK; std::string error;
  int frame_index = -1;
  bool seg_enabled = false, seg_reset = false;
  bool on_host = false;     // goes to a host lane (aa_ctx::HostLanes), not to the GPU's job queue
};

// one frame of one stream: header pre-pass on the host, compressed bytes into the pinned arena, record block + raster slot
aa_status submit_one( Batch * b, SubmitItem & it, int item, aa::ParseJob * jobs_host, aa_dev_frame * dframes_host, bool want_host_lane )
{
  aa_stream * s = it.s;
  aa_ctx * ctx = s->ctx;
  aa::ParseJob & J = jobs_host[item];
  FrameRec rec;
  try { s->parser.parse_header( it.data, it.size, rec.hdr, J.fp ); }
  catch ( const aa::ParseError & e ) { it.error = e.message; return e.code; }
  it.seg_enabled = J.fp.seg_enabled; it.seg_reset = s->parser.segment_map_reset();
  std::memcpy( b->host + it.data_off, it.data, it.size );

  const uint32_t nmb = uint32_t( J.fp.mbw ) * J.fp.mbh;
  const size_t mb_bytes = align_up( size_t( nmb ) * sizeof( aa_mb_info ) );
  const size_t words_per_row = ( J.fp.mbw + 63 ) / 64;
  const size_t rows_bytes = align_up( words_per_row * J.fp.mbh * sizeof( unsigned long long ) );
  const uint32_t flags_padded = ( nmb + 15u ) & ~15u;
  // one lane per partition: a second copy of the flags laid out partition by partition, a longer chunk list (tok_fsm.hh)
  const bool mp = ctx->tok.lane_per_partition && J.fp.nparts > 1;
  const uint32_t mp_stride = mp ? aa::mp_flag_stride( J.fp.mbw, J.fp.mbh, J.fp.nparts ) : 0u;
  const size_t flags_bytes = align_up( size_t( flags_padded ) + size_t( J.fp.nparts ) * mp_stride );
  const size_t list_bytes = align_up( size_t( aa::chunk_list_entries( nmb, mp ? J.fp.nparts : 1u ) ) * sizeof( uint32_t ) );

  const size_t pos_bytes = ctx->tok.packed ? align_up( size_t( nmb ) * sizeof( uint32_t ) ) : 0;
  rec.rec_bytes = mb_bytes + rows_bytes + flags_bytes + list_bytes + pos_bytes;
  rec.rec_block = b->dev + it.rec_off; rec.rec_in_arena = true;
  uint8_t * blk = rec.rec_block;

  J.data = b->dev + it.data_off;
  J.size = static_cast<uint32_t>( it.size ); J.data_padded = ( J.size + 15u ) & ~15u;
  J.nmb = nmb; J.flags_padded = flags_padded; J.mp_stride = mp_stride; J.mp_pad = 0;
  J.mbs = reinterpret_cast<aa_mb_info *>( blk );
  J.intra_rows = reinterpret_cast<unsigned long long *>( blk + mb_bytes );
  J.mbflags = blk + mb_bytes + rows_bytes;
  J.chunk_list = reinterpret_cast<uint32_t *>( blk + mb_bytes + rows_bytes + flags_bytes );
  J.packed_pos = pos_bytes ? reinterpret_cast<uint32_t *>( blk + mb_bytes + rows_bytes + flags_bytes + list_bytes ) : nullptr;
  rec.packed_pos = J.packed_pos; rec.dev_mbs = J.mbs;
  J.summary = reinterpret_cast<aa::FrameSummary *>( b->host_dev + b->summaries_off ) + item;     // pinned + mapped: no copy back
  rec.chunk_list = J.chunk_list;
  rec.summary = reinterpret_cast<volatile aa::FrameSummary *>( b->host + b->summaries_off ) + item;
  rec.parse_job = reinterpret_cast<const aa::ParseJob *>( b->dev ) + item;

  aa_dev_frame * job = &dframes_host[item];
  rec.hdr.has_intra_mb = 1;              // until the device parser has counted: the row masks say which macroblocks are intra
  fill_job( rec, job );
  job->mbs = J.mbs; job->intra_rows = J.intra_rows;
  job->coeffs = reinterpret_cast<const int16_t *>( ctx->tok.heap );     // coeff_index of a device-parsed macroblock = its first block's index in the heap
  job->packed = ctx->tok.packed ? 1u : 0u;                              // ... packed storage: the 40-bit offset of its words (tok_fsm.hh store_mb_packed)
  rec.host_job = job;
  rec.dev_job = reinterpret_cast<const aa_dev_frame *>( b->dev + ( reinterpret_cast<uint8_t *>( job ) - b->host ) );
  rec.batch = b; rec.batch_item = item; rec.summary_pending = true;
  if ( want_host_lane && !J.fp.seg_enabled ) {
    // A host lane takes the frame (not one of a stream that uses segmentation: the persistent segment map is the one piece of
    // macroblock data a frame inherits, and that stream's map lives on the device).  Its records are dense blocks in a piece of
    // their own, like a host-parsed frame's: no chunk list, no packed words; and it counts as handed over from now on -- whoever
    // releases it waits for the worker's `done` word.
    it.on_host = true;
    job->packed = 0;
    rec.packed_pos = nullptr; rec.chunk_list = nullptr;
    rec.enqueued = true;
  }
  it.frame_index = static_cast<int>( s->frames.size() );
  s->frames.push_back( std::move( rec ) );
  return AA_OK;
}
} // namespace

namespace {
// second phase of a batch: its frames go to the job queue of the token workers (behind the macroblock-header kernel on the
// batch's stream), the heap is grown for what they are expected to store, workgroups are launched if too few are alive
aa_status launch_tokens_of( aa_ctx * ctx, Batch * b )
{
  if ( !b->tokens_pending ) return AA_OK;
  auto & T = ctx->tok;
  aa::ParseJob * jobs_host = reinterpret_cast<aa::ParseJob *>( b->host );
  // the ring of job slots must not wrap onto jobs that have not been taken
  if ( static_cast<uint32_t>( T.jobs_enqueued ) - T.mirror_host->q_head + static_cast<uint32_t>( b->n ) + 4096u > T.q_slots ) {
    if ( aa_status st = tok_refresh_mirror( ctx ) ) return st;
    if ( static_cast<uint32_t>( T.jobs_enqueued ) - T.mirror_host->q_head + static_cast<uint32_t>( b->n ) + 4096u > T.q_slots )
      if ( aa_status st = tok_quiesce( ctx ) ) return st;
  }
  b->tokens_pending = false;
  ctx->deferred.erase( std::remove( ctx->deferred.begin(), ctx->deferred.end(), b ), ctx->deferred.end() );
  bool dropped = false;
  tok_prune_inflight( ctx );
  for ( int i = 0; i < b->n; i++ ) {
    Batch::Item & it = b->items[i];
    if ( !it.live ) { if ( jobs_host[i].nmb ) { jobs_host[i].nmb = 0; dropped = true; } continue; }   // rejected by the pre-pass, or released since: the lane that draws it drops it
    FrameRec & r = it.s->frames[it.frame];
    if ( it.on_host ) continue;                    // (a host lane's: no ticket, no chunks)
    r.enqueued = true;
    // what the frame is expected to store, in chunks: blocks per compressed byte as frames have turned out so far (the ratio
    // holds across key and inter frames and quantisers far better than blocks per macroblock), a margin, and the chunk its
    // lane will be filling when it ends
    if ( T.packed ) {
      const double words = std::min( double( aa::kMbWords ) * jobs_host[i].nmb, T.words_per_byte * 1.15 * jobs_host[i].size );
      r.est_chunks = static_cast<uint32_t>( words / ( aa::kChunkWords - aa::kMbWords ) ) + 1u;
    } else {
      const double blocks = std::min( 25.0 * jobs_host[i].nmb, T.blocks_per_byte * 1.15 * jobs_host[i].size );
      r.est_chunks = static_cast<uint32_t>( blocks / ( aa::kChunkBlocks - aa::kMbBlocks ) ) + 1u;
    }
    // (one lane per partition: every lane of the frame fills a chunk of its own -- P - 1 more partly filled chunks than the words say;
    // without them the heap was mapped for a quarter of what 4-partition frames take and their lanes sat out the 2-s memory wait)
    if ( T.lane_per_partition && jobs_host[i].mp_stride && jobs_host[i].fp.nparts > 1 ) r.est_chunks += static_cast<uint32_t>( jobs_host[i].fp.nparts ) - 1u;
    T.chunks_committed += r.est_chunks;
    ctx->stats.parsed_macroblocks += jobs_host[i].nmb;
  }
  if ( std::find( T.inflight.begin(), T.inflight.end(), b ) == T.inflight.end() ) T.inflight.push_back( b );
  if ( std::find( T.batches.begin(), T.batches.end(), b ) == T.batches.end() ) T.batches.push_back( b );
  if ( aa_status st = tok_grow_heap( ctx, static_cast<size_t>( std::max<int64_t>( 0, T.chunks_committed ) ) * kChunkBytesHeap ) ) return st;
  hipStream_t ps = b->ps;
  if ( b->patch_jobs && dropped ) HIP_TRY( hipMemcpyAsync( b->dev, b->host, b->head_bytes, hipMemcpyHostToDevice, ps ) );     // (behind the header kernel on its stream)
  if ( dropped ) {
    // frames released between the two phases of a two-phase submit get no ticket: release_records waits only for frames that
    // were handed to the queue, so a ticket nobody waits for could be drawn after the arena has been recycled
    int keep = 0;
    for ( int k = 0; k < b->n_order; k++ ) if ( b->items[b->launch_order_host[k]].live ) b->launch_order_host[keep++] = b->launch_order_host[k];
    b->n_order = keep;
    if ( keep ) HIP_TRY( hipMemcpyAsync( const_cast<uint32_t *>( b->launch_order_dev ), b->launch_order_host, size_t( keep ) * sizeof( uint32_t ), hipMemcpyHostToDevice, ps ) );
  }
  if ( b->n_order > 0 ) {
    if ( int e = aa::launch_enqueue_jobs( T.q, T.slots, reinterpret_cast<const aa::ParseJob *>( b->dev ), b->launch_order_dev, b->n_order, ps ) )
      return hip_fail( static_cast<hipError_t>( e ), "k_enqueue_jobs" );
    T.jobs_enqueued += static_cast<uint64_t>( b->n_order );
  }
  HIP_TRY( hipEventRecord( b->hdr_done, ps ) );
  HIP_TRY( hipEventRecord( ctx->parse_idle[b->parse_stream_index], ps ) );
  return tok_service( ctx, b->hdr_done );
}
} // namespace

namespace {
// ---- host lanes (aa_ctx::HostLanes) ----
// One frame: macroblock headers + tokens from the arena's pinned copy (the header pre-pass left everything else in its ParseJob),
// records to HBM where the frame's job record says they are, dense coefficient blocks to a pool piece of their own, then the
// summary and -- last, behind the copies -- the `done` word.  Scratch buffers are the worker's own, kept from frame to frame.
// (PINNED: a copy out of pageable memory makes the runtime pin and unpin pages under every transfer -- page-table updates that hold up
// the whole GPU; with sixteen workers uploading 5-MB key frames that was enough to run a row kernel's bounded wait out during priming)
struct HostLaneScratch {
  uint8_t * pin = nullptr; size_t pin_bytes = 0;      // macroblock records | intra row masks | worst-case dense coefficient blocks
  std::vector<uint8_t> above; hipEvent_t ev = nullptr;
  bool fit( size_t bytes )
  {
    if ( bytes <= pin_bytes ) return true;
    if ( pin ) (void) hipHostFree( pin );
    pin = nullptr; pin_bytes = 0;
    if ( hipHostMalloc( reinterpret_cast<void **>( &pin ), bytes, hipHostMallocDefault ) != hipSuccess ) { (void) hipGetLastError(); pin = nullptr; return false; }
    pin_bytes = bytes;
    return true;
  }
};
void host_lane_run( aa_ctx * ctx, Batch * b, int item, HostLaneScratch & S )
{
  const aa::ParseJob & J = reinterpret_cast<const aa::ParseJob *>( b->host )[item];
  volatile aa::FrameSummary * sum = reinterpret_cast<volatile aa::FrameSummary *>( b->host + b->summaries_off ) + item;
  Batch::Item & it = b->items[item];
  const uint32_t mbw = J.fp.mbw, mbh = J.fp.mbh, nmb = mbw * mbh;
  const size_t words_per_row = ( mbw + 63 ) / 64;
  uint32_t status = aa::TOK_OK, blocks = 0, intra = 0, split = 0;
  hipStream_t up = ctx->tok.host_up;              // (behind the arena's upload: the submit call made this stream wait for it)
  if ( !S.ev ) status = aa::TOK_HOST_FAILED;      // (no event to wait for the uploads with: see host_lanes_main)
  const size_t mbs_bytes = align_up( size_t( nmb ) * sizeof( aa_mb_info ) ), rows_bytes = align_up( words_per_row * mbh * sizeof( unsigned long long ) );
  try { S.above.resize( size_t( mbw ) * 9 ); } catch ( const std::bad_alloc & ) { status = aa::TOK_HOST_FAILED; }
  if ( status == aa::TOK_OK && !S.fit( mbs_bytes + rows_bytes + size_t( nmb ) * 25 * 32 + 256 ) ) status = aa::TOK_HOST_FAILED;
  if ( status == aa::TOK_OK ) {
    aa_mb_info * mbs = reinterpret_cast<aa_mb_info *>( S.pin );
    unsigned long long * rows = reinterpret_cast<unsigned long long *>( S.pin + mbs_bytes );
    int16_t * coeffs = reinterpret_cast<int16_t *>( S.pin + mbs_bytes + rows_bytes );
    std::memset( static_cast<void *>( mbs ), 0, size_t( nmb ) * sizeof( aa_mb_info ) );
    const double t_parse = now_ms();
    aa::parse_frame_body( b->host + it.data_off, J.fp, mbs, coeffs, S.above.data(), &blocks, &intra );
    ctx->host_lanes.parse_us += static_cast<uint64_t>( ( now_ms() - t_parse ) * 1e3 ); ctx->host_lanes.parsed_bytes += J.size;
    std::memset( rows, 0, words_per_row * mbh * sizeof( unsigned long long ) );
    for ( uint32_t r = 0; r < mbh; r++ ) for ( uint32_t c = 0; c < mbw; c++ ) {
      const aa_mb_info & mb = mbs[r * mbw + c];
      if ( !( mb.flags & AA_MB_INTER ) ) rows[r * words_per_row + ( c >> 6 )] |= 1ull << ( c & 63 );
      else if ( mb.y_mode == aa::SPLITMV ) split = 1;
    }
    uint8_t * dense = nullptr;
    const size_t dense_bytes = align_up( std::max<size_t>( size_t( blocks ) * 32, 32 ) );
    hipError_t e = hipSuccess;
    if ( dev_alloc( ctx, dense_bytes, &dense ) != AA_OK ) status = aa::TOK_HOST_FAILED;
    else {
      it.host_dense = dense; it.host_dense_bytes = dense_bytes;
      // (the arena's upload -- which carries the job record as the pre-pass left it -- was queued on the copy stream by the submit call,
      // which also made the host lanes' stream wait for it)
      e = hipMemcpyAsync( J.mbs, mbs, size_t( nmb ) * sizeof( aa_mb_info ), hipMemcpyHostToDevice, up );
      if ( e == hipSuccess ) e = hipMemcpyAsync( J.intra_rows, rows, words_per_row * mbh * sizeof( unsigned long long ), hipMemcpyHostToDevice, up );
      if ( e == hipSuccess && blocks ) e = hipMemcpyAsync( dense, coeffs, size_t( blocks ) * 32, hipMemcpyHostToDevice, up );
      if ( e == hipSuccess ) {
        // the frame's reconstruction job record (aa_dev_frame, in the arena behind the parse jobs): its blocks are here, not in the heap
        uint8_t * job_dev = b->dev + align_up( size_t( b->n ) * sizeof( aa::ParseJob ) ) + size_t( item ) * sizeof( aa_dev_frame );
        const int16_t ** dense_ptr = reinterpret_cast<const int16_t **>( S.pin + S.pin_bytes - 16 );       // (pinned too: the last bytes of the scratch)
        *dense_ptr = reinterpret_cast<const int16_t *>( dense );
        e = hipMemcpyAsync( job_dev + offsetof( aa_dev_frame, coeffs ), dense_ptr, sizeof *dense_ptr, hipMemcpyHostToDevice, up );
      }
      if ( e == hipSuccess ) e = hipEventRecord( S.ev, up );
      if ( e == hipSuccess ) e = hipEventSynchronize( S.ev );
      if ( e != hipSuccess ) status = aa::TOK_HOST_FAILED;
    }
  }
  sum->num_coeff_blocks = blocks; sum->num_intra_mbs = intra; sum->has_split = split; sum->steps = 0; sum->num_chunks = 0; sum->packed_words = 0;
  sum->status = status;
  ctx->host_lanes.backlog_bytes -= J.size;
  __atomic_thread_fence( __ATOMIC_RELEASE );
  sum->done = 1u;
}
void host_lanes_main( aa_ctx * ctx )
{
  (void) hipSetDevice( ctx->device );
  HostLaneScratch S;
  for ( int tries = 0; tries < 3 && !S.ev; tries++ )     // (without it the worker can only fail its frames: host_lane_run says TOK_HOST_FAILED)
    if ( hipEventCreateWithFlags( &S.ev, hipEventDisableTiming ) != hipSuccess ) { (void) hipGetLastError(); S.ev = nullptr; usleep( 1000 ); }
  auto & H = ctx->host_lanes;
  for ( ;; ) {
    aa_ctx::HostLanes::Task t;
    {
      std::unique_lock<std::mutex> g( H.mu );
      H.cv.wait( g, [&] { return H.stop || !H.q.empty(); } );
      if ( H.q.empty() ) break;                      // (stop: after the queue has been worked off -- somebody may wait for those frames)
      t = H.q.front(); H.q.pop_front();
    }
    host_lane_run( ctx, t.b, t.item, S );
  }
  if ( S.ev ) (void) hipEventDestroy( S.ev );
  if ( S.pin ) (void) hipHostFree( S.pin );
}
void host_lanes_start( aa_ctx * ctx )
{
  auto & H = ctx->host_lanes;
  if ( !H.threads.empty() ) return;
  int nt = effective_cpus();
  if ( const char * e = std::getenv( "ALFALFA_AMD_HOST_LANES" ) ) nt = std::max( 1, atoi( e ) );
  for ( int t = 0; t < nt; t++ ) H.threads.emplace_back( host_lanes_main, ctx );
}
// compressed bytes the host lanes get through per millisecond, all workers together (measured on the frames parsed so far --
// parse time only, so a cold first call's allocations do not depress it; before the first frame: 24 KB/ms per usable core)
double host_lanes_rate( aa_ctx * ctx )
{
  auto & H = ctx->host_lanes;
  int nt = static_cast<int>( H.threads.size() );
  if ( !nt ) { nt = effective_cpus(); if ( const char * e = std::getenv( "ALFALFA_AMD_HOST_LANES" ) ) nt = std::max( 1, atoi( e ) ); }
  const uint64_t us = H.parse_us.load(), bytes = H.parsed_bytes.load();
  const double per_worker = us > 1000 ? static_cast<double>( bytes ) / static_cast<double>( us ) * 1e3 : 24.0e3;     // bytes per ms
  return per_worker * std::min( nt, effective_cpus() );
}
void host_lanes_stop( aa_ctx * ctx )
{
  auto & H = ctx->host_lanes;
  { std::lock_guard<std::mutex> g( H.mu ); H.stop = true; }
  H.cv.notify_all();
  for ( auto & t : H.threads ) t.join();
  H.threads.clear();
}
} // namespace

aa_status aa_launch_tokens( aa_ctx * ctx, int max_batches, int * launched_out )
{
  if ( !ctx ) return fail( AA_ERR_ARGUMENT, "aa_launch_tokens: null context" );
  if ( aa_status st = set_device( ctx ) ) return st;
  int launched = 0;
  while ( !ctx->deferred.empty() && ( max_batches <= 0 || launched < max_batches ) ) {
    if ( aa_status st = launch_tokens_of( ctx, ctx->deferred.front() ) ) return st;
    launched++;
  }
  if ( launched_out ) *launched_out = launched;
  return AA_OK;
}

aa_status aa_submit_frames( aa_ctx * ctx, const aa_frame_in * frames, int n, int * frame_index_out, int threads )
{
  return aa_submit_frames_ex( ctx, frames, n, frame_index_out, threads, 0 );
}

aa_status aa_submit_frames_ex( aa_ctx * ctx, const aa_frame_in * frames, int n, int * frame_index_out, int threads, unsigned flags )
{
  if ( !ctx || !frames || n <= 0 ) return fail( AA_ERR_ARGUMENT, "aa_submit_frames: bad argument" );
  const bool defer_tokens = ( flags & AA_SUBMIT_DEFER_TOKENS ) != 0;
  if ( aa_status st = set_device( ctx ) ) return st;
  if ( aa_status st = tok_init( ctx ) ) return st;
  std::vector<SubmitItem> items( n );
  // arena layout: parse jobs | reconstruction job records | summaries | segment-pass lists | compressed frames
  const size_t jobs_bytes = align_up( size_t( n ) * sizeof( aa::ParseJob ) );
  const size_t dframes_bytes = align_up( size_t( n ) * sizeof( aa_dev_frame ) );
  const size_t sums_bytes = align_up( size_t( n ) * sizeof( aa::FrameSummary ) );
  const size_t seg_bytes = align_up( size_t( n ) * ( sizeof( aa_seg_stream ) + 2 * sizeof( uint32_t ) ) );   // + the launch order
  size_t off = jobs_bytes + dframes_bytes + sums_bytes + seg_bytes;
  std::map<aa_stream *, std::vector<int>> by_stream;
  std::vector<aa_stream *> stream_order;
  for ( int i = 0; i < n; i++ ) {
    if ( !frames[i].stream || !frames[i].data || frames[i].stream->ctx != ctx ) return fail( AA_ERR_ARGUMENT, "aa_submit_frames: null frame or stream of another context" );
    items[i].s = frames[i].stream; items[i].data = frames[i].data; items[i].size = frames[i].size;
    items[i].data_off = off;
    off += align_up( frames[i].size + 16 );
    auto & v = by_stream[frames[i].stream];
    if ( v.empty() ) stream_order.push_back( frames[i].stream );
    v.push_back( i );
  }
  for ( aa_stream * s : stream_order )
    if ( s->next_submit > static_cast<int>( s->frames.size() ) ) return fail( AA_ERR_LOGIC, "aa_submit_frames: stream state is inconsistent" );

  bool to_host_lanes = false;
  std::vector<char> host_lane_wanted( n, 0 );     // [i]: frame i should go to a host lane (if it may: submit_one)
  // ---- route: few chains -> the host's cores ----
  // A GPU lane decodes a bool in ~0.3 us, a host core in ~4 ns: one core is worth ~75 lanes, and a frame on a lane is a chain of
  // seconds whatever else the GPU does.  The GPU wins by holding 22 000 chains at once; a call with fewer streams than the host
  // has workers (an ExCamera bundle of 8 chunks, a player's single stream) is through sooner -- and at a higher rate -- when
  // each stream's frames are parsed by one host worker (Parser::parse, the same records) and uploaded.  Frames of one stream
  // are serial on a core, so what counts is streams per worker, not frames.
  {
    const int nt = worker_threads( threads );
    const char * route_env = std::getenv( "ALFALFA_AMD_ROUTE" );            // "device" / "host": tests and experiments; default: by size
    const bool force_device = ( flags & AA_SUBMIT_DEVICE ) || ( route_env && route_env[0] == 'd' );
    const bool force_host = ( flags & AA_SUBMIT_HOST ) || ( route_env && route_env[0] == 'h' );
    // measured (round 3, 1080p, 256-core host): 1 stream 0.67 M macroblocks/s on the host route vs 0.14 M on the GPU lanes, 8 streams
    // 4.8 M vs 2.4 M -- but 64 streams 9.8 M vs 14.9 M (64 workers do not scale on this host's memory system): the bound is 24
    const bool few = static_cast<int>( stream_order.size() ) <= std::min( nt, 24 );
    if ( !defer_tokens && !force_device && force_host && !few ) {
      // many streams, and the caller wants them parsed by the host's cores (AA_SUBMIT_HOST: frames that are needed at once -- the key
      // frames of the group a pipeline starts with: 35 ms on a core, 2 s as a chain on a lane): HOST LANES.  The frames take the
      // device route below -- header pre-pass, arena, records in HBM -- but their tickets go to worker threads of the context, which
      // finish them with the `done` word a GPU lane writes.  The call does not wait for them (round 4's version of this route did,
      // ~1 s for a group's key frames, and the hand-overs behind it started that much later).
      to_host_lanes = true;
    }
    // Round 6: FRAME-PARALLEL parse of few streams.  A call that brings SEVERAL frames per stream (a player's look-ahead, an ExCamera
    // bundle's chunks: player.cc:134-144, decode-bundle.cc:56-99) used to parse each stream's frames one after the other on one
    // worker -- one stream = one core, whatever the box has -- and inside the call.  VP8 has no backward adaptation: once the header
    // pre-pass (microseconds per frame, serial across a stream) has run, every frame body is an independent chain
    // (decoder_state.hh:92-97,126-131), which is exactly what the host lanes take.  Such a call therefore goes the host lanes' way
    // too: all its frames in parallel on the context's worker threads, the call returns at once, reconstruction waits for each
    // frame's `done` word.  Not for streams that use segmentation (their persistent map lives with whoever parses them in order:
    // the per-stream route below) and not for one-frame-per-stream calls (nothing to run in parallel; the caller waits anyway).
    if ( !defer_tokens && !force_device && !force_host && few && !to_host_lanes && n > static_cast<int>( stream_order.size() ) ) {
      const char * few_env = std::getenv( "ALFALFA_AMD_FEW_ROUTE" );
      const bool per_stream = few_env && few_env[0] == 's';                // "streams": the round-3 route (A/B runs, tests)
      bool seg = false;
      for ( aa_stream * s : stream_order ) seg = seg || s->parser.segmentation().enabled || s->segmap_on_device;
      if ( !per_stream && !seg ) to_host_lanes = true;
    }
    if ( !defer_tokens && !force_device && !to_host_lanes && ( force_host || few ) ) {
      std::atomic<size_t> next { 0 };
      auto work = [&]() {
        for ( ;; ) {
          const size_t k = next.fetch_add( 1 );
          if ( k >= stream_order.size() ) return;
          bool broken = false;
          for ( int i : by_stream[stream_order[k]] ) {
            SubmitItem & it = items[i];
            if ( broken ) { it.status = AA_ERR_LOGIC; it.error = "an earlier frame of this stream in the same call failed"; continue; }
            it.status = aa_stream_parse( it.s, it.data, it.size, &it.frame_index, nullptr );
            if ( it.status != AA_OK ) { it.error = g_last_error; it.frame_index = -1; broken = true; }
          }
        }
      };
      const int workers = std::max( 1, std::min<int>( nt, static_cast<int>( stream_order.size() ) ) );
      if ( workers == 1 ) work();
      else {
        std::vector<std::thread> pool;
        for ( int t = 0; t < workers; t++ ) pool.emplace_back( work );
        for ( auto & t : pool ) t.join();
      }
      ctx->stats.host_routed_frames += static_cast<uint64_t>( n );
      aa_status first_error = AA_OK; std::string first_message;
      for ( int i = 0; i < n; i++ ) {
        if ( frame_index_out ) frame_index_out[i] = items[i].frame_index;
        if ( items[i].status != AA_OK && first_error == AA_OK ) { first_error = items[i].status; first_message = items[i].error; }
      }
      return first_error == AA_OK ? AA_OK : fail( first_error, first_message );
    }
    // ---- hybrid: a big call's KEY frames on the host's cores, the rest on the lanes ----
    // A key frame's chain is the longest there is (2.2 s on a lane at 1080p, ~35 ms on a core) and nothing of its group can be
    // reconstructed before it is parsed; a call's key frames are few (one per stream and group of pictures).  Streams whose
    // frames in this call are all key frames go to the HOST LANES, biggest first, while what the host lanes have been given and
    // not finished stays within `host_share_ms` of their work (at the rate they have really achieved: bytes of compressed frames
    // per millisecond of a worker's parse time x workers; until measured, 24 KB/ms per usable core); everything else takes the
    // GPU's lanes.  Nothing here waits: round 4's version of this share parsed inside the call and was worth taking only when
    // the host could have half of a call's key frames within the budget.
    if ( !defer_tokens && !force_device && !to_host_lanes && ctx->host_share_ms > 0 ) {
      std::vector<std::pair<size_t, aa_stream *>> cand;
      for ( aa_stream * s : stream_order ) {
        size_t bytes = 0; bool all_key = true;
        for ( int i : by_stream[s] ) { all_key = all_key && frames[i].size >= 10 && ( frames[i].data[0] & 1u ) == 0u; bytes += frames[i].size; }
        if ( all_key ) cand.emplace_back( bytes, s );
      }
      std::stable_sort( cand.begin(), cand.end(), []( const auto & a, const auto & b ) { return a.first > b.first; } );
      const double capacity_bytes = ctx->host_share_ms * host_lanes_rate( ctx );
      double taken = static_cast<double>( ctx->host_lanes.backlog_bytes.load() );
      for ( auto & c : cand ) {
        if ( taken + static_cast<double>( c.first ) > capacity_bytes ) break;
        taken += static_cast<double>( c.first );
        for ( int i : by_stream[c.second] ) host_lane_wanted[i] = 1;
      }
    }
  }
  if ( to_host_lanes ) std::fill( host_lane_wanted.begin(), host_lane_wanted.end(), 1 );

  std::unique_ptr<Batch> b( new Batch );
  const size_t arena = ( off + ( size_t( 16 ) << 20 ) - 1 ) & ~( ( size_t( 16 ) << 20 ) - 1 );
  // The frames' record blocks (macroblock records, flags, chunk list: written by the parse kernels only) sit behind the mirrored
  // part in the DEVICE half of the arena: one piece of the pool per call instead of one per frame (thousands of allocator
  // round trips per call); they go back when the last frame of the call is released.
  size_t dev_arena = arena;
  for ( int i = 0; i < n; i++ ) {
    const aa_stream * s = frames[i].stream;
    const size_t nmb = size_t( s->parser.mb_width() ) * s->parser.mb_height();
    const size_t words_per_row = ( s->parser.mb_width() + 63 ) / 64;
    items[i].rec_off = dev_arena;
    dev_arena += align_up( nmb * sizeof( aa_mb_info ) ) + align_up( words_per_row * s->parser.mb_height() * sizeof( unsigned long long ) )
                 + ( ctx->tok.lane_per_partition      // (how many partitions a frame has is known after its pre-pass: room for 8)
                       ? align_up( ( ( nmb + 15u ) & ~size_t( 15 ) ) + 8 * size_t( aa::mp_flag_stride( s->parser.mb_width(), s->parser.mb_height(), 8 ) ) )
                         + align_up( size_t( aa::chunk_list_entries( static_cast<uint32_t>( nmb ), 8 ) ) * sizeof( uint32_t ) )
                       : align_up( ( nmb + 15u ) & ~size_t( 15 ) ) + align_up( size_t( aa::chunk_list_entries( static_cast<uint32_t>( nmb ) ) ) * sizeof( uint32_t ) ) )
                 + ( ctx->tok.packed ? align_up( nmb * sizeof( uint32_t ) ) : 0 );
  }
  dev_arena = ( dev_arena + ( size_t( 16 ) << 20 ) - 1 ) & ~( ( size_t( 16 ) << 20 ) - 1 );
  b->host = pinned_get( ctx, arena, &b->host_bytes );
  if ( !b->host ) return fail( AA_ERR_HIP, "aa_submit_frames: pinned staging allocation failed" );
  b->dev_bytes = dev_arena;
  if ( aa_status st = dev_alloc( ctx, dev_arena, &b->dev, true ) ) { std::lock_guard<std::mutex> g( ctx->pool_mu ); ctx->pinned_pool.emplace_back( b->host, b->host_bytes ); return st; }
  b->n = n; b->summaries_off = jobs_bytes + dframes_bytes;
  if ( hipError_t e = hipHostGetDevicePointer( reinterpret_cast<void **>( &b->host_dev ), b->host, 0 ) ) {
    { std::lock_guard<std::mutex> g( ctx->pool_mu ); ctx->pinned_pool.emplace_back( b->host, b->host_bytes ); }
    dev_free( ctx, b->dev, b->dev_bytes );
    return hip_fail( e, "hipHostGetDevicePointer (batch arena)" );
  }
  aa::ParseJob * jobs_host = reinterpret_cast<aa::ParseJob *>( b->host );
  aa_dev_frame * dframes_host = reinterpret_cast<aa_dev_frame *>( b->host + jobs_bytes );
  std::memset( b->host, 0, jobs_bytes + dframes_bytes + sums_bytes + seg_bytes );

  // ---- host half, one worker per stream at a time (the header pre-pass is serial across the frames of a stream) ----
  {
    std::atomic<size_t> next { 0 };
    // (while the host lanes have frames to parse -- frames somebody needs at once -- the pre-pass workers of later hand-overs step
    // back: a thread may lower its own priority.  On a box that grants 16 CPUs the first group's key frames were seen to take 4 s
    // instead of 1 when twenty hand-overs' worth of pre-pass threads and arena copies ran beside them.)
    const int nt = std::min<int>( worker_threads( threads ), static_cast<int>( stream_order.size() ) );
    const bool step_back = nt > 1 && ctx->host_lanes.backlog_bytes.load() > 0;          // (nt == 1: the caller's own thread does the work)
    auto work = [&]() {
      (void) hipSetDevice( ctx->device );
      if ( step_back ) (void) setpriority( PRIO_PROCESS, static_cast<id_t>( syscall( SYS_gettid ) ), 10 );
      for ( ;; ) {
        const size_t k = next.fetch_add( 1 );
        if ( k >= stream_order.size() ) return;
        bool broken = false;
        for ( int i : by_stream[stream_order[k]] ) {
          SubmitItem & it = items[i];
          if ( broken ) { it.status = AA_ERR_LOGIC; it.error = "an earlier frame of this stream in the same call failed"; continue; }
          it.status = submit_one( b.get(), it, i, jobs_host, dframes_host, host_lane_wanted[i] != 0 );
          if ( it.status != AA_OK ) broken = true;
        }
      }
    };
    if ( nt == 1 ) work();
    else {
      std::vector<std::thread> pool;
      for ( int t = 0; t < nt; t++ ) pool.emplace_back( work );
      for ( auto & t : pool ) t.join();
    }
  }
  aa_status first_error = AA_OK; std::string first_message;
  int appended = 0, max_mbw = 0, max_nparts = 1;
  for ( int i = 0; i < n; i++ ) {
    if ( frame_index_out ) frame_index_out[i] = items[i].frame_index;
    if ( items[i].status == AA_OK ) { appended++; max_mbw = std::max<int>( max_mbw, jobs_host[i].fp.mbw ); max_nparts = std::max<int>( max_nparts, jobs_host[i].fp.nparts ); }
    else {
      jobs_host[i].nmb = 0;                        // the kernels skip it
      if ( first_error == AA_OK ) { first_error = items[i].status; first_message = items[i].error; }
    }
  }
  if ( !appended ) {
    { std::lock_guard<std::mutex> g( ctx->pool_mu ); ctx->pinned_pool.emplace_back( b->host, b->host_bytes ); }
    dev_free( ctx, b->dev, b->dev_bytes );
    return fail( first_error, first_message );
  }
  b->live = appended;
  b->items.resize( n );
  for ( int i = 0; i < n; i++ ) { b->items[i] = { items[i].s, items[i].frame_index, items[i].status == AA_OK }; b->items[i].on_host = items[i].status == AA_OK && items[i].on_host; b->items[i].data_off = items[i].data_off; }
  b->head_bytes = jobs_bytes + dframes_bytes;
  b->max_mbw = max_mbw; b->max_nparts = max_nparts;
  if ( ctx->tok.lane_per_partition ) ctx->tok.mp_hint = std::max<uint32_t>( ctx->tok.mp_hint, static_cast<uint32_t>( max_nparts ) );   // (workgroups launched from now on leave that many lanes per ticket)
  // From here on the frames that were appended point at the batch.  If anything below fails they are given back one by one
  // (the last one frees the batch): no frame is left with a dangling batch, nothing leaks; the frames themselves stay in
  // their streams as frames whose records are gone (decoding them reports that).
  struct Abandon {
    aa_ctx * ctx; Batch * b; bool armed = true;
    ~Abandon() {
      if ( !armed ) return;
      const std::string keep = g_last_error;
      {                                                     // (host lanes' frames that no worker was given: nobody will write their `done` word)
        volatile aa::FrameSummary * sums = reinterpret_cast<volatile aa::FrameSummary *>( b->host + b->summaries_off );
        for ( size_t i = 0; i < b->items.size(); i++ ) if ( b->items[i].on_host ) { sums[i].status = aa::TOK_HOST_FAILED; sums[i].done = 1u; }
      }
      std::vector<Batch::Item> its = b->items;              // (the batch dies with its last frame)
      for ( auto & it : its ) if ( it.live ) release_records( it.s, it.s->frames[it.frame], true );
      g_last_error = keep;
    }
  } abandon { ctx, b.release() };
  Batch * const raw = abandon.b;
  if ( aa_status st = tok_set_lane_bytes( ctx, aa::tok::lane_lds_bytes( static_cast<uint32_t>( max_mbw ), max_nparts > 1, ctx->tok.lane_per_partition ) ) ) return st;

  // ---- segment-map pass lists (only streams that use segmentation in this batch) ----
  aa_seg_stream * seg_streams = reinterpret_cast<aa_seg_stream *>( raw->host + jobs_bytes + dframes_bytes + sums_bytes );
  uint32_t * seg_order = reinterpret_cast<uint32_t *>( seg_streams + n );
  int n_seg_streams = 0; uint32_t n_seg_order = 0;
  // a stream with nothing queued if there is one (else the next in turn: the batch waits behind that stream's work)
  int pick = ctx->next_parse_stream;
  for ( int k = 0; k < ctx->n_parse_streams; k++ ) {
    const int c = ( ctx->next_parse_stream + k ) % ctx->n_parse_streams;
    if ( hipEventQuery( ctx->parse_idle[c] ) == hipSuccess ) { pick = c; break; }
  }
  (void) hipGetLastError();
  ctx->next_parse_stream = ( pick + 1 ) % ctx->n_parse_streams;
  raw->parse_stream_index = pick;
  hipStream_t ps = ctx->parse_streams[pick];
  for ( aa_stream * s : stream_order ) {
    bool any = false;
    for ( int i : by_stream[s] ) if ( items[i].status == AA_OK && items[i].seg_enabled ) any = true;
    if ( !any && !s->segmap_on_device ) continue;
    if ( !any ) continue;
    const size_t map_bytes = size_t( s->parser.mb_width() ) * s->parser.mb_height();
    if ( !s->dev_segmap ) if ( aa_status st = dev_alloc( ctx, map_bytes, &s->dev_segmap ) ) return st;
    if ( !s->segmap_on_device ) {                  // the host parser owns the current map: hand it over
      HIP_TRY( hipMemcpyAsync( s->dev_segmap, s->parser.segment_map().data(), map_bytes, hipMemcpyHostToDevice, ps ) );
      HIP_TRY( hipStreamSynchronize( ps ) );
      s->segmap_on_device = true;
    }
    aa_seg_stream & ss = seg_streams[n_seg_streams++];
    ss.map = s->dev_segmap; ss.first = n_seg_order; ss.count = 0;
    for ( int i : by_stream[s] ) if ( items[i].status == AA_OK ) { seg_order[n_seg_order++] = static_cast<uint32_t>( i ) | ( items[i].seg_reset ? 0x80000000u : 0u ); ss.count++; }
  }

  // launch order: longest chains first (the compressed size is the length of a token chain, near enough), so that the lanes
  // of a wave finish together and the long waves start first
  // (only frames the pre-pass accepted: a ticket in the job queue is a pointer into this arena that a lane may follow long
  // after the call -- a rejected frame has nobody who would wait for its lane before the arena is recycled)
  uint32_t * launch_order = seg_order + n;
  int n_order = 0;
  for ( int i = 0; i < n; i++ ) if ( items[i].status == AA_OK && !items[i].on_host ) launch_order[n_order++] = static_cast<uint32_t>( i );      // (host lanes' frames: no ticket)
  std::stable_sort( launch_order, launch_order + n_order, [&]( uint32_t a, uint32_t b ) { return items[a].size > items[b].size; } );
  const uint32_t * launch_order_dev = reinterpret_cast<const uint32_t *>( raw->dev + ( reinterpret_cast<uint8_t *>( launch_order ) - raw->host ) );

  // ---- device half: arena to HBM on the copy stream, then the header kernels on one of the parse streams ----
  HIP_TRY( hipEventCreateWithFlags( &raw->hdr_done, hipEventDisableTiming ) );
  HIP_TRY( hipMemcpyAsync( raw->dev, raw->host, off, hipMemcpyHostToDevice, ctx->copy ) );
  hipEvent_t up = get_event( ctx );
  HIP_TRY( hipEventRecord( up, ctx->copy ) );
  HIP_TRY( hipStreamWaitEvent( ps, up, 0 ) );
  { bool any_host = false; for ( int i = 0; i < n; i++ ) any_host = any_host || raw->items[i].on_host;
    if ( any_host ) HIP_TRY( hipStreamWaitEvent( ctx->tok.host_up, up, 0 ) ); }      // (the host lanes patch job records the arena's upload carries)
  ctx->free_events.push_back( up );
  const aa::ParseJob * jobs_dev = reinterpret_cast<const aa::ParseJob *>( raw->dev );
  if ( n_order ) {
    LaunchTimer t( ctx, 3, ps );
    if ( int e = aa::launch_parse_mb_headers( jobs_dev, launch_order_dev, n_order, ps ) ) return hip_fail( static_cast<hipError_t>( e ), "k_parse_mb_headers" );
  }
  if ( n_seg_streams ) {
    if ( ctx->last_seg_batch ) HIP_TRY( hipStreamWaitEvent( ps, ctx->last_seg_batch, 0 ) );
    const uint8_t * segs_dev = raw->dev + jobs_bytes + dframes_bytes + sums_bytes;
    if ( int e = aa::launch_segment_fixup( jobs_dev, reinterpret_cast<const aa_seg_stream *>( segs_dev ), n_seg_streams,
                                           reinterpret_cast<const uint32_t *>( segs_dev + size_t( n ) * sizeof( aa_seg_stream ) ), ps ) )
      return hip_fail( static_cast<hipError_t>( e ), "k_segment_fixup" );
    if ( !ctx->last_seg_batch ) HIP_TRY( hipEventCreateWithFlags( &ctx->last_seg_batch, hipEventDisableTiming ) );
    HIP_TRY( hipEventRecord( ctx->last_seg_batch, ps ) );
  }
  raw->ps = ps; raw->launch_order_dev = launch_order_dev; raw->launch_order_host = launch_order; raw->n_order = n_order;
  raw->tokens_pending = true; raw->patch_jobs = defer_tokens;
  HIP_TRY( hipEventRecord( raw->hdr_done, ps ) );  // (the header kernels; recorded again behind the hand-over to the job queue)
  HIP_TRY( hipEventRecord( ctx->parse_idle[pick], ps ) );
  ctx->deferred.push_back( raw );
  abandon.armed = false;                           // the batch is on the books: from here on a failure leaves a consistent state
  // the frames of the host lanes: behind the arena's upload on the copy stream (a worker patches its frame's job record there)
  {
    int n_host = 0;
    for ( int i = 0; i < n; i++ ) if ( raw->items[i].on_host ) n_host++;
    if ( n_host ) {
      host_lanes_start( ctx );
      { std::lock_guard<std::mutex> g( ctx->host_lanes.mu );
        for ( int i = 0; i < n; i++ ) if ( raw->items[i].on_host ) { ctx->host_lanes.backlog_bytes += items[i].size; ctx->host_lanes.q.push_back( { raw, i } ); } }
      ctx->host_lanes.cv.notify_all();
      ctx->stats.host_routed_frames += static_cast<uint64_t>( n_host );
      if ( std::find( ctx->tok.inflight.begin(), ctx->tok.inflight.end(), raw ) == ctx->tok.inflight.end() ) ctx->tok.inflight.push_back( raw );
    }
  }
  if ( !defer_tokens ) if ( aa_status st = launch_tokens_of( ctx, raw ) ) return st;
  if ( first_error != AA_OK ) return fail( first_error, first_message );
  return AA_OK;
}

// counts the device parser produced (intra macroblocks, coefficient blocks, SPLITMV) -> the frame's header; waits for the parse
static aa_status resolve_summary( aa_stream * s, FrameRec & r )
{
  if ( !r.summary_pending ) return AA_OK;
  aa_ctx * ctx = s->ctx;
  auto & T = ctx->tok;
  Batch * b = r.batch;
  if ( !b || !r.summary ) return fail( AA_ERR_LOGIC, "frame records were released before the frame was decoded" );
  if ( b->tokens_pending ) if ( aa_status st = launch_tokens_of( ctx, b ) ) return st;      // two-phase submit, second phase not asked for yet
  volatile aa::FrameSummary * sum = r.summary;
  for ( int attempt = 0; ; attempt++ ) {
    if ( aa_status st = tok_wait_done( ctx, sum ) ) return st;
    if ( sum->status != aa::TOK_NO_MEMORY ) break;
    // The coefficient pool ran dry under this frame's lane and nothing came back in time: the lane handed the frame back.  Its
    // chunks are returned, room is made (more heap if the memory limit allows it, else by letting everything else in flight
    // finish) and the frame goes to the queue again -- its macroblock headers are parsed already.
    ctx->stats.nomem_retries++;
    const uint32_t worst = aa::chunk_list_entries( r.hdr.num_macroblocks, T.lane_per_partition ? 8u : 1u );      // (what no frame of this size exceeds, in either storage format)
    {
      std::lock_guard<std::mutex> g( ctx->pool_mu );
      // (marked as returned at once: a call that fails below and is repeated must not push the list a second time)
      if ( !r.chunks_returned ) { T.pending_lists.push_back( r.chunk_list ); r.chunks_returned = true; T.chunks_committed -= r.est_chunks; r.est_chunks = 0; }
      flush_chunk_frees( ctx );                     // (and the chunks of everything the caller has released since)
      if ( !T.pending_lists.empty() ) return fail( AA_ERR_HIP, "k_pool_free_lists could not be launched" );
    }
    HIP_TRY( hipStreamSynchronize( ctx->compute ) );
    if ( aa_status st = tok_grow_heap( ctx, T.heap_mapped + static_cast<size_t>( worst ) * kChunkBytesHeap ) ) return st;
    if ( aa_status st = tok_refresh_mirror( ctx ) ) return st;
    if ( T.mirror_host->pool_avail < static_cast<int32_t>( worst ) ) {
      if ( aa_status st = tok_quiesce( ctx ) ) return st;
      if ( aa_status st = tok_refresh_mirror( ctx ) ) return st;
    }
    if ( T.mirror_host->pool_avail < static_cast<int32_t>( worst ) ) {
      // Still not enough: the heap is held by frames that were parsed AHEAD of this one (submitted later, needed later).  They
      // give their chunks back -- newest first -- and are parsed again when their turn comes: to them it is as if their lane had
      // handed them back.  Thrashing is slow, but a caller that looks far ahead never gets stuck on the frame it needs now.
      int32_t have = T.mirror_host->pool_avail;
      std::lock_guard<std::mutex> g( ctx->pool_mu );
      for ( auto bi = T.batches.rbegin(); bi != T.batches.rend() && have < static_cast<int32_t>( worst ); ++bi ) {
        Batch * vb = *bi;
        volatile aa::FrameSummary * sums = reinterpret_cast<volatile aa::FrameSummary *>( vb->host + vb->summaries_off );
        for ( int i = vb->n - 1; i >= 0 && have < static_cast<int32_t>( worst ); i-- ) {
          if ( !vb->items[i].live ) continue;
          aa_stream * vs = vb->items[i].s;
          FrameRec & v = vs->frames[vb->items[i].frame];
          if ( &v == &r || vb->items[i].on_host || !v.enqueued || v.records_released || v.chunks_returned || !sums[i].done || vb->items[i].frame < vs->next_submit ) continue;     // (a host lane's frame holds no chunks)
          T.pending_lists.push_back( v.chunk_list );
          have += static_cast<int32_t>( sums[i].num_chunks );
          v.chunks_returned = true; v.summary_pending = true;
          T.chunks_committed -= v.est_chunks; v.est_chunks = 0;
          sums[i].status = aa::TOK_NO_MEMORY;
          ctx->stats.frames_evicted++;
        }
      }
      flush_chunk_frees( ctx );
      if ( !T.pending_lists.empty() ) return fail( AA_ERR_HIP, "k_pool_free_lists could not be launched" );
    }
    HIP_TRY( hipStreamSynchronize( ctx->compute ) );
    if ( aa_status st = tok_refresh_mirror( ctx ) ) return st;
    if ( attempt >= 8 || T.mirror_host->pool_avail < static_cast<int32_t>( worst ) )
      return fail( AA_ERR_NO_MEMORY, "device parser: the coefficient heap is exhausted (" + std::to_string( T.heap_mapped >> 20 ) + " MiB mapped, "
                                     + std::to_string( T.mirror_host->pool_avail ) + " chunks free, this frame may need " + std::to_string( worst )
                                     + "): release decoded frames (aa_stream_release_before) or raise the limit (aa_ctx_set_memory_limit); the call can be repeated" );
    sum->status = 0; sum->done = 0; sum->num_chunks = 0;
    r.chunks_returned = false;
    r.est_chunks = worst; T.chunks_committed += worst;
    __atomic_thread_fence( __ATOMIC_SEQ_CST );
    if ( int e = aa::launch_enqueue_jobs( T.q, T.slots, r.parse_job, T.one_dev, 1, T.util ) ) return hip_fail( static_cast<hipError_t>( e ), "k_enqueue_jobs" );
    T.jobs_enqueued += 1;
    if ( std::find( T.inflight.begin(), T.inflight.end(), b ) == T.inflight.end() ) T.inflight.push_back( b );
    if ( aa_status st = tok_service( ctx ) ) return st;
  }
  if ( sum->status == aa::TOK_HOST_FAILED ) return fail( AA_ERR_HIP, "host lane: the frame's records could not be placed in device memory" );
  if ( sum->status == aa::TOK_STEP_BOUND ) return fail( AA_ERR_HIP, "device parser: a token lane exceeded the step bound of its frame size (records are not valid)" );
  r.hdr.num_coeff_blocks = sum->num_coeff_blocks;
  r.hdr.num_intra_mbs = sum->num_intra_mbs;
  r.hdr.has_intra_mb = sum->num_intra_mbs != 0;
  r.has_split = sum->has_split != 0;
  r.intra_diagonals.assign( r.hdr.mb_width + 2 * ( r.hdr.mb_height - 1 ), r.hdr.has_intra_mb ? 1 : 0 );   // diagonal schedule: all of them
  // the books: what the frame really took; what macroblocks store on this content (feeds the estimate of later frames)
  T.chunks_committed += static_cast<int64_t>( sum->num_chunks ) - static_cast<int64_t>( r.est_chunks );
  r.est_chunks = sum->num_chunks;
  if ( r.hdr.compressed_size ) T.blocks_per_byte += 0.05 * ( static_cast<double>( sum->num_coeff_blocks ) / r.hdr.compressed_size - T.blocks_per_byte );
  if ( r.hdr.compressed_size && r.packed_pos ) T.words_per_byte += 0.05 * ( static_cast<double>( sum->packed_words ) / r.hdr.compressed_size - T.words_per_byte );
  if ( r.packed_pos ) { ctx->stats.packed_frames++; ctx->stats.packed_words += sum->packed_words; ctx->stats.packed_blocks += sum->num_coeff_blocks; }
  ctx->stats.token_steps += sum->steps; ctx->stats.token_frames++;
  r.summary_pending = false;
  return AA_OK;
}

aa_status aa_stream_frame_header( aa_stream * s, int fi, aa_frame_header * out )
{
  if ( !s || !out || fi < 0 || fi >= static_cast<int>( s->frames.size() ) ) return fail( AA_ERR_ARGUMENT, "aa_stream_frame_header: bad argument" );
  if ( aa_status st = set_device( s->ctx ) ) return st;
  if ( aa_status st = resolve_summary( s, s->frames[fi] ) ) return st;
  *out = s->frames[fi].hdr;
  return AA_OK;
}

/* Debug / test view of a frame's parsed records as they sit in HBM (host- or device-parsed alike). */
aa_status aa_stream_read_records( aa_stream * s, int fi, aa_mb_info * mb_out, int16_t * coeff_out, size_t coeff_capacity_blocks )
{
  if ( !s || fi < 0 || fi >= static_cast<int>( s->frames.size() ) ) return fail( AA_ERR_ARGUMENT, "aa_stream_read_records: bad argument" );
  if ( aa_status st = set_device( s->ctx ) ) return st;
  FrameRec & r = s->frames[fi];
  if ( r.records_released ) return fail( AA_ERR_LOGIC, "aa_stream_read_records: frame records were released" );
  if ( aa_status st = resolve_summary( s, r ) ) return st;
  if ( aa_status st = aa_stream_upload( s ) ) return st;
  HIP_TRY( hipStreamSynchronize( s->ctx->copy ) );        // (also covers the arena of a host-parsed submit call: same stream)
  aa_dev_frame job;
  HIP_TRY( hipMemcpy( &job, r.dev_job, sizeof job, hipMemcpyDeviceToHost ) );
  if ( coeff_out && coeff_capacity_blocks < r.hdr.num_coeff_blocks ) return fail( AA_ERR_ARGUMENT, "aa_stream_read_records: coefficient buffer too small" );
  if ( !r.chunk_list ) {                     // host-parsed: the blocks follow each other in the frame-store chunk
    if ( mb_out ) HIP_TRY( hipMemcpy( mb_out, job.mbs, size_t( r.hdr.num_macroblocks ) * sizeof( aa_mb_info ), hipMemcpyDeviceToHost ) );
    if ( coeff_out ) HIP_TRY( hipMemcpy( coeff_out, job.coeffs, size_t( r.hdr.num_coeff_blocks ) * 32, hipMemcpyDeviceToHost ) );
    return AA_OK;
  }
  // device-parsed: the blocks sit in chunks of the coefficient heap and coeff_index is a heap index.  The caller gets the
  // frame's own view -- blocks back to back in parse order, coeff_index counted from the frame's first block -- which is
  // what the host parser produces.
  std::vector<aa_mb_info> mbs( r.hdr.num_macroblocks );
  HIP_TRY( hipMemcpy( mbs.data(), job.mbs, mbs.size() * sizeof( aa_mb_info ), hipMemcpyDeviceToHost ) );
  std::vector<uint32_t> list( size_t( r.est_chunks ) + 1 );
  HIP_TRY( hipMemcpy( list.data(), r.chunk_list, list.size() * sizeof( uint32_t ), hipMemcpyDeviceToHost ) );
  if ( list[0] != r.est_chunks ) return fail( AA_ERR_LOGIC, "aa_stream_read_records: chunk list disagrees with the parse summary" );
  if ( r.packed_pos ) {
    // packed storage: the frame's words are expanded here as k_dense_index / k_expand_coeffs do it on the device (coeff_pack.hh)
    std::vector<uint32_t> pos( r.hdr.num_macroblocks );
    HIP_TRY( hipMemcpy( pos.data(), r.packed_pos, pos.size() * sizeof( uint32_t ), hipMemcpyDeviceToHost ) );
    std::vector<std::vector<int16_t>> words( list[0] );
    if ( coeff_out )
      for ( uint32_t k = 0; k < list[0]; k++ ) {
        words[k].resize( aa::kChunkWords );
        HIP_TRY( hipMemcpy( words[k].data(), s->ctx->tok.heap + size_t( list[1 + k] ) * kChunkBytesHeap, kChunkBytesHeap, hipMemcpyDeviceToHost ) );
      }
    uint32_t running = 0;
    for ( size_t mi = 0; mi < mbs.size(); mi++ ) {
      aa_mb_info & mb = mbs[mi];
      const uint32_t nblk = aa::pack::blocks_of( mb.nz_mask );
      if ( nblk && coeff_out ) {
        const uint32_t ord = pos[mi] >> 15, off = pos[mi] & ( aa::kChunkWords - 1u );
        if ( ord >= list[0] || running + nblk > r.hdr.num_coeff_blocks )
          return fail( AA_ERR_LOGIC, "aa_stream_read_records: a macroblock's coefficients lie outside the frame's chunks" );
        // (a macroblock's words end inside its chunk -- the lane made sure of it; checked all the same: the mask slots, then the
        // values its masks announce)
        const int16_t * w = words[ord].data() + off, * end = words[ord].data() + aa::kChunkWords;
        if ( w + aa::pack::kMaskSlots > end ) return fail( AA_ERR_LOGIC, "aa_stream_read_records: packed coefficients run past their chunk" );
        uint32_t values = 0;
        for ( uint32_t b = 0; b < 25; b++ ) if ( ( mb.nz_mask >> b ) & 1u ) values += aa::pack::popc( static_cast<uint16_t>( w[b] ) );
        if ( w + aa::pack::kMaskSlots + values > end ) return fail( AA_ERR_LOGIC, "aa_stream_read_records: packed coefficients run past their chunk" );
        // (the record's own offset -- what the reconstruction kernels follow -- must name the same words)
        if ( ( static_cast<unsigned long long>( mb.reserved ) << 32 | mb.coeff_index ) != static_cast<unsigned long long>( list[1 + ord] ) * aa::kChunkWords + off )
          return fail( AA_ERR_LOGIC, "aa_stream_read_records: a macroblock's word offset disagrees with its place in the frame's chunks" );
        aa::pack::expand_macroblock( w, mb.nz_mask, coeff_out + size_t( running ) * 16 );
      }
      mb.coeff_index = running; mb.reserved = 0;
      running += nblk;
    }
    if ( running != r.hdr.num_coeff_blocks ) return fail( AA_ERR_LOGIC, "aa_stream_read_records: non-zero masks disagree with the parse summary" );
    if ( mb_out ) std::memcpy( mb_out, mbs.data(), mbs.size() * sizeof( aa_mb_info ) );
    return AA_OK;
  }
  std::map<uint32_t, std::vector<uint8_t>> chunk;
  if ( coeff_out )
    for ( uint32_t k = 0; k < list[0]; k++ ) {
      auto & v = chunk[list[1 + k]];
      v.resize( kChunkBytesHeap );
      HIP_TRY( hipMemcpy( v.data(), s->ctx->tok.heap + size_t( list[1 + k] ) * kChunkBytesHeap, kChunkBytesHeap, hipMemcpyDeviceToHost ) );
    }
  uint32_t running = 0;
  for ( auto & mb : mbs ) {
    const uint32_t nblk = static_cast<uint32_t>( __builtin_popcount( mb.nz_mask ) );
    if ( nblk && coeff_out ) {
      auto it = chunk.find( mb.coeff_index / aa::kChunkBlocks );
      if ( it == chunk.end() || mb.coeff_index % aa::kChunkBlocks + nblk > aa::kChunkBlocks || running + nblk > r.hdr.num_coeff_blocks )
        return fail( AA_ERR_LOGIC, "aa_stream_read_records: a macroblock's coefficients lie outside the frame's chunks" );
      std::memcpy( coeff_out + size_t( running ) * 16, it->second.data() + size_t( mb.coeff_index % aa::kChunkBlocks ) * 32, size_t( nblk ) * 32 );
    }
    mb.coeff_index = running;
    running += nblk;
  }
  if ( running != r.hdr.num_coeff_blocks ) return fail( AA_ERR_LOGIC, "aa_stream_read_records: non-zero masks disagree with the parse summary" );
  if ( mb_out ) std::memcpy( mb_out, mbs.data(), mbs.size() * sizeof( aa_mb_info ) );
  return AA_OK;
}

namespace {
// Row-pipelined loop filter over a set of frames: four frames of ONE geometry per wave.  Bucket by geometry, pad every bucket
// to a multiple of four with null frames (slot 0 of a group is never null), split at group boundaries when a list is full.
aa_status launch_lf_rows( aa_ctx * ctx, std::vector<std::pair<uint32_t, const aa_dev_frame *>> & keyed, bool same_geometry, int max_mbh, int max_mbw )
{
  if ( !same_geometry ) std::stable_sort( keyed.begin(), keyed.end(), []( const auto & a, const auto & b ) { return a.first < b.first; } );
  aa_frame_list list;
  int filled = 0;
  auto launch = [&]() -> aa_status {
    if ( !filled ) return AA_OK;
    for ( int k = filled; k < AA_MAX_BATCH; k++ ) list.f[k] = nullptr;
    if ( aa_status st = zero_ws( ctx, ctx->ws, filled / 4, max_mbh ) ) return st;
    const size_t need = size_t( filled ) * max_mbh * max_mbw * 128;
    if ( need > ctx->boundary_bytes ) {
      HIP_TRY( hipStreamSynchronize( ctx->compute ) );
      if ( ctx->boundary ) (void) hipFree( ctx->boundary );
      ctx->boundary = nullptr; ctx->boundary_bytes = 0;
      HIP_TRY( hipMalloc( reinterpret_cast<void **>( &ctx->boundary ), need ) );
      ctx->boundary_bytes = need;
    }
    LaunchTimer t( ctx, 2 );
    if ( const int e = aa::launch_loopfilter_rows4( list, filled / 4, max_mbh, max_mbw, ctx->ws, ctx->boundary, ctx->n_xcd, ctx->compute ) ) return hip_fail( static_cast<hipError_t>( e ), "k_loopfilter_rows4" );
    filled = 0;
    return AA_OK;
  };
  for ( size_t i = 0; i < keyed.size(); ) {
    size_t j = i;
    while ( j < keyed.size() && j - i < 4 && keyed[j].first == keyed[i].first ) j++;
    for ( size_t k = 0; k < 4; k++ ) list.f[filled++] = i + k < j ? keyed[i + k].second : nullptr;
    i = j;
    if ( filled + 4 > AA_MAX_BATCH ) if ( aa_status st = launch() ) return st;
  }
  return launch();
}
} // namespace

namespace {
// ALFALFA_AMD_DECODE_TIMING=1 (diagnostics, off by default): where the host's time inside aa_decode_batch goes, section by section, printed
// to stderr when a context is destroyed.  (Round 5: the plateau's calls take 24 ms each with the counted waits at zero)
struct DecodeTiming {
  bool on = false; double ms[10] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 }; uint64_t calls = 0;
  ~DecodeTiming()      // (the static below: at process exit)
  {
    if ( !on || !calls ) return;
    static const char * const names[10] = { "checks", "stream uploads", "collect_pending", "resolve_summary (waits for the parser included)", "dense buffer",
                                            "bind_batch (waits for a binding buffer included)", "expand_packed", "job lists", "inter launches", "intra + loop-filter launches" };
    std::fprintf( stderr, "alfalfa_amd: aa_decode_batch host time over %llu calls, ms per call:", static_cast<unsigned long long>( calls ) );
    for ( int k = 0; k < 10; k++ ) std::fprintf( stderr, " [%s] %.3f", names[k], ms[k] / static_cast<double>( calls ) );
    std::fprintf( stderr, "\n" );
  }
};
DecodeTiming * decode_timing() { static DecodeTiming d; static const bool init = [] { const char * e = std::getenv( "ALFALFA_AMD_DECODE_TIMING" ); d.on = e && atoi( e ) != 0; return true; }(); (void) init; return &d; }
struct DecodeMark {
  DecodeTiming & d; double t;
  DecodeMark() : d( *decode_timing() ), t( d.on ? now_ms() : 0.0 ) { if ( d.on ) d.calls++; }
  void lap( int k ) { if ( d.on ) { const double n = now_ms(); d.ms[k] += n - t; t = n; } }
};
} // namespace

aa_status aa_decode_batch( aa_ctx * ctx, aa_stream * const * streams, int n, const int * frame_index )
{
  if ( !ctx || !streams || !frame_index || n <= 0 ) return fail( AA_ERR_ARGUMENT, "aa_decode_batch: bad argument" );
  if ( aa_status st = set_device( ctx ) ) return st;
  std::vector<const aa_dev_frame *> inter_jobs, intra_jobs, lf_jobs;
  std::vector<const FrameRec *> intra_recs;
  std::vector<uint32_t> lf_geometry;
  unsigned max_mbs = 0; int max_mbw = 0, max_mbh = 0;
  bool any_split = false;
  bool same_geometry = true;     // two-frames-per-wave loop filter needs equal macroblock dimensions in the batch
  uint64_t total_mbs = 0;
  DecodeMark mk;
  for ( int i = 0; i < n; i++ ) {
    aa_stream * s = streams[i];
    if ( !s || s->ctx != ctx ) return fail( AA_ERR_ARGUMENT, "aa_decode_batch: stream belongs to another context" );
    const int fi = frame_index[i];
    if ( fi < 0 || fi >= static_cast<int>( s->frames.size() ) ) return fail( AA_ERR_ARGUMENT, "aa_decode_batch: frame index out of range" );
    if ( fi != s->next_submit ) return fail( AA_ERR_LOGIC, "aa_decode_batch: frames of a stream must be submitted in order" );
  }
  mk.lap( 0 );
  for ( int i = 0; i < n; i++ ) if ( aa_status st = aa_stream_upload( streams[i] ) ) return st;
  mk.lap( 1 );
  // coefficient chunks of the frames released since the last call go back to the pool (behind the kernels that read them:
  // those were queued before the release)
  // ... and what was released since then gets its epoch now: reusable as soon as the kernels queued before this call have run
  // (not whenever some later allocation happens to miss its free list)
  { std::lock_guard<std::mutex> g( ctx->pool_mu ); collect_pending( ctx, false ); }
  mk.lap( 2 );
  for ( int i = 0; i < n; i++ ) {
    if ( streams[i]->frames[frame_index[i]].records_released ) return fail( AA_ERR_LOGIC, "aa_decode_batch: frame records were released" );
    if ( aa_status st = resolve_summary( streams[i], streams[i]->frames[frame_index[i]] ) ) return st;
  }
  mk.lap( 3 );
  // (packed coefficient storage: the reconstruction kernels read the packed words themselves -- kernels.hip, residual_x4 -- since
  // round 6; rounds 3-5 expanded every frame of the call into a transient dense array here: k_dense_index + k_expand_coeffs, a
  // quarter of the reconstruction's GPU time and 972 bytes of HBM traffic per macroblock)
  // rasters released while binding (old references, outputs nobody holds) must not be recycled before this call's kernels
  // are queued: no release epoch is closed until then
  struct BindGuard { aa_ctx * c;
                     explicit BindGuard( aa_ctx * x ) : c( x ) { std::lock_guard<std::mutex> g( c->pool_mu ); c->binding_depth++; }
                     ~BindGuard() {
                       std::lock_guard<std::mutex> g( c->pool_mu );
                       if ( --c->binding_depth == 0 ) {      // every launch of the call is queued (or the call failed before any): what it released may be handed out
                         for ( auto & h : c->compute_hold ) { c->compute_free[h.second].push_back( h.first ); c->compute_free_bytes += h.second; }
                         c->compute_hold.clear();
                       } } } bind_guard( ctx );
  mk.lap( 4 );
  if ( aa_status st = bind_batch( ctx, streams, n, frame_index ) ) return st;
  mk.lap( 5 );
  // frames count as submitted only once every launch of the batch has been queued (a failed launch must not leave them
  // looking decoded)
  struct Advance { aa_stream * const * streams; int n; bool ok = false; ~Advance() { if ( ok ) for ( int i = 0; i < n; i++ ) streams[i]->next_submit++; } } advance { streams, n };
  mk.lap( 6 );
  for ( int i = 0; i < n; i++ ) {
    aa_stream * s = streams[i];
    const FrameRec & r = s->frames[frame_index[i]];
    const aa_frame_header & h = r.hdr;
    if ( h.num_intra_mbs < h.num_macroblocks ) { inter_jobs.push_back( r.dev_job ); any_split = any_split || r.has_split; }
    if ( h.has_intra_mb ) { intra_jobs.push_back( r.dev_job ); intra_recs.push_back( &r ); }
    if ( h.loop_filter_level ) { lf_jobs.push_back( r.dev_job ); lf_geometry.push_back( ( static_cast<uint32_t>( h.mb_width ) << 16 ) | h.mb_height ); }
    if ( i > 0 && ( h.mb_width != streams[0]->frames[frame_index[0]].hdr.mb_width || h.mb_height != streams[0]->frames[frame_index[0]].hdr.mb_height ) ) same_geometry = false;
    max_mbs = std::max<unsigned>( max_mbs, h.num_macroblocks );
    max_mbw = std::max<int>( max_mbw, h.mb_width ); max_mbh = std::max<int>( max_mbh, h.mb_height );
    total_mbs += h.num_macroblocks;
  }
  ctx->stats.macroblocks += total_mbs;
  mk.lap( 7 );

  auto for_each_list = [&]( const std::vector<const aa_dev_frame *> & jobs, auto && fn ) -> int {
    for ( size_t base = 0; base < jobs.size(); base += AA_MAX_BATCH ) {
      aa_frame_list list;
      const int cnt = static_cast<int>( std::min<size_t>( AA_MAX_BATCH, jobs.size() - base ) );
      for ( int k = 0; k < cnt; k++ ) list.f[k] = jobs[base + k];
      for ( int k = cnt; k < AA_MAX_BATCH; k++ ) list.f[k] = nullptr;
      if ( int e = fn( list, cnt ) ) return e;
    }
    return 0;
  };
  auto check = [&]( int e, const char * what ) -> aa_status {
    if ( e ) return hip_fail( static_cast<hipError_t>( e ), what );
    return AA_OK;
  };

  // 1. every inter-coded macroblock of the batch: whole-vector macroblocks four per wave; SPLITMV ones (if any) one per wave
  if ( !inter_jobs.empty() ) {
    int e = for_each_list( inter_jobs, [&]( const aa_frame_list & l, int cnt ) {
      LaunchTimer t( ctx, 0 ); return aa::launch_recon_inter4( l, cnt, max_mbs, ctx->compute ); } );
    if ( aa_status st = check( e, "k_recon_inter4" ) ) return st;
    if ( any_split ) {
      e = for_each_list( inter_jobs, [&]( const aa_frame_list & l, int cnt ) {
        LaunchTimer t( ctx, 5 ); return aa::launch_recon_inter( l, cnt, max_mbs, true, ctx->compute ); } );
      if ( aa_status st = check( e, "k_recon_inter" ) ) return st;
    }
  }
  mk.lap( 8 );
  const int ndiag = max_mbw + 2 * ( max_mbh - 1 );
  if ( ctx->schedule == 0 ) {
    // 2+3. row-pipelined kernels: one launch each; macroblock rows are ordered in-launch (ticket + progress words)
    if ( aa_status st = ensure_ws( ctx, &ctx->ws, &ctx->ws_bytes, max_mbh ) ) return st;
    // intra rows: four frames per wave (any geometry), groups padded with null frames
    for ( size_t base = 0; base < intra_jobs.size(); base += AA_MAX_BATCH ) {
      aa_frame_list list;
      const int cnt = static_cast<int>( std::min<size_t>( AA_MAX_BATCH, intra_jobs.size() - base ) );
      const int groups = ( cnt + 3 ) / 4;
      for ( int k = 0; k < AA_MAX_BATCH; k++ ) list.f[k] = k < cnt ? intra_jobs[base + k] : nullptr;
      if ( aa_status st = zero_ws( ctx, ctx->ws, groups * 4, max_mbh ) ) return st;
      LaunchTimer t( ctx, 1 );
      if ( const int e = aa::launch_recon_intra4( list, groups, max_mbh, ctx->ws, ctx->n_xcd, ctx->compute ) ) return hip_fail( static_cast<hipError_t>( e ), "k_recon_intra4" );
    }
    if ( !lf_jobs.empty() ) {
      std::vector<std::pair<uint32_t, const aa_dev_frame *>> keyed;
      keyed.reserve( lf_geometry.size() );
      for ( size_t i = 0; i < lf_jobs.size(); i++ ) keyed.emplace_back( lf_geometry[i], lf_jobs[i] );
      if ( aa_status st = launch_lf_rows( ctx, keyed, same_geometry, max_mbh, max_mbw ) ) return st;
    }
    mk.lap( 9 );
    advance.ok = true;
    return AA_OK;
  }
  // ---- ALFALFA_AMD_SCHEDULE=diagonal: the kernel boundary is the inter-workgroup synchronisation ----
  // 2. intra macroblocks, 2:1 anti-diagonal order (left, above-left, above, above-right are final)
  if ( !intra_jobs.empty() ) {
    for ( int d = 0; d < ndiag; d++ ) {
      bool needed = false;
      for ( const FrameRec * r : intra_recs ) if ( d < static_cast<int>( r->intra_diagonals.size() ) && r->intra_diagonals[d] ) { needed = true; break; }
      if ( !needed ) continue;
      const int row_lo = std::max( 0, ( d - ( max_mbw - 1 ) + 1 ) / 2 ), row_hi = std::min( max_mbh - 1, d / 2 );
      if ( row_hi < row_lo ) continue;
      const int e = for_each_list( intra_jobs, [&]( const aa_frame_list & l, int cnt ) {
        LaunchTimer t( ctx, 1 ); return aa::launch_recon_intra_diagonal( l, cnt, d, row_lo, row_hi - row_lo + 1, ctx->compute ); } );
      if ( aa_status st = check( e, "k_recon_intra" ) ) return st;
    }
  }
  // 3. loop filter over the fully reconstructed frames, same diagonal order (loopfilter.cc:133-154 dependencies)
  if ( !lf_jobs.empty() ) {
    for ( int d = 0; d < ndiag; d++ ) {
      const int row_lo = std::max( 0, ( d - ( max_mbw - 1 ) + 1 ) / 2 ), row_hi = std::min( max_mbh - 1, d / 2 );
      if ( row_hi < row_lo ) continue;
      const int e = for_each_list( lf_jobs, [&]( const aa_frame_list & l, int cnt ) {
        LaunchTimer t( ctx, 2 ); return aa::launch_loopfilter_diagonal( l, cnt, d, row_lo, row_hi - row_lo + 1, ctx->compute ); } );
      if ( aa_status st = check( e, "k_loopfilter" ) ) return st;
    }
  }
  advance.ok = true;
  return AA_OK;
}

aa_status aa_stream_decode( aa_stream * s, const uint8_t * data, size_t size, int * frame_index, int * shown )
{
  int fi = -1; aa_frame_header h;
  if ( aa_status st = aa_stream_parse( s, data, size, &fi, &h ) ) return st;
  aa_stream * one[1] = { s };
  if ( aa_status st = aa_decode_batch( s->ctx, one, 1, &fi ) ) return st;
  if ( frame_index ) *frame_index = fi;
  if ( shown ) *shown = h.show_frame;
  return AA_OK;
}

int aa_stream_frame_count( const aa_stream * s ) { return s ? static_cast<int>( s->frames.size() ) : 0; }

aa_status aa_stream_set_error_concealment( aa_stream * s, int on )
{
  if ( !s ) return fail( AA_ERR_ARGUMENT, "null stream" );
  s->parser.set_error_concealment( on != 0 );       // (the host parser and the header pre-pass of the device parser: one Parser)
  return AA_OK;
}
int aa_stream_error_concealment( const aa_stream * s ) { return s && s->parser.error_concealment() ? 1 : 0; }

aa_status aa_stream_release_before( aa_stream * s, int first_kept )
{
  if ( !s ) return fail( AA_ERR_ARGUMENT, "null stream" );
  const int n = std::min<int>( first_kept, static_cast<int>( s->frames.size() ) );
  for ( int i = s->first_live; i < n; i++ ) {
    FrameRec & f = s->frames[i];
    if ( f.handle_held ) { f.handle_held = false; if ( f.placed ) release( s, f.out_slot ); }
    if ( i < s->next_submit ) release_records( s, f, true );       // decoded: nothing will read its records again once queued kernels ran
    std::vector<uint8_t>().swap( f.intra_diagonals );
  }
  while ( s->first_live < n && s->frames[s->first_live].records_released ) s->first_live++;
  return AA_OK;
}

aa_status aa_stream_rewind( aa_stream * s )
{
  if ( !s ) return fail( AA_ERR_ARGUMENT, "null stream" );
  for ( const auto & f : s->frames ) if ( !f.handle_held || f.records_released ) return fail( AA_ERR_LOGIC, "aa_stream_rewind: frames were released; slots may have been reused" );
  s->next_submit = 0;
  return AA_OK;
}

aa_status aa_stream_rewind_to( aa_stream * s, int fi )
{
  if ( !s || fi < 0 || fi > static_cast<int>( s->frames.size() ) ) return fail( AA_ERR_ARGUMENT, "aa_stream_rewind_to: bad argument" );
  if ( fi > s->next_submit ) return fail( AA_ERR_LOGIC, "aa_stream_rewind_to: that frame has not been decoded yet" );
  for ( size_t i = fi; i < s->frames.size(); i++ )
    if ( !s->frames[i].handle_held || s->frames[i].records_released ) return fail( AA_ERR_LOGIC, "aa_stream_rewind_to: frames were released; slots may have been reused" );
  if ( fi < static_cast<int>( s->frames.size() ) && fi > 0 && !s->frames[fi].hdr.key_frame )
    for ( int r : s->frames[fi - 1].ref_after )
      if ( r < 0 || !s->frames[r].handle_held ) return fail( AA_ERR_LOGIC, "aa_stream_rewind_to: a raster that frame predicts from was released" );
  s->next_submit = fi;
  return AA_OK;
}

aa_status aa_stream_download( aa_stream * s, int fi, uint8_t * y, uint8_t * u, uint8_t * v )
{
  if ( !s || fi < 0 || fi >= static_cast<int>( s->frames.size() ) ) return fail( AA_ERR_ARGUMENT, "aa_stream_download: bad frame index" );
  if ( aa_status st = set_device( s->ctx ) ) return st;
  const FrameRec & r = s->frames[fi];
  if ( !r.handle_held ) return fail( AA_ERR_LOGIC, "aa_stream_download: frame was released" );
  if ( fi >= s->next_submit ) return fail( AA_ERR_LOGIC, "aa_stream_download: frame not decoded yet" );
  HIP_TRY( hipStreamSynchronize( s->ctx->compute ) );
  if ( aa_status st = check_watchdog( s->ctx ) ) return st;
  uint8_t * dst[3] = { y, u, v };
  for ( int p = 0; p < 3; p++ ) if ( dst[p] ) HIP_TRY( hipMemcpy( dst[p], slot_plane( s, r.out_slot, p ), s->plane_bytes[p], hipMemcpyDeviceToHost ) );
  return AA_OK;
}

/* Pinned host memory for asynchronous downloads (hipHostMalloc) */
aa_status aa_pinned_alloc( aa_ctx * ctx, size_t bytes, void ** out )
{
  if ( !ctx || !out || !bytes ) return fail( AA_ERR_ARGUMENT, "aa_pinned_alloc: bad argument" );
  if ( aa_status st = set_device( ctx ) ) return st;
  HIP_TRY( hipHostMalloc( out, bytes, hipHostMallocDefault ) );
  return AA_OK;
}
void aa_pinned_free( void * p ) { if ( p ) (void) hipHostFree( p ); }

/* Shown frames on their way out without stalling the decoder: the copy is queued on the COPY stream behind everything the
 * compute stream holds now; the planes (pinned memory) are valid after aa_ctx_sync or aa_stream_download_wait. */
aa_status aa_stream_download_async( aa_stream * s, int fi, uint8_t * y, uint8_t * u, uint8_t * v )
{
  if ( !s || fi < 0 || fi >= static_cast<int>( s->frames.size() ) ) return fail( AA_ERR_ARGUMENT, "aa_stream_download_async: bad frame index" );
  if ( aa_status st = set_device( s->ctx ) ) return st;
  const FrameRec & r = s->frames[fi];
  if ( !r.handle_held ) return fail( AA_ERR_LOGIC, "aa_stream_download_async: frame was released" );
  if ( fi >= s->next_submit || !r.placed ) return fail( AA_ERR_LOGIC, "aa_stream_download_async: frame not decoded yet" );
  aa_ctx * ctx = s->ctx;
  hipEvent_t e = get_event( ctx );
  HIP_TRY( hipEventRecord( e, ctx->compute ) );
  HIP_TRY( hipStreamWaitEvent( ctx->copy, e, 0 ) );
  ctx->free_events.push_back( e );
  uint8_t * dst[3] = { y, u, v };
  // the raster may be released before the copy has run: the epoch that frees it waits for the copy stream as well
  // (the flag stays up until the event recorded behind THIS call's copies has fired: downloads are numbered, and a release on
  // another thread that finds the event of an earlier download complete does not take the flag down while a later one is still
  // between its copies and its event)
  { std::lock_guard<std::mutex> g( ctx->pool_mu ); ctx->copy_reads_rasters = true; ctx->raster_download_pending = true; ++ctx->downloads_queued; }
  // Every exit -- the error exits of the HIP calls below too -- counts this download as recorded and leaves an event behind what it
  // queued on the copy stream: a download that begun and never "recorded" would keep downloads_recorded != downloads_queued for
  // the life of the context, and every raster released from then on would take the slow epoch route.
  struct Recorded {
    aa_ctx * ctx;
    ~Recorded() {
      std::lock_guard<std::mutex> g( ctx->pool_mu );
      if ( !ctx->last_raster_download && hipEventCreateWithFlags( &ctx->last_raster_download, hipEventDisableTiming ) != hipSuccess ) { ctx->last_raster_download = nullptr; (void) hipGetLastError(); }
      if ( ctx->last_raster_download && hipEventRecord( ctx->last_raster_download, ctx->copy ) != hipSuccess ) (void) hipGetLastError();
      ++ctx->downloads_recorded;                         // (a count of finished calls: equal to downloads_queued when none is between its copies and its event)
    }
  } recorded { ctx };
  for ( int p = 0; p < 3; p++ ) if ( dst[p] ) HIP_TRY( hipMemcpyAsync( dst[p], slot_plane( s, r.out_slot, p ), s->plane_bytes[p], hipMemcpyDeviceToHost, ctx->copy ) );
  // (rasters released from now on are recycled through an epoch until this copy is through: dev_free_compute)
  return AA_OK;
}
aa_status aa_stream_download_wait( aa_stream * s )
{
  if ( !s ) return fail( AA_ERR_ARGUMENT, "null stream" );
  if ( aa_status st = set_device( s->ctx ) ) return st;
  HIP_TRY( hipStreamSynchronize( s->ctx->copy ) );
  return check_watchdog( s->ctx );
}

/* A whole frame index of a batch on its way out (frontend/vp8decode.cc:78-93, decode-bundle.cc:92-99 deliver every shown frame): the
 * rasters of n decoded frames -- each three padded planes, contiguous in its pool piece -- are gathered by ONE kernel on the
 * compute stream into one staging piece, and ONE copy on the copy stream takes them to dst + i * stride (pinned memory).  The
 * per-plane form (aa_stream_download_async) costs 3 n copies of 0.5-2 MB: measured 16.8 GB/s over a link that does 50+. */
aa_status aa_download_batch_async( aa_ctx * ctx, aa_stream * const * streams, int n, const int * frame_index, uint8_t * dst, size_t stride )
{
  if ( !ctx || !streams || !frame_index || !dst || n <= 0 ) return fail( AA_ERR_ARGUMENT, "aa_download_batch_async: bad argument" );
  if ( aa_status st = set_device( ctx ) ) return st;
  size_t max_bytes = 0;
  for ( int i = 0; i < n; i++ ) {
    const aa_stream * s = streams[i];
    if ( !s || s->ctx != ctx ) return fail( AA_ERR_ARGUMENT, "aa_download_batch_async: stream belongs to another context" );
    const int fi = frame_index[i];
    if ( fi < 0 || fi >= static_cast<int>( s->frames.size() ) ) return fail( AA_ERR_ARGUMENT, "aa_download_batch_async: bad frame index" );
    const FrameRec & r = s->frames[fi];
    if ( !r.handle_held ) return fail( AA_ERR_LOGIC, "aa_download_batch_async: frame was released" );
    if ( fi >= s->next_submit || !r.placed ) return fail( AA_ERR_LOGIC, "aa_download_batch_async: frame not decoded yet" );
    const size_t bytes = s->plane_bytes[0] + 2 * s->plane_bytes[1];
    if ( bytes > stride ) return fail( AA_ERR_ARGUMENT, "aa_download_batch_async: stride smaller than a raster" );
    max_bytes = std::max( max_bytes, bytes );
  }
  if ( stride & 15 ) return fail( AA_ERR_ARGUMENT, "aa_download_batch_async: stride must be a multiple of 16" );
  aa_ctx::GatherBuf & gb = ctx->gather_bufs[ctx->next_gather_buf];
  ctx->next_gather_buf = ( ctx->next_gather_buf + 1 ) % aa_ctx::kBindBufs;
  if ( gb.busy ) { HIP_TRY( hipEventSynchronize( gb.done ) ); gb.busy = false; }
  if ( gb.cap < static_cast<size_t>( n ) ) {
    if ( gb.host ) (void) hipHostFree( gb.host );
    gb.host = nullptr; gb.dev = nullptr; gb.cap = 0;
    const size_t cap = std::max<size_t>( 512, size_t( n ) * 2 );
    HIP_TRY( hipHostMalloc( reinterpret_cast<void **>( &gb.host ), cap * sizeof( aa_gather_job ), hipHostMallocDefault ) );
    HIP_TRY( hipHostGetDevicePointer( reinterpret_cast<void **>( &gb.dev ), gb.host, 0 ) );
    gb.cap = cap;
  }
  if ( !gb.done ) HIP_TRY( hipEventCreateWithFlags( &gb.done, hipEventDisableTiming ) );
  for ( int i = 0; i < n; i++ ) {
    aa_stream * s = streams[i];
    gb.host[i].src = s->slots[s->frames[frame_index[i]].out_slot].dev;          // (Y, U, V back to back: slot_plane)
    gb.host[i].bytes = s->plane_bytes[0] + 2 * s->plane_bytes[1];
  }
  const size_t total = stride * static_cast<size_t>( n );
  uint8_t * staging = nullptr;
  if ( aa_status st = dev_alloc( ctx, total, &staging ) ) return st;
  // (the piece goes back behind the copy that reads it: the epoch that frees it waits for the copy stream as well)
  struct Back { aa_ctx * c; uint8_t * p; size_t b; ~Back() { { std::lock_guard<std::mutex> g( c->pool_mu ); c->copy_reads_rasters = true; } dev_free( c, p, b, true ); } } back { ctx, staging, total };
  if ( int e = aa::launch_gather_rasters( gb.dev, n, staging, stride, max_bytes, ctx->compute ) ) return hip_fail( static_cast<hipError_t>( e ), "k_gather_rasters" );
  HIP_TRY( hipEventRecord( gb.done, ctx->compute ) );
  gb.busy = true;
  HIP_TRY( hipStreamWaitEvent( ctx->copy, gb.done, 0 ) );
  HIP_TRY( hipMemcpyAsync( dst, staging, total, hipMemcpyDeviceToHost, ctx->copy ) );
  return AA_OK;
}
aa_status aa_ctx_download_wait( aa_ctx * ctx )
{
  if ( !ctx ) return fail( AA_ERR_ARGUMENT, "null ctx" );
  if ( aa_status st = set_device( ctx ) ) return st;
  HIP_TRY( hipStreamSynchronize( ctx->copy ) );
  return check_watchdog( ctx );
}

aa_status aa_stream_raster_device( aa_stream * s, int fi, void ** y, void ** u, void ** v )
{
  if ( !s || fi < 0 || fi >= static_cast<int>( s->frames.size() ) ) return fail( AA_ERR_ARGUMENT, "bad frame index" );
  const FrameRec & r = s->frames[fi];
  if ( y ) *y = slot_plane( s, r.out_slot, 0 );
  if ( u ) *u = slot_plane( s, r.out_slot, 1 );
  if ( v ) *v = slot_plane( s, r.out_slot, 2 );
  return AA_OK;
}

aa_status aa_stream_references( const aa_stream * s, int * last, int * golden, int * alternate )
{
  if ( !s ) return fail( AA_ERR_ARGUMENT, "null stream" );
  int r[3] = { -1, -1, -1 };
  if ( s->next_submit > 0 ) std::memcpy( r, s->frames[s->next_submit - 1].ref_after, sizeof r );
  if ( last ) *last = r[0];
  if ( golden ) *golden = r[1];
  if ( alternate ) *alternate = r[2];
  return AA_OK;
}

aa_status aa_stream_reference_slots( const aa_stream * s, int slots[3] )
{
  if ( !s || !slots ) return fail( AA_ERR_ARGUMENT, "null argument" );
  for ( int i = 0; i < 3; i++ ) slots[i] = s->cur_ref_slot[i];
  return AA_OK;
}

aa_status aa_stream_export_raster( aa_stream * s, int fi, void * y, void * u, void * v )
{
  if ( !s || fi < 0 || fi >= static_cast<int>( s->frames.size() ) || !y || !u || !v ) return fail( AA_ERR_ARGUMENT, "aa_stream_export_raster: bad argument" );
  if ( aa_status st = set_device( s->ctx ) ) return st;
  if ( fi >= s->next_submit ) return fail( AA_ERR_LOGIC, "aa_stream_export_raster: frame not decoded yet" );
  void * dst[3] = { y, u, v };
  for ( int p = 0; p < 3; p++ )
    HIP_TRY( hipMemcpyAsync( dst[p], slot_plane( s, s->frames[fi].out_slot, p ), s->plane_bytes[p], hipMemcpyDeviceToDevice, s->ctx->compute ) );
  return AA_OK;
}
size_t aa_stream_state_size( const aa_stream * s ) { return s ? s->parser.state_size() : 0; }
aa_status aa_stream_export_state( const aa_stream * s, uint8_t * buf, size_t capacity )
{
  if ( !s ) return fail( AA_ERR_ARGUMENT, "null stream" );
  if ( aa_status st = segmap_to_host( const_cast<aa_stream *>( s ) ) ) return st;
  return export_state_common( s->parser, buf, capacity );
}
aa_status aa_stream_import_state( aa_stream * s, const uint8_t * buf, size_t size )
{
  if ( !s ) return fail( AA_ERR_ARGUMENT, "null stream" );
  if ( aa_status st = segmap_to_host( s ) ) return st;      // (also settles ownership: the imported map lives on the host side)
  return import_state_common( s->parser, buf, size );
}

/* Decoder::serialize / Decoder::deserialize (decoder.cc:54-81): [DECODER][u32 len] DecoderState References, where
 * References = [REFERENCES][u32 len][u16 display w][u16 display h][REF_LAST][u32 len] padded Y, U, V of the LAST reference
 * only (decoder.cc:177-197; golden and alternative alias it after loading, :171-175). */
aa_status aa_stream_serialize( aa_stream * s, uint8_t * buf, size_t capacity, size_t * size )
{
  if ( !s ) return fail( AA_ERR_ARGUMENT, "null stream" );
  if ( aa_status st = set_device( s->ctx ) ) return st;
  if ( s->next_submit != static_cast<int>( s->frames.size() ) ) return fail( AA_ERR_LOGIC, "aa_stream_serialize: parsed frames are still waiting to be decoded" );
  const int slot = s->cur_ref_slot[0];
  if ( slot < 0 ) return fail( AA_ERR_LOGIC, "aa_stream_serialize: nothing decoded or imported yet (the reference would write an uninitialised raster)" );
  if ( aa_status st = segmap_to_host( s ) ) return st;
  std::vector<uint8_t> blob;
  blob.push_back( 11 /* DECODER */ );
  blob.insert( blob.end(), 4, 0 );
  s->parser.serialize_reference( blob );
  const size_t raster = s->plane_bytes[0] + 2 * s->plane_bytes[1];
  blob.push_back( 7 /* REFERENCES */ );
  const uint32_t rlen = static_cast<uint32_t>( 4 + 5 + raster );
  for ( int i = 0; i < 4; i++ ) blob.push_back( static_cast<uint8_t>( rlen >> ( 8 * i ) ) );
  const uint16_t dims[2] = { s->parser.width(), s->parser.height() };
  for ( int k = 0; k < 2; k++ ) { blob.push_back( static_cast<uint8_t>( dims[k] ) ); blob.push_back( static_cast<uint8_t>( dims[k] >> 8 ) ); }
  blob.push_back( 8 /* REF_LAST */ );
  for ( int i = 0; i < 4; i++ ) blob.push_back( static_cast<uint8_t>( static_cast<uint32_t>( raster ) >> ( 8 * i ) ) );
  const uint32_t total = static_cast<uint32_t>( blob.size() + raster - 5 );
  for ( int i = 0; i < 4; i++ ) blob[1 + i] = static_cast<uint8_t>( total >> ( 8 * i ) );
  if ( size ) *size = blob.size() + raster;
  if ( !buf ) return AA_OK;                         // size query
  if ( capacity < blob.size() + raster ) return fail( AA_ERR_ARGUMENT, "aa_stream_serialize: buffer too small" );
  std::memcpy( buf, blob.data(), blob.size() );
  HIP_TRY( hipStreamSynchronize( s->ctx->compute ) );
  if ( aa_status st = check_watchdog( s->ctx ) ) return st;
  uint8_t * dst = buf + blob.size();
  for ( int p = 0; p < 3; p++ ) { HIP_TRY( hipMemcpy( dst, slot_plane( s, slot, p ), s->plane_bytes[p], hipMemcpyDeviceToHost ) ); dst += s->plane_bytes[p]; }
  return AA_OK;
}

static aa_status import_common( aa_stream * s, const void * const src[3], hipMemcpyKind kind );
aa_status aa_stream_deserialize( aa_stream * s, const uint8_t * buf, size_t size )
{
  if ( !s || !buf ) return fail( AA_ERR_ARGUMENT, "null argument" );
  if ( s->next_submit != static_cast<int>( s->frames.size() ) ) return fail( AA_ERR_LOGIC, "aa_stream_deserialize: parsed frames are still waiting to be decoded" );
  if ( size < 5 || buf[0] != 11 ) return fail( AA_ERR_INVALID, "invalid decoder state: expected DECODER" );
  if ( aa_status st = segmap_to_host( s ) ) return st;
  const size_t raster = s->plane_bytes[0] + 2 * s->plane_bytes[1];
  aa::Parser trial = s->parser;                     // the stream changes only if the whole blob is good
  size_t at = 5;
  try { at += trial.deserialize_reference( buf + at, size - at ); }
  catch ( const aa::ParseError & e ) { return fail( e.code, e.message ); }
  if ( size < at + 9 + 5 + raster || buf[at] != 7 ) return fail( AA_ERR_INVALID, "invalid decoder state: expected REFERENCES" );
  const unsigned w = buf[at + 5] | ( buf[at + 6] << 8 ), h = buf[at + 7] | ( buf[at + 8] << 8 );
  at += 9;
  if ( w != s->parser.width() || h != s->parser.height() ) return fail( AA_ERR_INVALID, "invalid decoder state: reference size differs from this decoder's" );
  if ( buf[at] != 8 ) return fail( AA_ERR_INVALID, "invalid decoder state: no REF_LAST raster" );
  const size_t rl = buf[at + 1] | ( buf[at + 2] << 8 ) | ( buf[at + 3] << 16 ) | ( static_cast<size_t>( buf[at + 4] ) << 24 );
  at += 5;
  if ( rl != raster || size != at + raster ) return fail( AA_ERR_INVALID, "invalid decoder state: raster length" );
  const void * src[3] = { buf + at, buf + at + s->plane_bytes[0], buf + at + s->plane_bytes[0] + s->plane_bytes[1] };
  if ( aa_status st = import_common( s, src, hipMemcpyHostToDevice ) ) return st;
  s->parser = trial;
  return AA_OK;
}

static aa_status import_common( aa_stream * s, const void * const src[3], hipMemcpyKind kind )
{
  if ( aa_status st = set_device( s->ctx ) ) return st;
  if ( s->next_submit != static_cast<int>( s->frames.size() ) )
    return fail( AA_ERR_LOGIC, "import_reference: parsed frames are still waiting to be decoded" );
  int slot;
  if ( aa_status st = alloc_slot( s, &slot ) ) return st;
  for ( int p = 0; p < 3; p++ )
    HIP_TRY( hipMemcpyAsync( slot_plane( s, slot, p ), src[p], s->plane_bytes[p], kind, s->ctx->compute ) );
  if ( kind == hipMemcpyHostToDevice ) HIP_TRY( hipStreamSynchronize( s->ctx->compute ) );
  for ( int i = 0; i < 3; i++ ) set_ref( s, i, slot, -1 );
  return AA_OK;
}
aa_status aa_stream_import_reference( aa_stream * s, const void * y, const void * u, const void * v )
{
  if ( !s || !y || !u || !v ) return fail( AA_ERR_ARGUMENT, "null argument" );
  const void * src[3] = { y, u, v };
  return import_common( s, src, hipMemcpyDeviceToDevice );
}
aa_status aa_stream_import_reference_host( aa_stream * s, const uint8_t * y, const uint8_t * u, const uint8_t * v )
{
  if ( !s || !y || !u || !v ) return fail( AA_ERR_ARGUMENT, "null argument" );
  const void * src[3] = { y, u, v };
  return import_common( s, src, hipMemcpyHostToDevice );
}

/* ---------------- hashes: boost::hash_combine / hash_range as the reference uses them (pre-1.81 formula) ---------------- */
namespace {
inline void hcombine( uint64_t & seed, uint64_t v ) { seed ^= v + 0x9e3779b9ull + ( seed << 6 ) + ( seed >> 2 ); }
inline void hrange_u8( uint64_t & seed, const uint8_t * p, size_t n ) { for ( size_t i = 0; i < n; i++ ) hcombine( seed, p[i] ); }
inline void hrange_i8( uint64_t & seed, const int8_t * p, size_t n ) { for ( size_t i = 0; i < n; i++ ) hcombine( seed, static_cast<uint64_t>( static_cast<int64_t>( p[i] ) ) ); }

// DecoderState::hash (decoder.cc:266-281) with ProbabilityTables::hash (probability_tables.cc:36-57), Segmentation::hash
// (decoder.cc:379-394; the map is sized by the frame's PIXEL dimensions, decoder.cc:238) and FilterAdjustments::hash
// (decoder.cc:331-340 -- whose second range runs from mode_adjustments.begin() to REF_adjustments.end(), i.e. is empty: the
// mode adjustments are not hashed; kept as is).
uint64_t state_hash( const aa::Parser & ps )
{
  const aa::ProbTables & t = ps.probs();
  uint64_t ph = 0;
  hrange_u8( ph, &t.coeff[0][0][0][0], 1056 ); hrange_u8( ph, t.y_mode, 4 ); hrange_u8( ph, t.uv_mode, 3 ); hrange_u8( ph, &t.mv[0][0], 38 );
  uint64_t h = 0;
  hcombine( h, ps.width() ); hcombine( h, ps.height() ); hcombine( h, ph );
  const aa::SegmentationState & sg = ps.segmentation();
  if ( sg.enabled ) {
    uint64_t sh = 0;
    hcombine( sh, sg.absolute ? 1 : 0 );
    hrange_i8( sh, sg.quant, 4 ); hrange_i8( sh, sg.lf, 4 );
    const unsigned w = ps.width(), hgt = ps.height(), mbw = ps.mb_width(), mbh = ps.mb_height();
    for ( unsigned r = 0; r < hgt; r++ ) for ( unsigned c = 0; c < w; c++ ) hcombine( sh, ( r < mbh && c < mbw ) ? sg.map[size_t( r ) * mbw + c] : 3 );
    hcombine( h, sh );
  }
  const aa::FilterAdjustState & fa = ps.filter_adjustments();
  if ( fa.enabled ) { uint64_t fh = 0; hrange_i8( fh, fa.ref, 4 ); hcombine( h, fh ); }
  return h;
}

// BaseRaster::raw_hash (raster.cc:52-61) of a raster slot; the recurrence is serial, so the planes come to the host
aa_status slot_hash( aa_stream * s, int slot, uint64_t * out )
{
  Slot & sl = s->slots[slot];
  if ( !sl.hash_valid ) {
    HIP_TRY( hipStreamSynchronize( s->ctx->compute ) );
    if ( aa_status st = check_watchdog( s->ctx ) ) return st;
    std::vector<uint8_t> host( s->plane_bytes[0] + 2 * s->plane_bytes[1] );
    HIP_TRY( hipMemcpy( host.data(), sl.dev, host.size(), hipMemcpyDeviceToHost ) );
    uint64_t h = 0;
    hrange_u8( h, host.data(), host.size() );
    sl.hash = h; sl.hash_valid = true;
  }
  *out = sl.hash;
  return AA_OK;
}
} // namespace

aa_status aa_parser_state_hash( const aa_parser * p, uint64_t * out )
{
  if ( !p || !out ) return fail( AA_ERR_ARGUMENT, "null argument" );
  *out = state_hash( p->impl );
  return AA_OK;
}
aa_status aa_stream_state_hash( aa_stream * s, uint64_t * out )
{
  if ( !s || !out ) return fail( AA_ERR_ARGUMENT, "null argument" );
  if ( aa_status st = set_device( s->ctx ) ) return st;
  if ( aa_status st = segmap_to_host( s ) ) return st;
  *out = state_hash( s->parser );
  return AA_OK;
}
aa_status aa_stream_raster_hash( aa_stream * s, int fi, uint64_t * out )
{
  if ( !s || !out || fi < 0 || fi >= static_cast<int>( s->frames.size() ) ) return fail( AA_ERR_ARGUMENT, "aa_stream_raster_hash: bad argument" );
  if ( aa_status st = set_device( s->ctx ) ) return st;
  const FrameRec & r = s->frames[fi];
  if ( fi >= s->next_submit || !r.placed ) return fail( AA_ERR_LOGIC, "aa_stream_raster_hash: frame not decoded yet" );
  if ( !r.handle_held ) return fail( AA_ERR_LOGIC, "aa_stream_raster_hash: frame was released" );
  return slot_hash( s, r.out_slot, out );
}
/* DecoderHash (decoder.cc:143-153,482-490): state, last, golden, alternative; everything parsed must have been submitted */
aa_status aa_stream_decoder_hash( aa_stream * s, uint64_t parts[4], uint64_t * whole )
{
  if ( !s ) return fail( AA_ERR_ARGUMENT, "null stream" );
  if ( aa_status st = set_device( s->ctx ) ) return st;
  if ( s->next_submit != static_cast<int>( s->frames.size() ) ) return fail( AA_ERR_LOGIC, "aa_stream_decoder_hash: parsed frames are still waiting to be decoded" );
  if ( aa_status st = segmap_to_host( s ) ) return st;
  uint64_t h[4];
  h[0] = state_hash( s->parser );
  for ( int i = 0; i < 3; i++ ) if ( aa_status st = slot_hash( s, s->cur_ref_slot[i], &h[1 + i] ) ) return st;
  if ( parts ) std::memcpy( parts, h, sizeof h );
  if ( whole ) { uint64_t w = 0; for ( int i = 0; i < 4; i++ ) hcombine( w, h[i] ); *whole = w; }
  return AA_OK;
}
aa_status aa_stream_minihash( aa_stream * s, uint32_t * out )
{
  if ( !out ) return fail( AA_ERR_ARGUMENT, "null argument" );
  uint64_t w = 0;
  if ( aa_status st = aa_stream_decoder_hash( s, nullptr, &w ) ) return st;
  *out = static_cast<uint32_t>( w );
  return AA_OK;
}

/* One frame's raster handle and records (what a dying RasterHandle gives back, raster_handle.cc:113-122) */
aa_status aa_stream_release_frame( aa_stream * s, int fi )
{
  if ( !s || fi < 0 || fi >= static_cast<int>( s->frames.size() ) ) return fail( AA_ERR_ARGUMENT, "aa_stream_release_frame: bad frame index" );
  FrameRec & f = s->frames[fi];
  if ( f.handle_held ) { f.handle_held = false; if ( f.placed ) release( s, f.out_slot ); }
  if ( fi < s->next_submit ) release_records( s, f, true );
  std::vector<uint8_t>().swap( f.intra_diagonals );
  while ( s->first_live < static_cast<int>( s->frames.size() ) && s->frames[s->first_live].records_released && !s->frames[s->first_live].handle_held ) s->first_live++;
  return AA_OK;
}

/* References( last, golden, alternative ): Decoder( DecoderState, References ) (decoder.cc:43-46).  planes[i] = Y, U, V of
 * reference i, in HBM (is_host[i] == 0) or on the host; references whose Y pointers are equal become one raster here too. */
aa_status aa_stream_set_references( aa_stream * s, const void * const planes[3][3], const int is_host[3] )
{
  if ( !s || !planes || !is_host ) return fail( AA_ERR_ARGUMENT, "null argument" );
  if ( aa_status st = set_device( s->ctx ) ) return st;
  if ( s->next_submit != static_cast<int>( s->frames.size() ) ) return fail( AA_ERR_LOGIC, "aa_stream_set_references: parsed frames are still waiting to be decoded" );
  int slot[3] = { -1, -1, -1 };
  bool any_host = false;
  for ( int i = 0; i < 3; i++ ) {
    for ( int p = 0; p < 3; p++ ) if ( !planes[i][p] ) return fail( AA_ERR_ARGUMENT, "aa_stream_set_references: null plane" );
    for ( int k = 0; k < i; k++ ) if ( planes[k][0] == planes[i][0] && is_host[k] == is_host[i] ) slot[i] = slot[k];
    if ( slot[i] >= 0 ) continue;
    if ( aa_status st = alloc_slot( s, &slot[i] ) ) return st;
    retain( s, slot[i] );                      // (held while the three are being set up)
    for ( int p = 0; p < 3; p++ )
      HIP_TRY( hipMemcpyAsync( slot_plane( s, slot[i], p ), planes[i][p], s->plane_bytes[p], is_host[i] ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice, s->ctx->compute ) );
    any_host = any_host || is_host[i];
  }
  if ( any_host ) HIP_TRY( hipStreamSynchronize( s->ctx->compute ) );       // the caller's host planes may go away
  for ( int i = 0; i < 3; i++ ) set_ref( s, i, slot[i], -1 );
  for ( int i = 0; i < 3; i++ ) { bool first = true; for ( int k = 0; k < i; k++ ) if ( slot[k] == slot[i] ) first = false; if ( first ) release( s, slot[i] ); }
  return AA_OK;
}
namespace {
// x264's pixel_ssim_wxh over per-window terms (floats, four windows at a time, row by row) / count, for n candidates; then the
// reference's choice: ascending levels, the first one that does not improve on the best so far ends the search.
void score_candidates( const std::vector<float> & win, int n, int w4, int h4, int level_lo, int * best_level, double * best_ssim, double * ssim_out )
{
  const size_t windows = size_t( w4 - 1 ) * ( h4 - 1 );
  int best = level_lo; double best_q = -1.0; bool searching = true;
  for ( int i = 0; i < n; i++ ) {
    float total = 0.0f;
    for ( int y = 0; y < h4 - 1; y++ )
      for ( int x = 0; x < w4 - 1; x += 4 ) {
        float part = 0.0f;
        for ( int k = x; k < std::min( x + 4, w4 - 1 ); k++ ) part += win[windows * i + size_t( y ) * ( w4 - 1 ) + k];
        total += part;
      }
    const double q = static_cast<double>( total ) / static_cast<double>( windows );
    if ( ssim_out ) ssim_out[i] = q;
    if ( searching ) { if ( q > best_q ) { best_q = q; best = level_lo + i; } else searching = false; }
  }
  if ( best_level ) *best_level = best;
  if ( best_ssim ) *best_ssim = best_q;
}
uint32_t segment_levels( const aa::SegmentationState & seg, int level )      // frame.cc:144-166 + the clamp of macroblock.cc:611-623
{
  uint32_t word = 0;
  for ( int k = 0; k < 4; k++ ) {
    const int v = level ? ( seg.enabled ? seg.lf[k] + ( seg.absolute ? 0 : level ) : level ) : 0;
    word |= static_cast<uint32_t>( v <= 0 ? 0 : ( v > 63 ? 63 : v ) ) << ( 8 * k );
  }
  return word;
}
} // namespace

/* BaseRaster::quality on HOST planes (util/raster.cc:63-66 -> util/ssim.cc:57-71 -> libx264's pixel_ssim_wxh): the same
 * measure aa_stream_lf_search computes on the device, for callers that hold rasters in host memory (VP8Raster::quality in the
 * shim).  Plain host arithmetic -- the reference does this on the CPU too. */
aa_status aa_ssim_host( const uint8_t * a, const uint8_t * b, int width, int height, double * out )
{
  if ( !a || !b || !out || width < 8 || height < 8 ) return fail( AA_ERR_ARGUMENT, "aa_ssim_host: null plane or a plane smaller than one 8x8 window" );
  const int w4 = width >> 2, h4 = height >> 2;
  // sums over every 4x4 block: pixels of a, pixels of b, squares of both, products
  std::vector<int> blocks( size_t( w4 ) * h4 * 4 );
  for ( int by = 0; by < h4; by++ )
    for ( int bx = 0; bx < w4; bx++ ) {
      int s1 = 0, s2 = 0, ss = 0, s12 = 0;
      for ( int y = 0; y < 4; y++ ) {
        const uint8_t * pa = a + size_t( 4 * by + y ) * width + 4 * bx, * pb = b + size_t( 4 * by + y ) * width + 4 * bx;
        for ( int x = 0; x < 4; x++ ) { const int p = pa[x], q = pb[x]; s1 += p; s2 += q; ss += p * p + q * q; s12 += p * q; }
      }
      int * t = &blocks[( size_t( by ) * w4 + bx ) * 4];
      t[0] = s1; t[1] = s2; t[2] = ss; t[3] = s12;
    }
  // one term per 8x8 window (2x2 blocks, windows step by 4 pixels), in single precision as x264's ssim_end1
  std::vector<float> win( size_t( w4 - 1 ) * ( h4 - 1 ) );
  for ( int y = 0; y < h4 - 1; y++ )
    for ( int x = 0; x < w4 - 1; x++ ) {
      int v[4];
      for ( int k = 0; k < 4; k++ )
        v[k] = blocks[( size_t( y ) * w4 + x ) * 4 + k] + blocks[( size_t( y ) * w4 + x + 1 ) * 4 + k]
               + blocks[( size_t( y + 1 ) * w4 + x ) * 4 + k] + blocks[( size_t( y + 1 ) * w4 + x + 1 ) * 4 + k];
      const int c1 = 416, c2 = 235963;           // (int)(.01*.01*255*255*64 + .5), (int)(.03*.03*255*255*64*63 + .5)
      const int vars = v[2] * 64 - v[0] * v[0] - v[1] * v[1], covar = v[3] * 64 - v[0] * v[1];
      const float num = static_cast<float>( 2 * v[0] * v[1] + c1 ) * static_cast<float>( 2 * covar + c2 );
      const float den = static_cast<float>( v[0] * v[0] + v[1] * v[1] + c1 ) * static_cast<float>( vars + c2 );
      win[size_t( y ) * ( w4 - 1 ) + x] = num / den;
    }
  score_candidates( win, 1, w4, h4, 0, nullptr, nullptr, out );
  return AA_OK;
}


static aa_status lf_search_by_decoders( aa_stream * s, const uint8_t * data, size_t size, const uint8_t * original_luma,
                                        int level_lo, int level_hi, int * best_level, double * best_ssim, double * ssim_out, uint8_t * rasters_out )
{
  if ( !s || !data || !original_luma ) return fail( AA_ERR_ARGUMENT, "aa_stream_lf_search: null argument" );
  if ( level_lo < 0 || level_hi > 63 || level_lo > level_hi ) return fail( AA_ERR_ARGUMENT, "aa_stream_lf_search: levels must be 0 <= lo <= hi <= 63" );
  aa_ctx * ctx = s->ctx;
  if ( aa_status st = set_device( ctx ) ) return st;
  if ( s->next_submit != static_cast<int>( s->frames.size() ) ) return fail( AA_ERR_LOGIC, "aa_stream_lf_search: parsed frames are still waiting to be decoded" );
  if ( aa_status st = segmap_to_host( s ) ) return st;
  const int n = level_hi - level_lo + 1;
  std::vector<aa_stream *> cand( n, nullptr );
  std::vector<int> fis( n, -1 );
  struct Cleanup { std::vector<aa_stream *> & v; uint8_t * orig = nullptr; float * win = nullptr;
                   ~Cleanup() { for ( aa_stream * c : v ) if ( c ) aa_stream_destroy( c ); if ( orig ) (void) hipFree( orig ); if ( win ) (void) hipFree( win ); } } cleanup { cand };
  const void * planes[3][3]; const int on_device[3] = { 0, 0, 0 };
  for ( int r = 0; r < 3; r++ ) for ( int p = 0; p < 3; p++ ) planes[r][p] = slot_plane( s, s->cur_ref_slot[r], p );
  const size_t job_bytes = align_up( sizeof( aa_dev_frame ) );
  for ( int i = 0; i < n; i++ ) {
    const int level = level_lo + i;
    if ( aa_status st = aa_stream_create( ctx, s->parser.width(), s->parser.height(), &cand[i] ) ) return st;
    aa_stream * c = cand[i];
    c->parser = s->parser;                                               // DecoderState as it stands before this frame
    if ( aa_status st = aa_stream_set_references( c, planes, on_device ) ) return st;
    if ( aa_status st = aa_stream_parse( c, data, size, &fis[i], nullptr ) ) return st;
    FrameRec & r = c->frames[fis[i]];
    // the candidate's header: this level, adjustments present and zero.  Per macroblock: the segment's level (frame.cc:144-166)
    // clamped to 0..63 (macroblock.cc:611-623), nothing added (loopfilter.cc:59-79 with zero adjustments).
    const uint32_t seg_level = segment_levels( c->parser.segmentation(), level );
    aa_mb_info * mbs = reinterpret_cast<aa_mb_info *>( reinterpret_cast<uint8_t *>( r.host_job ) + job_bytes );
    for ( unsigned m = 0; m < r.hdr.num_macroblocks; m++ ) mbs[m].lf_level = static_cast<uint8_t>( ( seg_level >> ( 8 * ( mbs[m].segment_id & 3 ) ) ) & 255u );
    r.hdr.loop_filter_level = static_cast<uint8_t>( level );
    r.host_job->loop_filter_level = static_cast<uint8_t>( level );
  }
  if ( aa_status st = aa_decode_batch( ctx, cand.data(), n, fis.data() ) ) return st;

  // quality of every candidate against the original: per-window terms on the device, x264's summation order on the host
  const int pw = s->pw, ph = s->ph, w4 = pw >> 2, h4 = ph >> 2;
  const size_t windows = size_t( w4 - 1 ) * ( h4 - 1 );
  HIP_TRY( hipMalloc( reinterpret_cast<void **>( &cleanup.orig ), s->plane_bytes[0] ) );
  HIP_TRY( hipMalloc( reinterpret_cast<void **>( &cleanup.win ), windows * n * sizeof( float ) ) );
  HIP_TRY( hipMemcpyAsync( cleanup.orig, original_luma, s->plane_bytes[0], hipMemcpyHostToDevice, ctx->compute ) );
  for ( int i = 0; i < n; i++ ) {
    const FrameRec & r = cand[i]->frames[fis[i]];
    if ( const int e = aa::launch_ssim_windows( slot_plane( cand[i], r.out_slot, 0 ), cleanup.orig, pw, ph, cleanup.win + windows * i, ctx->compute ) )
      return hip_fail( static_cast<hipError_t>( e ), "k_ssim_windows" );
  }
  std::vector<float> win( windows * n );
  HIP_TRY( hipMemcpyAsync( win.data(), cleanup.win, win.size() * sizeof( float ), hipMemcpyDeviceToHost, ctx->compute ) );
  HIP_TRY( hipStreamSynchronize( ctx->compute ) );
  if ( aa_status st = check_watchdog( ctx ) ) return st;
  score_candidates( win, n, w4, h4, level_lo, best_level, best_ssim, ssim_out );
  if ( rasters_out )
    for ( int i = 0; i < n; i++ ) {
      uint8_t * dst = rasters_out + size_t( i ) * ( s->plane_bytes[0] + 2 * s->plane_bytes[1] );
      if ( aa_status st = aa_stream_download( cand[i], fis[i], dst, dst + s->plane_bytes[0], dst + s->plane_bytes[0] + s->plane_bytes[1] ) ) return st;
    }
  return AA_OK;
}

/* Encoder feedback, the loop-filter level search of Encoder::apply_best_loopfilter_settings
 * (encoder.cc:459-516): the frame as serialised with ANY level is parsed and reconstructed ONCE, unfiltered, on a scratch copy
 * of the decoder (state + references; the stream itself is untouched).  Every candidate level then is an independent frame
 * for the loop filter: a copy of the unfiltered raster, the macroblock records with the candidate's per-segment level and zero
 * mode / reference adjustments (`filter_adjustments.reset( frame.header() )` after the adjustments were zeroed,
 * encoder.cc:464-470) -- ONE k_loopfilter_rows4 launch over all candidates -- scored with BaseRaster::quality = x264's SSIM of
 * the padded luma planes (util/raster.cc:63-66, util/ssim.cc:57-71).  Selection as in the reference: levels in ascending
 * order, the first one that does not improve on the best so far ends the search (encoder.cc:489-503).
 * (With the diagonal schedule -- no row-pipelined filter -- every candidate is decoded by a scratch decoder of its own.) */
aa_status aa_stream_lf_search( aa_stream * s, const uint8_t * data, size_t size, const uint8_t * original_luma,
                               int level_lo, int level_hi, int * best_level, double * best_ssim, double * ssim_out, uint8_t * rasters_out )
{
  if ( !s || !data || !original_luma ) return fail( AA_ERR_ARGUMENT, "aa_stream_lf_search: null argument" );
  if ( level_lo < 0 || level_hi > 63 || level_lo > level_hi ) return fail( AA_ERR_ARGUMENT, "aa_stream_lf_search: levels must be 0 <= lo <= hi <= 63" );
  aa_ctx * ctx = s->ctx;
  if ( aa_status st = set_device( ctx ) ) return st;
  if ( s->next_submit != static_cast<int>( s->frames.size() ) ) return fail( AA_ERR_LOGIC, "aa_stream_lf_search: parsed frames are still waiting to be decoded" );
  if ( ctx->schedule != 0 ) return lf_search_by_decoders( s, data, size, original_luma, level_lo, level_hi, best_level, best_ssim, ssim_out, rasters_out );
  if ( aa_status st = segmap_to_host( s ) ) return st;
  const int n = level_hi - level_lo + 1;

  // ---- the unfiltered reconstruction, on a scratch decoder ----
  aa_stream * x = nullptr;
  struct Cleanup { aa_ctx * ctx; aa_stream *& x; uint8_t * dev = nullptr;
                   ~Cleanup() { (void) hipStreamSynchronize( ctx->compute ); if ( x ) aa_stream_destroy( x ); if ( dev ) (void) hipFree( dev ); } } cleanup { ctx, x };
  if ( aa_status st = aa_stream_create( ctx, s->parser.width(), s->parser.height(), &x ) ) return st;
  x->parser = s->parser;                                                 // DecoderState as it stands before this frame
  const void * planes[3][3]; const int on_device[3] = { 0, 0, 0 };
  for ( int r = 0; r < 3; r++ ) for ( int p = 0; p < 3; p++ ) planes[r][p] = slot_plane( s, s->cur_ref_slot[r], p );
  if ( aa_status st = aa_stream_set_references( x, planes, on_device ) ) return st;
  int fx = -1;
  if ( aa_status st = aa_stream_parse( x, data, size, &fx, nullptr ) ) return st;
  {
    FrameRec & r = x->frames[fx];
    r.hdr.loop_filter_level = 0; r.host_job->loop_filter_level = 0;      // Frame::decode only (frame.cc:208-250)
  }
  if ( aa_status st = aa_decode_batch( ctx, &x, 1, &fx ) ) return st;
  const FrameRec & rx = x->frames[fx];
  const aa_dev_frame job_x = *rx.host_job;
  const unsigned nmb = rx.hdr.num_macroblocks;
  const uint8_t * unfiltered = slot_plane( x, rx.out_slot, 0 );
  const size_t raster_bytes = s->plane_bytes[0] + 2 * s->plane_bytes[1];

  // ---- per candidate: raster | macroblock records | job; then the original and the SSIM terms ----
  const size_t mb_bytes = align_up( size_t( nmb ) * sizeof( aa_mb_info ) ), job_bytes = align_up( sizeof( aa_dev_frame ) );
  const size_t per = align_up( raster_bytes ) + mb_bytes + job_bytes;
  const int pw = s->pw, ph = s->ph, w4 = pw >> 2, h4 = ph >> 2;
  const size_t windows = size_t( w4 - 1 ) * ( h4 - 1 );
  const size_t orig_off = per * n, win_off = orig_off + align_up( s->plane_bytes[0] );
  HIP_TRY( hipMalloc( reinterpret_cast<void **>( &cleanup.dev ), win_off + windows * n * sizeof( float ) ) );
  uint8_t * base = cleanup.dev;
  std::vector<aa_dev_frame> jobs( n );
  std::vector<std::pair<uint32_t, const aa_dev_frame *>> keyed;
  std::vector<const uint8_t *> luma( n );
  for ( int i = 0; i < n; i++ ) {
    const int level = level_lo + i;
    uint8_t * raster = base + per * i;
    aa_mb_info * mbs = reinterpret_cast<aa_mb_info *>( raster + align_up( raster_bytes ) );
    aa_dev_frame * job_dev = reinterpret_cast<aa_dev_frame *>( reinterpret_cast<uint8_t *>( mbs ) + mb_bytes );
    luma[i] = level ? raster : unfiltered;                               // level 0: Frame::loopfilter does nothing (frame.cc:144)
    if ( !level ) continue;
    HIP_TRY( hipMemcpyAsync( raster, unfiltered, raster_bytes, hipMemcpyDeviceToDevice, ctx->compute ) );
    if ( const int e = aa::launch_lf_relevel( job_x.mbs, mbs, nmb, segment_levels( x->parser.segmentation(), level ), ctx->compute ) )
      return hip_fail( static_cast<hipError_t>( e ), "k_lf_relevel" );
    jobs[i] = job_x;
    jobs[i].cur[0] = raster; jobs[i].cur[1] = raster + s->plane_bytes[0]; jobs[i].cur[2] = raster + s->plane_bytes[0] + s->plane_bytes[1];
    jobs[i].mbs = mbs;
    jobs[i].loop_filter_level = static_cast<uint8_t>( level );
    HIP_TRY( hipMemcpyAsync( job_dev, &jobs[i], sizeof( aa_dev_frame ), hipMemcpyHostToDevice, ctx->compute ) );
    keyed.emplace_back( ( static_cast<uint32_t>( rx.hdr.mb_width ) << 16 ) | rx.hdr.mb_height, job_dev );
  }
  if ( !keyed.empty() ) if ( aa_status st = launch_lf_rows( ctx, keyed, true, rx.hdr.mb_height, rx.hdr.mb_width ) ) return st;
  uint8_t * orig_dev = base + orig_off;
  float * win_dev = reinterpret_cast<float *>( base + win_off );
  HIP_TRY( hipMemcpyAsync( orig_dev, original_luma, s->plane_bytes[0], hipMemcpyHostToDevice, ctx->compute ) );
  for ( int i = 0; i < n; i++ )
    if ( const int e = aa::launch_ssim_windows( luma[i], orig_dev, pw, ph, win_dev + windows * i, ctx->compute ) ) return hip_fail( static_cast<hipError_t>( e ), "k_ssim_windows" );
  std::vector<float> win( windows * n );
  HIP_TRY( hipMemcpyAsync( win.data(), win_dev, win.size() * sizeof( float ), hipMemcpyDeviceToHost, ctx->compute ) );
  HIP_TRY( hipStreamSynchronize( ctx->compute ) );
  if ( aa_status st = check_watchdog( ctx ) ) return st;
  score_candidates( win, n, w4, h4, level_lo, best_level, best_ssim, ssim_out );
  if ( rasters_out )
    for ( int i = 0; i < n; i++ ) HIP_TRY( hipMemcpy( rasters_out + raster_bytes * i, luma[i], raster_bytes, hipMemcpyDeviceToHost ) );
  return AA_OK;
}

/* Device planes of References::last / golden / alternative as they stand (which: 0, 1, 2) */
aa_status aa_stream_reference_device( aa_stream * s, int which, void ** y, void ** u, void ** v )
{
  if ( !s || which < 0 || which > 2 ) return fail( AA_ERR_ARGUMENT, "aa_stream_reference_device: bad argument" );
  const int slot = s->cur_ref_slot[which];
  if ( y ) *y = slot_plane( s, slot, 0 );
  if ( u ) *u = slot_plane( s, slot, 1 );
  if ( v ) *v = slot_plane( s, slot, 2 );
  return AA_OK;
}
aa_status aa_stream_reference_download( aa_stream * s, int which, uint8_t * y, uint8_t * u, uint8_t * v )
{
  if ( !s || which < 0 || which > 2 ) return fail( AA_ERR_ARGUMENT, "aa_stream_reference_download: bad argument" );
  if ( aa_status st = set_device( s->ctx ) ) return st;
  HIP_TRY( hipStreamSynchronize( s->ctx->compute ) );
  if ( aa_status st = check_watchdog( s->ctx ) ) return st;
  uint8_t * dst[3] = { y, u, v };
  for ( int p = 0; p < 3; p++ ) if ( dst[p] ) HIP_TRY( hipMemcpy( dst[p], slot_plane( s, s->cur_ref_slot[which], p ), s->plane_bytes[p], hipMemcpyDeviceToHost ) );
  return AA_OK;
}

} // extern "C"
