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

// ONE translation unit, nine source files (round 6; the reference does the same: macroblock.cc #includes tokens.cc, transform.cc, ...):
// the pieces share file-local state (the anonymous-namespace helpers, g_last_error) and are #included in dependency order.
#include "runtime_types.inc"
#include "runtime_pool.inc"
#include "runtime_tokens.inc"
#include "runtime_records.inc"
#include "runtime_ctx.inc"
#include "runtime_submit.inc"
#include "runtime_decode.inc"
#include "runtime_rasters.inc"
#include "runtime_lf_search.inc"
