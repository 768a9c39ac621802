"""ctypes binding of the C ABI (include/alfalfa_amd.h) -- the same stub a cgo/JNI/N-API binding would write.

The library is built in-tree by alfalfa_amd.build (hipcc, gfx950).  There is no Python or CPU
implementation behind these calls: if the library is missing or has no GPU to run on, they raise.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

AA_OK = 0
ERR_NAMES = {-1: "Invalid", -2: "Unsupported", -3: "LogicError", -4: "OutOfRange", -5: "HipError",
             -6: "NoDevice", -7: "BadArgument", -8: "NoMemory"}

AA_MB_HAS_NONZERO, AA_MB_HAS_Y2, AA_MB_INTER, AA_MB_SKIP, AA_MB_LF_SKIP_INNER = 1, 2, 4, 8, 16

MB_INFO_DTYPE = np.dtype([("y_mode", "u1"), ("uv_mode", "u1"), ("ref_frame", "u1"), ("segment_id", "u1"),
                          ("flags", "u1"), ("lf_level", "u1"), ("split_partition", "u1"), ("reserved", "u1"),
                          ("nz_mask", "<u4"), ("coeff_index", "<u4"), ("u", "u1", (64,))])
assert MB_INFO_DTYPE.itemsize == 80


class FrameHeader(C.Structure):
    _fields_ = [(n, C.c_uint8) for n in (
        "key_frame", "show_frame", "loop_filter_level", "sharpness_level", "num_dct_partitions",
        "segmentation_enabled", "filter_adjustments_enabled", "refresh_last", "refresh_golden",
        "refresh_alternate", "copy_buffer_to_golden", "copy_buffer_to_alternate", "sign_bias_golden",
        "sign_bias_alternate", "q_index", "has_intra_mb")] + [
        ("mb_width", C.c_uint16), ("mb_height", C.c_uint16), ("width", C.c_uint16), ("height", C.c_uint16),
        ("quant", (C.c_uint16 * 6) * 4), ("num_macroblocks", C.c_uint32), ("num_coeff_blocks", C.c_uint32),
        ("num_intra_mbs", C.c_uint32), ("compressed_size", C.c_uint32)]

    def as_dict(self):
        d = {n: getattr(self, n) for n, _ in self._fields_ if n != "quant"}
        d["quant"] = [[self.quant[s][k] for k in range(6)] for s in range(4)]
        return d


class FrameIn(C.Structure):
    _fields_ = [("stream", C.c_void_p), ("data", C.c_char_p), ("size", C.c_size_t)]


class KernelStats(C.Structure):
    _fields_ = [("recon_inter_ms", C.c_double), ("recon_intra_ms", C.c_double), ("loopfilter_ms", C.c_double),
                ("recon_inter_launches", C.c_uint64), ("recon_intra_launches", C.c_uint64),
                ("loopfilter_launches", C.c_uint64), ("macroblocks", C.c_uint64),
                ("parse_headers_ms", C.c_double), ("parse_tokens_ms", C.c_double), ("parse_launches", C.c_uint64),
                ("parsed_macroblocks", C.c_uint64), ("recon_split_ms", C.c_double), ("recon_split_launches", C.c_uint64),
                ("pool_waits", C.c_uint64), ("pool_wait_ms", C.c_double), ("parse_wait_ms", C.c_double),
                ("bind_wait_ms", C.c_double), ("alloc_ms", C.c_double), ("slab_mallocs", C.c_uint64),
                ("token_steps", C.c_uint64), ("token_frames", C.c_uint64), ("worker_launches", C.c_uint64), ("worker_wgs", C.c_uint64),
                ("worker_retires", C.c_uint64), ("heap_grows", C.c_uint64), ("heap_mapped_bytes", C.c_uint64), ("nomem_retries", C.c_uint64), ("frames_evicted", C.c_uint64), ("host_routed_frames", C.c_uint64),
                ("row_handoff_stale_polls", C.c_uint64), ("row_handoff_rereads", C.c_uint64), ("packed_frames", C.c_uint64), ("packed_words", C.c_uint64), ("packed_blocks", C.c_uint64),
                ("host_batch_ms", C.c_double), ("host_batch_parse_wall_ms", C.c_double), ("host_batch_parse_cpu_ms", C.c_double), ("host_batch_arena_ms", C.c_double),
                ("pinned_allocs", C.c_uint64)]


class CtxInfo(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("memory_limit_bytes", "pool_bytes", "pool_free_bytes", "pool_pending_bytes", "heap_mapped_bytes", "heap_limit_bytes", "heap_used_bytes",
                                          "pinned_host_bytes")] + [
        (n, C.c_uint32) for n in ("heap_is_virtual", "token_lanes_per_workgroup", "token_workgroups_capacity", "token_workgroups_alive",
                                  "token_lane_lds_bytes", "token_workgroup_lds_bytes", "jobs_waiting", "compute_units")] + [
        ("heap_free_chunks", C.c_int32), ("lanes_starved", C.c_uint32), ("token_profile", C.c_uint64 * 8),
        ("packed_coefficients", C.c_uint32), ("lane_per_partition", C.c_uint32), ("clock_mhz", C.c_uint32), ("host_share_ms", C.c_uint32), ("host_rate_kb_per_ms", C.c_uint32), ("reserved1", C.c_uint32), ("stream_concurrency", C.c_uint32), ("streams_needed", C.c_uint32),
        ("host_waited_parse_ms", C.c_uint32), ("host_waited_compute_ms", C.c_uint32)]


class AlfalfaError(RuntimeError):
    """Mirrors the reference's exception types (exception.hh:76-98) by name in `.kind`."""

    def __init__(self, code, message):
        super().__init__("%s: %s" % (ERR_NAMES.get(code, str(code)), message))
        self.code, self.kind, self.message = code, ERR_NAMES.get(code, str(code)), message


# every entry point declared in include/alfalfa_amd.h: (name, restype, argtypes)
_P = C.c_void_p
_U8P = C.POINTER(C.c_uint8)
SYMBOLS = [
    ("aa_last_error", C.c_char_p, []), ("aa_abi_version", C.c_int, []), ("aa_runtime_prepare", C.c_int, []), ("aa_host_cpus", C.c_int, []), ("aa_device_count", C.c_int, []),
    ("aa_parser_create", C.c_int, [C.c_uint16, C.c_uint16, C.POINTER(_P)]), ("aa_parser_destroy", None, [_P]),
    ("aa_parser_parse", C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(FrameHeader), _P, _P]),
    ("aa_parser_get_probs", C.c_int, [_P, _U8P]), ("aa_parser_set_error_concealment", C.c_int, [_P, C.c_int]),
    ("aa_parse_frame_tag", C.c_int, [C.c_char_p, C.c_size_t, C.c_uint16, C.c_uint16, C.c_int] + [C.POINTER(C.c_int)] * 4),
    ("aa_stream_set_error_concealment", C.c_int, [_P, C.c_int]), ("aa_stream_error_concealment", C.c_int, [_P]),
    ("aa_parser_get_segmentation", C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int8), C.POINTER(C.c_int8), _U8P]),
    ("aa_parser_get_filter_adjustments", C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int8), C.POINTER(C.c_int8)]),
    ("aa_parser_serialize_state", C.c_int, [_P, _P, C.c_size_t, _P]), ("aa_parser_deserialize_state", C.c_int, [_P, _P, C.c_size_t]),
    ("aa_stream_serialize", C.c_int, [_P, _P, C.c_size_t, _P]), ("aa_stream_deserialize", C.c_int, [_P, _P, C.c_size_t]),
    ("aa_parser_state_size", C.c_size_t, [_P]), ("aa_parser_export_state", C.c_int, [_P, _P, C.c_size_t]),
    ("aa_parser_import_state", C.c_int, [_P, C.c_char_p, C.c_size_t]),
    ("aa_stream_state_size", C.c_size_t, [_P]), ("aa_stream_export_state", C.c_int, [_P, _P, C.c_size_t]),
    ("aa_stream_import_state", C.c_int, [_P, C.c_char_p, C.c_size_t]),
    ("aa_stream_export_raster", C.c_int, [_P, C.c_int, _P, _P, _P]),
    ("aa_ctx_create", C.c_int, [C.c_int, C.POINTER(_P)]), ("aa_ctx_destroy", None, [_P]), ("aa_ctx_sync", C.c_int, [_P]),
    ("aa_ctx_memory", C.c_int, [_P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    ("aa_ctx_set_memory_limit", C.c_int, [_P, C.c_size_t]), ("aa_ctx_set_packed_coefficients", C.c_int, [_P, C.c_int]), ("aa_ctx_set_host_share_ms", C.c_int, [_P, C.c_double]), ("aa_ctx_set_lane_per_partition", C.c_int, [_P, C.c_int]), ("aa_ctx_get_info", C.c_int, [_P, C.POINTER(CtxInfo)]),
    ("aa_ctx_set_schedule", C.c_int, [_P, C.c_int]), ("aa_ctx_clear_error", C.c_int, [_P]), ("aa_ctx_compute_stream", _P, [_P]), ("aa_ctx_copy_stream", _P, [_P]),
    ("aa_stream_create", C.c_int, [_P, C.c_uint16, C.c_uint16, C.POINTER(_P)]), ("aa_stream_destroy", None, [_P]),
    ("aa_stream_parse", C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(FrameHeader)]),
    ("aa_stream_append_records", C.c_int, [_P, C.POINTER(FrameHeader), _P, _P, C.POINTER(C.c_int)]),
    ("aa_stream_upload", C.c_int, [_P]), ("aa_stream_release_staging", C.c_int, [_P]),
    ("aa_decode_batch", C.c_int, [_P, C.POINTER(_P), C.c_int, C.POINTER(C.c_int)]),
    ("aa_stream_decode", C.c_int, [_P, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("aa_submit_frames", C.c_int, [_P, C.POINTER(FrameIn), C.c_int, C.POINTER(C.c_int), C.c_int]),
    ("aa_submit_frames_ex", C.c_int, [_P, C.POINTER(FrameIn), C.c_int, C.POINTER(C.c_int), C.c_int, C.c_uint]),
    ("aa_launch_tokens", C.c_int, [_P, C.c_int, C.POINTER(C.c_int)]),
    ("aa_ssim_host", C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    ("aa_stream_lf_search", C.c_int, [_P, C.c_char_p, C.c_size_t, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                      C.POINTER(C.c_double), C.c_void_p]),
    ("aa_stream_frame_header", C.c_int, [_P, C.c_int, C.POINTER(FrameHeader)]),
    ("aa_stream_read_records", C.c_int, [_P, C.c_int, _P, _P, C.c_size_t]),
    ("aa_stream_frame_count", C.c_int, [_P]), ("aa_stream_release_before", C.c_int, [_P, C.c_int]),
    ("aa_stream_rewind", C.c_int, [_P]), ("aa_stream_rewind_to", C.c_int, [_P, C.c_int]),
    ("aa_stream_download", C.c_int, [_P, C.c_int, _P, _P, _P]),
    ("aa_pinned_alloc", C.c_int, [_P, C.c_size_t, C.POINTER(_P)]), ("aa_pinned_free", None, [_P]),
    ("aa_stream_download_async", C.c_int, [_P, C.c_int, _P, _P, _P]), ("aa_stream_download_wait", C.c_int, [_P]),
    ("aa_download_batch_async", C.c_int, [_P, C.POINTER(_P), C.c_int, C.POINTER(C.c_int), _P, C.c_size_t]), ("aa_ctx_download_wait", C.c_int, [_P]), ("aa_ctx_download_wait_until", C.c_int, [_P, C.c_int]),
    ("aa_stream_raster_device", C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    ("aa_stream_references", C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("aa_stream_reference_slots", C.c_int, [_P, C.POINTER(C.c_int)]),
    ("aa_stream_import_reference", C.c_int, [_P, _P, _P, _P]),
    ("aa_stream_import_reference_host", C.c_int, [_P, _P, _P, _P]),
    ("aa_parser_state_hash", C.c_int, [_P, C.POINTER(C.c_uint64)]), ("aa_stream_state_hash", C.c_int, [_P, C.POINTER(C.c_uint64)]),
    ("aa_stream_raster_hash", C.c_int, [_P, C.c_int, C.POINTER(C.c_uint64)]),
    ("aa_stream_decoder_hash", C.c_int, [_P, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    ("aa_stream_minihash", C.c_int, [_P, C.POINTER(C.c_uint32)]),
    ("aa_stream_release_frame", C.c_int, [_P, C.c_int]),
    ("aa_stream_set_references", C.c_int, [_P, C.POINTER(_P * 3), C.POINTER(C.c_int)]),
    ("aa_stream_reference_device", C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(_P), C.POINTER(_P)]),
    ("aa_stream_reference_download", C.c_int, [_P, C.c_int, _P, _P, _P]),
    ("aa_raster_geometry", None, [C.c_uint16, C.c_uint16, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    ("aa_ctx_profile", C.c_int, [_P, C.c_int]), ("aa_ctx_kernel_stats", C.c_int, [_P, C.POINTER(KernelStats), C.c_int]),
]

_lib = None


def lib():
    """Load (building if stale) the native library.  Raises if it cannot be built or loaded: no fallback."""
    global _lib
    if _lib is None:
        path = _build.LIB
        if os.environ.get("ALFALFA_AMD_LIB"):              # (A/B runs of a differently built library; never set in production)
            path = os.environ["ALFALFA_AMD_LIB"]
        elif _build.stale():
            if os.path.exists(_build.HIPCC):
                _build.build()
            elif not os.path.exists(path):
                raise RuntimeError("libalfalfa_amd.so is missing and hipcc is not available: there is no CPU fallback")
        L = C.CDLL(path)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.restype, fn.argtypes = restype, argtypes
        L.aa_runtime_prepare()             # GPU_MAX_HW_QUEUES=16 unless the environment says otherwise: before this process's first HIP call, if it is ours
        _lib = L
    return _lib


def check(rc):
    if rc != AA_OK:
        raise AlfalfaError(rc, lib().aa_last_error().decode("utf-8", "replace"))


def device_count():
    return lib().aa_device_count()


def raster_geometry(width, height):
    pw, ph = C.c_uint32(), C.c_uint32()
    lib().aa_raster_geometry(width, height, C.byref(pw), C.byref(ph))
    return pw.value, ph.value
