"""Python mirror of the reference's decoder surface for THIS path, over the C ABI.

Names follow the reference: Parser ~ DecoderState::parse_and_apply (decoder_state.hh:72-167),
Decoder ~ Decoder (decoder.hh:244-300: decode_frame / get_frame_output / parse_and_decode_frame /
get_references), FilePlayer ~ FilePlayer (player.hh:66-97: advance / eof), DecodeBatch = N
independent streams decoded in lockstep (ExCamera chunks / GOPs, one batch per GPU).
"""
import ctypes as C
import struct

import numpy as np

from . import capi
from .capi import AlfalfaError, FrameHeader, MB_INFO_DTYPE  # noqa: F401


class Parser:
    """Host-only bitstream parser with the reference's persistent DecoderState."""

    def __init__(self, width, height):
        self.L = capi.lib()
        self.h = C.c_void_p()
        capi.check(self.L.aa_parser_create(width, height, C.byref(self.h)))
        self.width, self.height = width, height
        self.mbw, self.mbh = (width + 15) // 16, (height + 15) // 16
        n = self.mbw * self.mbh
        self._mb = np.zeros(n, dtype=MB_INFO_DTYPE)
        self._coeff = np.zeros(n * 25 * 16, dtype=np.int16)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.aa_parser_destroy(self.h); self.h = None

    def parse(self, frame_bytes):
        """-> (header dict, mb_info structured array [mbh, mbw], coefficient blocks [n, 16])."""
        hdr = FrameHeader()
        capi.check(self.L.aa_parser_parse(self.h, frame_bytes, len(frame_bytes), C.byref(hdr),
                                          self._mb.ctypes.data_as(C.c_void_p), self._coeff.ctypes.data_as(C.c_void_p)))
        h = hdr.as_dict()
        return h, self._mb.copy().reshape(self.mbh, self.mbw), self._coeff[:h["num_coeff_blocks"] * 16].copy().reshape(-1, 16)

    def set_error_concealment(self, on):
        """Decoder::set_error_concealment (decoder.hh:298): accept frames that end early."""
        capi.check(self.L.aa_parser_set_error_concealment(self.h, int(on)))

    def export_state(self):
        """DecoderState as bytes (host half of the entry-state hand-off, decoder.cc:43-46)."""
        n = self.L.aa_parser_state_size(self.h)
        buf = (C.c_uint8 * n)()
        capi.check(self.L.aa_parser_export_state(self.h, buf, n))
        return bytes(buf)

    def import_state(self, blob):
        capi.check(self.L.aa_parser_import_state(self.h, blob, len(blob)))

    def serialize_state(self):
        """DecoderState in the reference's wire format (DecoderState::serialize, decoder.cc:283-313)."""
        n = C.c_size_t(0)
        capi.check(self.L.aa_parser_serialize_state(self.h, None, 0, C.byref(n)))
        buf = (C.c_uint8 * n.value)()
        capi.check(self.L.aa_parser_serialize_state(self.h, buf, n.value, C.byref(n)))
        return bytes(buf)

    def deserialize_state(self, blob):
        capi.check(self.L.aa_parser_deserialize_state(self.h, blob, len(blob)))

    def state_hash(self):
        """DecoderState::hash (decoder.cc:266-281)."""
        h = C.c_uint64()
        capi.check(self.L.aa_parser_state_hash(self.h, C.byref(h)))
        return h.value

    def probs(self):
        out = (C.c_uint8 * 1101)()
        capi.check(self.L.aa_parser_get_probs(self.h, out))
        return np.frombuffer(bytes(out), dtype=np.uint8)

    def segmentation(self):
        en, ab = C.c_int(), C.c_int()
        q, lf = (C.c_int8 * 4)(), (C.c_int8 * 4)()
        m = np.zeros(self.mbw * self.mbh, dtype=np.uint8)
        capi.check(self.L.aa_parser_get_segmentation(self.h, C.byref(en), C.byref(ab), q, lf, m.ctypes.data_as(C.POINTER(C.c_uint8))))
        return {"enabled": bool(en.value), "absolute": bool(ab.value), "quant": list(q), "lf": list(lf), "map": m.reshape(self.mbh, self.mbw)}

    def filter_adjustments(self):
        en = C.c_int(); r, m = (C.c_int8 * 4)(), (C.c_int8 * 4)()
        capi.check(self.L.aa_parser_get_filter_adjustments(self.h, C.byref(en), r, m))
        return {"enabled": bool(en.value), "ref": list(r), "mode": list(m)}


class Context:
    """One HIP device: compute + copy streams.  One per process in multi-GPU runs."""

    def __init__(self, device=0):
        self.L = capi.lib()
        self.h = C.c_void_p()
        capi.check(self.L.aa_ctx_create(device, C.byref(self.h)))
        self.device = device

    def __del__(self):
        if getattr(self, "h", None):
            self.L.aa_ctx_destroy(self.h); self.h = None

    def sync(self):
        capi.check(self.L.aa_ctx_sync(self.h))

    def memory(self):
        """(free, total) bytes of this device's HBM right now."""
        f, t = C.c_size_t(), C.c_size_t()
        capi.check(self.L.aa_ctx_memory(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def set_memory_limit(self, nbytes):
        """HBM the context may take for its pools (default: 7/8 of what was free at creation)."""
        capi.check(self.L.aa_ctx_set_memory_limit(self.h, int(nbytes)))

    def set_host_share_ms(self, ms):
        """Key frames of big submit calls are parsed by host workers while that is expected to take no longer than `ms` (0: never)."""
        capi.check(self.L.aa_ctx_set_host_share_ms(self.h, float(ms)))

    def set_packed_coefficients(self, on=True):
        """Device-parsed frames store packed coefficients (mask word + non-zero values per block; expanded on the device when
        a frame is reconstructed) instead of dense blocks.  Before the context's first submit_frames only."""
        capi.check(self.L.aa_ctx_set_packed_coefficients(self.h, int(bool(on))))

    def set_lane_per_partition(self, on=True):
        """Frames with several DCT partitions may be entropy-decoded by one token lane per partition (lanes of one wave).  Before
        the context's first submit_frames only.  Simulated on the host; not yet run on a GPU."""
        capi.check(self.L.aa_ctx_set_lane_per_partition(self.h, int(bool(on))))

    def info(self):
        """What the context holds right now (aa_ctx_info): memory by kind, token-worker shape and occupancy."""
        st = capi.CtxInfo()
        capi.check(self.L.aa_ctx_get_info(self.h, C.byref(st)))
        d = {n: getattr(st, n) for n, _ in capi.CtxInfo._fields_}
        d["token_profile"] = list(st.token_profile)
        return d

    def pinned_alloc(self, nbytes):
        """Pinned host memory for asynchronous downloads (aa_pinned_alloc) -> address; free with pinned_free."""
        p = C.c_void_p()
        capi.check(self.L.aa_pinned_alloc(self.h, nbytes, C.byref(p)))
        return p.value

    def pinned_free(self, address):
        self.L.aa_pinned_free(C.c_void_p(address))

    def set_schedule(self, name):
        """"rows" (default): row-pipelined persistent kernels; "diagonal": one launch per anti-diagonal."""
        capi.check(self.L.aa_ctx_set_schedule(self.h, {"rows": 0, "diagonal": 1}[name]))

    def profile(self, enable):
        capi.check(self.L.aa_ctx_profile(self.h, int(enable)))

    def kernel_stats(self, reset=False):
        st = capi.KernelStats()
        capi.check(self.L.aa_ctx_kernel_stats(self.h, C.byref(st), int(reset)))
        return {n: getattr(st, n) for n, _ in capi.KernelStats._fields_}

    def compute_stream(self):
        return self.L.aa_ctx_compute_stream(self.h)

    ROUTES = {"auto": 0, "device": 2, "host": 4}

    def submit_frames(self, pairs, threads=0, defer_tokens=False, route="auto"):
        """Device-side entropy decode (aa_submit_frames): pairs = [(decoder, frame bytes), ...], frames of one decoder in
        stream order.  Host: frame-header pre-pass only; the macroblock headers and tokens are parsed on the GPU.
        -> frame index of every pair in its stream."""
        return self.submit_prepared(self.prepare_frames(pairs), threads, defer_tokens, route)

    def prepare_frames(self, pairs):
        """The ctypes argument block of submit_frames, reusable across calls with the same (decoder, bytes) pairs."""
        n = len(pairs)
        arr = (capi.FrameIn * n)()
        for i, (d, fr) in enumerate(pairs):
            arr[i].stream, arr[i].data, arr[i].size = d.h.value, fr, len(fr)
        return arr, (C.c_int * n)(), [fr for _, fr in pairs]      # (keeps the byte strings alive)

    def submit_prepared(self, prepared, threads=0, defer_tokens=False, route="auto"):
        """defer_tokens: two-phase form (AA_SUBMIT_DEFER_TOKENS) -- macroblock headers now, tokens at launch_tokens().
        route: "auto" (few streams -> host workers, many -> GPU lanes), "device", "host"."""
        arr, out, _keep = prepared
        capi.check(self.L.aa_submit_frames_ex(self.h, arr, len(arr), out, threads, (1 if defer_tokens else 0) | self.ROUTES[route]))
        return list(out)

    def launch_tokens(self, max_batches=0):
        """Second phase of the oldest `max_batches` deferred batches (0: all) -> how many were launched."""
        n = C.c_int()
        capi.check(self.L.aa_launch_tokens(self.h, max_batches, C.byref(n)))
        return n.value

    def download_batch_async(self, decoders, frame_indices, dst_ptr, stride):
        """aa_download_batch_async: frame frame_indices[i] of decoders[i] -> dst_ptr + i * stride (pinned host memory), one gather
        kernel + one copy for the lot; valid after download_wait() / sync()."""
        n = len(decoders)
        arr = (C.c_void_p * n)(*[d.h for d in decoders])
        idx = (C.c_int * n)(*frame_indices)
        capi.check(self.L.aa_download_batch_async(self.h, arr, n, idx, C.c_void_p(dst_ptr), stride))

    def download_wait(self, max_in_flight=None):
        """aa_ctx_download_wait: every batched download queued so far has arrived; with max_in_flight: only until at most that many are
        still on their way (aa_ctx_download_wait_until: a ring of r destination buffers waits with r - 1 before it reuses one)."""
        if max_in_flight is None:
            capi.check(self.L.aa_ctx_download_wait(self.h))
        else:
            capi.check(self.L.aa_ctx_download_wait_until(self.h, int(max_in_flight)))

    def decode_batch(self, decoders, frame_indices):
        n = len(decoders)
        arr = (C.c_void_p * n)(*[d.h for d in decoders])
        idx = (C.c_int * n)(*frame_indices)
        capi.check(self.L.aa_decode_batch(self.h, arr, n, idx))


class Decoder:
    """Decoder(width, height) of the reference, rasters resident in HBM."""

    def __init__(self, ctx, width, height):
        self.ctx, self.L = ctx, capi.lib()
        self.h = C.c_void_p()
        capi.check(self.L.aa_stream_create(ctx.h, width, height, C.byref(self.h)))
        self.width, self.height = width, height
        self.padded_width, self.padded_height = capi.raster_geometry(width, height)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.aa_stream_destroy(self.h); self.h = None

    def set_error_concealment(self, on):
        """Decoder::set_error_concealment (decoder.hh:298): accept frames that end early (host and GPU parser alike)."""
        capi.check(self.L.aa_stream_set_error_concealment(self.h, int(on)))

    def error_concealment(self):
        return bool(self.L.aa_stream_error_concealment(self.h))

    # -- two-step form: parse_frame + decode_frame (decoder.cc:89-118) --
    def parse_frame(self, frame_bytes):
        fi, hdr = C.c_int(), FrameHeader()
        capi.check(self.L.aa_stream_parse(self.h, frame_bytes, len(frame_bytes), C.byref(fi), C.byref(hdr)))
        return fi.value, hdr.as_dict()

    def append_records(self, header, mb, coeff_blocks):
        """A frame given as records (aa_stream_append_records): header dict as frame_header() / Parser.parse() return it,
        macroblock records, coefficient blocks [n, 16] -> frame index.  What an encoder has in hand when it updates its
        references (Encoder::write_frame, encoder.cc:146-160)."""
        hdr = FrameHeader()
        for n, _ in FrameHeader._fields_:
            if n == "quant":
                for sgm in range(4):
                    for k in range(6):
                        hdr.quant[sgm][k] = header["quant"][sgm][k]
            else:
                setattr(hdr, n, header[n])
        mbs = np.ascontiguousarray(mb, dtype=MB_INFO_DTYPE).reshape(-1)
        cf = np.ascontiguousarray(coeff_blocks, dtype=np.int16).reshape(-1, 16)
        hdr.num_coeff_blocks = len(cf)
        fi = C.c_int()
        capi.check(self.L.aa_stream_append_records(self.h, C.byref(hdr), mbs.ctypes.data_as(C.c_void_p), cf.ctypes.data_as(C.c_void_p), C.byref(fi)))
        return fi.value

    def upload(self):
        capi.check(self.L.aa_stream_upload(self.h))

    def release_staging(self):
        """Free the pinned host staging of everything parsed so far (uploads it first)."""
        capi.check(self.L.aa_stream_release_staging(self.h))

    def decode_frame(self, frame_index):
        self.ctx.decode_batch([self], [frame_index])

    # -- one-step form: get_frame_output (decoder.cc:125-135) -> (shown, frame index of the raster) --
    def get_frame_output(self, frame_bytes):
        fi, shown = C.c_int(), C.c_int()
        capi.check(self.L.aa_stream_decode(self.h, frame_bytes, len(frame_bytes), C.byref(fi), C.byref(shown)))
        return bool(shown.value), fi.value

    def parse_and_decode_frame(self, frame_bytes):
        shown, fi = self.get_frame_output(frame_bytes)
        return fi if shown else None

    def frame_count(self):
        return self.L.aa_stream_frame_count(self.h)

    def frame_header(self, frame_index):
        hdr = FrameHeader()
        capi.check(self.L.aa_stream_frame_header(self.h, frame_index, C.byref(hdr)))
        return hdr.as_dict()

    def read_records(self, frame_index):
        """A frame's parsed records as they sit in HBM -> (header dict, mb_info [mbh, mbw], coefficient blocks [n, 16])."""
        h = self.frame_header(frame_index)
        mbw, mbh = h["mb_width"], h["mb_height"]
        mb = np.zeros(mbw * mbh, dtype=MB_INFO_DTYPE)
        cf = np.zeros((max(1, h["num_coeff_blocks"]), 16), dtype=np.int16)
        capi.check(self.L.aa_stream_read_records(self.h, frame_index, mb.ctypes.data_as(C.c_void_p), cf.ctypes.data_as(C.c_void_p), len(cf)))
        return h, mb.reshape(mbh, mbw), cf[:h["num_coeff_blocks"]]

    def rewind(self):
        capi.check(self.L.aa_stream_rewind(self.h))

    def decoder_hash(self):
        """DecoderHash (decoder.cc:143-153): ([state, last, golden, alternative], hash of the four)."""
        parts, whole = (C.c_uint64 * 4)(), C.c_uint64()
        capi.check(self.L.aa_stream_decoder_hash(self.h, parts, C.byref(whole)))
        return list(parts), whole.value

    def minihash(self):
        h = C.c_uint32()
        capi.check(self.L.aa_stream_minihash(self.h, C.byref(h)))
        return h.value

    def raster_hash(self, frame_index):
        h = C.c_uint64()
        capi.check(self.L.aa_stream_raster_hash(self.h, frame_index, C.byref(h)))
        return h.value

    def release_frame(self, frame_index):
        capi.check(self.L.aa_stream_release_frame(self.h, frame_index))

    def rewind_to(self, frame_index):
        capi.check(self.L.aa_stream_rewind_to(self.h, frame_index))

    def lf_search(self, frame_bytes, original_luma, level_lo, level_hi, want_rasters=False):
        """Encoder::apply_best_loopfilter_settings (encoder.cc:459-516) as one batch: -> (best level, its SSIM, [SSIM of every
        candidate], [raster bytes of every candidate] or None).  original_luma: padded_height x padded_width uint8."""
        n = level_hi - level_lo + 1
        orig = np.ascontiguousarray(original_luma, dtype=np.uint8)
        assert orig.shape == (self.padded_height, self.padded_width), orig.shape
        best, q = C.c_int(), C.c_double()
        qs = (C.c_double * n)()
        rb = sum(self.plane_sizes())
        rasters = np.empty(n * rb, np.uint8) if want_rasters else None
        capi.check(self.L.aa_stream_lf_search(self.h, frame_bytes, len(frame_bytes), orig.ctypes.data_as(C.c_void_p), level_lo, level_hi,
                                              C.byref(best), C.byref(q), qs, rasters.ctypes.data_as(C.c_void_p) if want_rasters else None))
        return best.value, q.value, list(qs), ([rasters[i * rb:(i + 1) * rb].tobytes() for i in range(n)] if want_rasters else None)

    def release_before(self, first_kept):
        capi.check(self.L.aa_stream_release_before(self.h, first_kept))

    def raster(self, frame_index):
        """VP8Raster of a decoded frame as three padded numpy planes (lazy D2H, like RasterHandle::get())."""
        pw, ph = self.padded_width, self.padded_height
        y = np.empty((ph, pw), np.uint8); u = np.empty((ph // 2, pw // 2), np.uint8); v = np.empty((ph // 2, pw // 2), np.uint8)
        capi.check(self.L.aa_stream_download(self.h, frame_index, y.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))
        return y, u, v

    def download_async(self, frame_index, y_ptr, u_ptr, v_ptr):
        """aa_stream_download_async: the frame's planes into PINNED host memory (Context.pinned_alloc), queued on the copy stream
        behind what the compute stream holds now; valid after Context.download_wait()."""
        capi.check(self.L.aa_stream_download_async(self.h, frame_index, C.c_void_p(y_ptr), C.c_void_p(u_ptr), C.c_void_p(v_ptr)))

    def download_wait(self):
        capi.check(self.L.aa_stream_download_wait(self.h))

    def raster_bytes(self, frame_index):
        return b"".join(p.tobytes() for p in self.raster(frame_index))

    def display_bytes(self, frame_index):
        """BaseRaster::dump (raster.cc:85-114): display rectangle as planar I420."""
        y, u, v = self.raster(frame_index)
        w, h = self.width, self.height
        return y[:h, :w].tobytes() + u[:(h + 1) // 2, :(w + 1) // 2].tobytes() + v[:(h + 1) // 2, :(w + 1) // 2].tobytes()

    def raster_device_pointers(self, frame_index):
        y, u, v = C.c_void_p(), C.c_void_p(), C.c_void_p()
        capi.check(self.L.aa_stream_raster_device(self.h, frame_index, C.byref(y), C.byref(u), C.byref(v)))
        return y.value, u.value, v.value

    def get_references(self):
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        capi.check(self.L.aa_stream_references(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return {"last": a.value, "golden": b.value, "alternative": c.value}

    def export_state(self):
        n = self.L.aa_stream_state_size(self.h)
        buf = (C.c_uint8 * n)()
        capi.check(self.L.aa_stream_export_state(self.h, buf, n))
        return bytes(buf)

    def import_state(self, blob):
        capi.check(self.L.aa_stream_import_state(self.h, blob, len(blob)))

    def serialize(self):
        """The decoder as the reference writes it to a .state file (Decoder::serialize, decoder.cc:54-69)."""
        n = C.c_size_t(0)
        capi.check(self.L.aa_stream_serialize(self.h, None, 0, C.byref(n)))
        buf = (C.c_uint8 * n.value)()
        capi.check(self.L.aa_stream_serialize(self.h, buf, n.value, C.byref(n)))
        return bytes(buf)

    def deserialize(self, blob):
        """Load a reference-format decoder state (EncoderStateDeserializer::build<Decoder>, decoder.cc:48-52,71-81)."""
        capi.check(self.L.aa_stream_deserialize(self.h, blob, len(blob)))

    def plane_sizes(self):
        pw, ph = self.padded_width, self.padded_height
        return pw * ph, (pw // 2) * (ph // 2), (pw // 2) * (ph // 2)

    def export_raster_device(self, frame_index, y_ptr, u_ptr, v_ptr):
        """D2D copy of a decoded raster into caller-owned device planes (async on the compute stream)."""
        capi.check(self.L.aa_stream_export_raster(self.h, frame_index, C.c_void_p(y_ptr), C.c_void_p(u_ptr), C.c_void_p(v_ptr)))

    def import_reference_device(self, y_ptr, u_ptr, v_ptr):
        capi.check(self.L.aa_stream_import_reference(self.h, C.c_void_p(y_ptr), C.c_void_p(u_ptr), C.c_void_p(v_ptr)))

    def import_reference_host(self, y, u, v):
        y, u, v = (np.ascontiguousarray(p, dtype=np.uint8) for p in (y, u, v))
        capi.check(self.L.aa_stream_import_reference_host(self.h, y.ctypes.data_as(C.c_void_p), u.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))


def read_ivf(path_or_bytes):
    """IVF container (util/ivf.cc:36-82): -> (width, height, [frame bytes])."""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    if data[:4] != b"DKIF":
        raise AlfalfaError(-1, "invalid bitstream: missing IVF file header")
    if struct.unpack_from("<H", data, 4)[0] != 0:
        raise AlfalfaError(-2, "unsupported bitstream: not an IVF version 0 file")
    hdr_len = struct.unpack_from("<H", data, 6)[0]
    if hdr_len != 32:
        raise AlfalfaError(-2, "unsupported bitstream: unsupported IVF header length")
    width, height = struct.unpack_from("<HH", data, 12)
    nframes = struct.unpack_from("<I", data, 24)[0]
    frames, pos = [], hdr_len
    for _ in range(nframes):
        if pos + 12 > len(data):
            raise AlfalfaError(-1, "invalid bitstream: IVF file truncated")
        n = struct.unpack_from("<I", data, pos)[0]
        if pos + 12 + n > len(data):
            raise AlfalfaError(-1, "invalid bitstream: IVF file truncated")
        frames.append(bytes(data[pos + 12:pos + 12 + n])); pos += 12 + n
    return width, height, frames


class FilePlayer:
    """FilePlayer (player.cc:85-144): starts at the first key frame; advance() returns the next SHOWN raster."""

    def __init__(self, ctx, path_or_bytes):
        self.width, self.height, self.frames = read_ivf(path_or_bytes)
        self.decoder = Decoder(ctx, self.width, self.height)
        self.frame_no = 0
        while self.frame_no < len(self.frames) and (self.frames[self.frame_no][0] & 1):
            self.frame_no += 1

    def eof(self):
        return self.frame_no == len(self.frames)

    def advance(self):
        while not self.eof():
            fi = self.decoder.parse_and_decode_frame(self.frames[self.frame_no]); self.frame_no += 1
            if fi is not None:
                return fi
        raise AlfalfaError(-2, "unsupported bitstream: hidden frames at end of file")
