"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import struct
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_DIR = os.path.join(HERE, "_ref")


class MB(C.Structure):
    _fields_ = [("y_mode", C.c_uint8), ("uv_mode", C.c_uint8), ("ref_frame", C.c_uint8),
                ("segment_id", C.c_uint8), ("skip", C.c_uint8), ("has_nonzero", C.c_uint8),
                ("has_y2", C.c_uint8), ("split_partition", C.c_uint8),
                ("b_mode", C.c_uint8 * 16), ("mv", (C.c_int16 * 2) * 16), ("uv_mv", (C.c_int16 * 2) * 4),
                ("coeff", (C.c_int16 * 16) * 25), ("block_nonzero", C.c_uint8 * 25)]


MB_DTYPE = np.dtype([("y_mode", "u1"), ("uv_mode", "u1"), ("ref_frame", "u1"), ("segment_id", "u1"),
                     ("skip", "u1"), ("has_nonzero", "u1"), ("has_y2", "u1"), ("split_partition", "u1"),
                     ("b_mode", "u1", (16,)), ("mv", "<i2", (16, 2)), ("uv_mv", "<i2", (4, 2)),
                     ("coeff", "<i2", (25, 16)), ("block_nonzero", "u1", (25,))], align=True)


class FrameInfo(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("key_frame", "shown", "loop_filter_level", "sharpness", "num_partitions",
                                      "segmentation_enabled", "filter_adjustments_enabled", "q_index",
                                      "refresh_last", "refresh_golden", "refresh_alt", "copy_to_golden", "copy_to_alt")]


def build():
    subprocess.run(["make", "-s", "-C", HERE, "liboracle.so"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.vp8o_create.restype = C.c_void_p; L.vp8o_create.argtypes = [C.c_int, C.c_int]
        L.vp8o_destroy.argtypes = [C.c_void_p]
        L.vp8o_decode_frame.restype = C.c_int
        L.vp8o_decode_frame.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
        L.vp8o_plane.restype = C.POINTER(C.c_uint8)
        L.vp8o_plane.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.vp8o_ref_plane.restype = C.POINTER(C.c_uint8); L.vp8o_ref_plane.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.vp8o_error.restype = C.c_char_p; L.vp8o_error.argtypes = [C.c_void_p]
        L.vp8o_macroblocks.restype = C.POINTER(MB)
        L.vp8o_macroblocks.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.vp8o_get_probs.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
        L.vp8o_set_error_concealment.argtypes = [C.c_void_p, C.c_int]
        L.vp8o_get_frame_info.argtypes = [C.c_void_p, C.POINTER(FrameInfo)]
        L.vp8o_set_phases.argtypes = [C.c_void_p, C.c_int]
        L.oracle_ssim_plane.restype = C.c_double
        L.oracle_ssim_plane.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_void_p]
        L.oracle_ssim_window.restype = C.c_float
        L.oracle_ssim_window.argtypes = [C.c_int] * 4
        assert C.sizeof(MB) == MB_DTYPE.itemsize, (C.sizeof(MB), MB_DTYPE.itemsize)
        _lib = L
    return _lib


class OracleError(Exception):
    def __init__(self, code, msg):
        super().__init__("%d: %s" % (code, msg)); self.code = code


class OracleDecoder:
    """One decoder instance == the reference's `Decoder(width, height)` (decoder.hh:244)."""

    def __init__(self, width, height):
        self.L = lib(); self.h = self.L.vp8o_create(width, height)
        self.width, self.height = width, height

    def __del__(self):
        if getattr(self, "h", None):
            self.L.vp8o_destroy(self.h); self.h = None

    def set_error_concealment(self, on):
        self.L.vp8o_set_error_concealment(self.h, int(on))

    def decode(self, frame_bytes):
        shown = C.c_int(0)
        rc = self.L.vp8o_decode_frame(self.h, frame_bytes, len(frame_bytes), C.byref(shown))
        if rc != 0:
            raise OracleError(rc, self.L.vp8o_error(self.h).decode())
        return bool(shown.value)

    def plane(self, p):
        w, h = C.c_int(), C.c_int()
        ptr = self.L.vp8o_plane(self.h, p, C.byref(w), C.byref(h))
        return np.ctypeslib.as_array(ptr, shape=(h.value, w.value)).copy()

    def planes(self):
        return [self.plane(p) for p in range(3)]

    def raster_bytes(self):
        return b"".join(self.plane(p).tobytes() for p in range(3))

    def macroblocks(self):
        w, h = C.c_int(), C.c_int()
        ptr = self.L.vp8o_macroblocks(self.h, C.byref(w), C.byref(h))
        n = w.value * h.value
        buf = C.string_at(ptr, n * C.sizeof(MB))
        return np.frombuffer(buf, dtype=MB_DTYPE).reshape(h.value, w.value)

    def probs(self):
        out = (C.c_uint8 * 1101)(); self.L.vp8o_get_probs(self.h, out)
        return np.frombuffer(bytes(out), dtype=np.uint8)

    def frame_info(self):
        fi = FrameInfo(); self.L.vp8o_get_frame_info(self.h, C.byref(fi))
        return {n: getattr(fi, n) for n, _ in FrameInfo._fields_}

    def set_phases(self, mask):
        self.L.vp8o_set_phases(self.h, mask)


def ssim_plane(a, b, width, height):
    """x264's pixel_ssim_wxh / count over two planes (bytes, stride = width) as restated in oracle/ssim_x264.c."""
    assert len(a) == len(b) == width * height
    return lib().oracle_ssim_plane(bytes(a), bytes(b), width, height, None)


def read_ivf(path_or_bytes):
    """IVF container (util/ivf.cc:36-82): returns (width, height, [frame bytes])."""
    data = path_or_bytes if isinstance(path_or_bytes, (bytes, bytearray)) else open(path_or_bytes, "rb").read()
    assert data[:4] == b"DKIF", "missing IVF file header"
    hdr_len = struct.unpack_from("<H", data, 6)[0]
    width, height = struct.unpack_from("<HH", data, 12)
    nframes = struct.unpack_from("<I", data, 24)[0]
    frames, pos = [], hdr_len
    for _ in range(nframes):
        n = struct.unpack_from("<I", data, pos)[0]
        frames.append(bytes(data[pos + 12:pos + 12 + n])); pos += 12 + n
    return width, height, frames


def write_ivf(path, width, height, frames, fps=30):
    with open(path, "wb") as f:
        f.write(b"DKIF" + struct.pack("<HH4sHHIIII", 0, 32, b"VP80", width, height, fps, 1, len(frames), 0))
        for i, fr in enumerate(frames):
            f.write(struct.pack("<IQ", len(fr), i)); f.write(fr)


def ref_available():
    return os.path.exists(os.path.join(REF_DIR, "ref_decode"))


def ref_decode(ivf_path, out_path, display=False, conceal=False):
    """Run the REFERENCE decoder (oracle/_ref/ref_decode). Returns list of (key, shown) per frame.
    conceal: Decoder::set_error_concealment( true ) first."""
    cmd = [os.path.join(REF_DIR, "ref_decode")] + (["--display"] if display else []) + (["--conceal"] if conceal else []) + [ivf_path, out_path]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("ref_decode failed: " + r.stderr.strip())
    info = []
    for line in r.stdout.splitlines():
        kv = dict(p.split("=") for p in line.split()[2:])
        info.append((int(kv["key"]), int(kv["shown"])))
    return info
