"""alfalfa_amd: MI355X-native VP8 decode hot path behind excamera/alfalfa's Decoder API.

csrc/  HIP kernels (gfx950), host bitstream parser, C ABI (include/alfalfa_amd.h)
capi   ctypes binding of the C ABI;  decoder  Python mirror of Decoder / FilePlayer for this path
"""
import os as _os

# HIP reads GPU_MAX_HW_QUEUES when its runtime starts: a context runs 14 streams side by side and needs that many hardware
# queues (the library sets the same default when it is loaded, which is too late if another HIP user started the runtime).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .decoder import AlfalfaError, Context, Decoder, FilePlayer, Parser, read_ivf  # noqa: F401
