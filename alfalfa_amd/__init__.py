"""alfalfa_amd: MI355X-native VP8 decode hot path behind excamera/alfalfa's Decoder API.

csrc/  HIP kernels (gfx950), host bitstream parser, C ABI (include/alfalfa_amd.h)
capi   ctypes binding of the C ABI;  decoder  Python mirror of Decoder / FilePlayer for this path
"""
import os as _os

# HIP reads GPU_MAX_HW_QUEUES when its runtime starts: a context runs 16 streams side by side and needs that many hardware
# queues.  The C library itself never touches the process's environment on its own (a host program calls aa_runtime_prepare()
# for that); THIS package is the host program's Python side, and importing it is that call made early: before torch or any other
# HIP user of the process starts the runtime.  A value already in the environment is left alone.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

from .decoder import AlfalfaError, Context, Decoder, FilePlayer, Parser, read_ivf  # noqa: F401
