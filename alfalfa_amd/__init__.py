"""alfalfa_amd: MI355X-native VP8 decode hot path behind excamera/alfalfa's Decoder API.

csrc/  HIP kernels (gfx950), host bitstream parser, C ABI (include/alfalfa_amd.h)
capi   ctypes binding of the C ABI;  decoder  Python mirror of Decoder / FilePlayer for this path
"""
from .decoder import AlfalfaError, Context, Decoder, FilePlayer, Parser, read_ivf  # noqa: F401
