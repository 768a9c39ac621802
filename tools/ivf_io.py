"""Minimal IVF container read/write (util/ivf.cc:36-82, util/ivf_writer.cc) for test/bench tooling."""
import struct


def read_ivf(path_or_bytes):
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
