#!/usr/bin/env python3
"""Deterministic synthetic source video written as YUV4MPEG2 C420jpeg.

luma = 128 + 50 sin(x/a) + 40 cos(y/b) + 20 sin((x+y)/c) + texture, sampled from a
(W+64)x(H+64) canvas translated by (3n mod 64, 2n mod 64) per frame (real sub-pel motion
once the encoder searches it), plus per-pixel uniform noise.  Test/bench tooling only.
"""
import argparse
import numpy as np


def synth_frames(width, height, frames, seed, entropy="low"):
    rng = np.random.default_rng(seed)
    if entropy == "low":
        a, b, c, tex, noise = 37.0, 23.0, 11.0, 12, 3
    else:
        a, b, c, tex, noise = 7.0, 5.0, 11.0, 40, 10
    cw, ch = width + 64, height + 64
    x = np.arange(cw)[None, :]
    y = np.arange(ch)[:, None]
    canvas = 128 + 50 * np.sin(x / a) + 40 * np.cos(y / b) + 20 * np.sin((x + y) / c)
    canvas = canvas + rng.integers(-tex, tex + 1, size=(ch, cw))
    cu = 128 + 30 * np.sin(x[:, ::2] / 29.0) + 20 * np.cos(y[::2, :] / 31.0)
    cv = 128 + 30 * np.cos(x[:, ::2] / 19.0) + 20 * np.sin(y[::2, :] / 41.0)
    for n in range(frames):
        ox, oy = (3 * n) % 64, (2 * n) % 64
        Y = canvas[oy:oy + height, ox:ox + width] + rng.integers(-noise, noise + 1, size=(height, width))
        cwid, chei = (width + 1) // 2, (height + 1) // 2
        U = cu[oy // 2:oy // 2 + chei, ox // 2:ox // 2 + cwid]
        V = cv[oy // 2:oy // 2 + chei, ox // 2:ox // 2 + cwid]
        yield (np.clip(Y, 0, 255).astype(np.uint8), np.clip(U, 0, 255).astype(np.uint8),
               np.clip(V, 0, 255).astype(np.uint8))


def write_y4m(path, width, height, frames, seed, entropy="low", fps=30):
    with open(path, "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F%d:1 Ip A1:1 C420jpeg\n" % (width, height, fps))
        for Y, U, V in synth_frames(width, height, frames, seed, entropy):
            f.write(b"FRAME\n")
            f.write(Y.tobytes()); f.write(U.tobytes()); f.write(V.tobytes())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out"); ap.add_argument("--width", type=int, default=176)
    ap.add_argument("--height", type=int, default=144); ap.add_argument("--frames", type=int, default=6)
    ap.add_argument("--seed", type=int, default=7); ap.add_argument("--entropy", default="low")
    ap.add_argument("--fps", type=int, default=30)
    a = ap.parse_args()
    write_y4m(a.out, a.width, a.height, a.frames, a.seed, a.entropy, a.fps)
