"""Randomised soak of the wave simulation (tests/cpp/wave_sim.cc): random geometries, partition counts, lane counts, pool sizes,
arrival patterns, both coefficient formats, with and without a lane per partition -- every frame against the host parser.
    python tools/wave_sim_soak.py [runs] [seed]
"""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import test_wave_sim as t  # noqa: E402
import vp8_synth  # noqa: E402


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    rng = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    for r in range(runs):
        w, h = rng.choice([(16, 16), (48, 32), (176, 144), (200, 48), (320, 176), (64, 400), (352, 288)])
        n_streams = rng.randint(1, 6)
        streams = []
        for s in range(n_streams):
            if rng.random() < 0.5:
                streams.append(vp8_synth.feature_stream(w, h, rng.randint(0, 10 ** 6), rng.randint(1, 5)).frames)
            else:
                streams.append(t.partitioned_stream(w, h, rng.randint(0, 10 ** 6), rng.randint(0, 3), frames=rng.randint(1, 4), density=rng.choice([0.05, 0.4, 0.9])))
        lanes = rng.choice([1, 2, 3, 8, 16, 22, 33, 64])
        packed, mp = rng.random() < 0.5, rng.random() < 0.6
        plenty = t.run_wave(w, h, streams, lanes, packed=packed, seed=r, burst=rng.choice([1, 2, 1000]), burst_gap=rng.choice([1, 7, 60]), mp=mp, mp_hint=rng.choice([1, 2, 4, 8]))
        line = "%3d %dx%d streams %d lanes %d %s%s: %s" % (r, w, h, n_streams, lanes, "packed " if packed else "", "mp " if mp else "", plenty)
        if rng.random() < 0.4 and plenty["peak_chunks_out"] > 2:
            # a scarce pool: enough for the biggest frame (with a lane per partition every lane of it holds a chunk), far less than the wave wants
            pool = max(10 if mp else 3, plenty["peak_chunks_out"] // 3)
            scarce = t.run_wave(w, h, streams, lanes, pool_chunks=pool, packed=packed, seed=r, mp=mp)
            line += " | pool %d: handed back %d" % (pool, scarce["handed_back"])
        print(line, flush=True)


if __name__ == "__main__":
    main()
