"""Where does a row-pipelined launch spend its time?  Times the three phases (HIP events, aa_ctx_profile) for geometries
that isolate the terms of  T = mbw * step + (mbh - 1) * lag :
    probe_1row  1920x16    -> step (no cross-row waits at all)
    probe_1col  16x1088    -> lag  (pure hand-off chain)
    1080p_inter_lf         -> the real thing
at several stream counts (throughput- vs latency-bound).     python tools/row_kernel_probe.py [streams ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import alfalfa_amd as aa  # noqa: E402
import workload  # noqa: E402


def run(config, S, F=3):
    w, h = workload.CONFIGS[config][:2]
    paths = workload.make_streams(config, F, list(range(S)))
    ctx = aa.Context(0)
    decs = []
    for p in paths:
        d = aa.Decoder(ctx, w, h)
        for fr in aa.read_ivf(p)[2]:
            d.parse_frame(fr)
        d.upload(); decs.append(d)
    for rep in range(2):
        if rep == 1:
            ctx.profile(True); ctx.kernel_stats(reset=True)
        for f in range(F):
            ctx.decode_batch(decs, [f] * S)
        ctx.sync()
        for d in decs:
            d.rewind()
    st = ctx.kernel_stats(reset=True); ctx.profile(False)
    mbw, mbh = (w + 15) // 16, (h + 15) // 16
    lf_us = st["loopfilter_ms"] * 1e3 / max(1, st["loopfilter_launches"])
    print("%-16s S=%4d  mb %3dx%-3d  LF %8.1f us/launch (%d)   intra total %8.1f us (%d launches)   inter %8.1f us/launch" % (
        config, S, mbw, mbh, lf_us, st["loopfilter_launches"], st["recon_intra_ms"] * 1e3, st["recon_intra_launches"],
        st["recon_inter_ms"] * 1e3 / max(1, st["recon_inter_launches"])), flush=True)


if __name__ == "__main__":
    Ss = [int(a) for a in sys.argv[1:]] or [4, 32, 120]
    cfgs = os.environ.get("PROBE_CONFIGS", "probe_1row,probe_4rows,probe_1col,1080p_inter_lf").split(",")
    for cfg in cfgs:
        for S in Ss:
            run(cfg, S)
