"""bench.py end to end on a small workload, including the one-process form of the multi-GPU path (AA_BENCH_FORCE_DIST: RCCL
process group of size 1, entry-state hand-off over torch.distributed broadcast): the JSON contract, bit-exactness against
the reference, the hand-off."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dist", [False, True])
def test_bench_small(dist):
    env = dict(os.environ)
    env.pop("AA_BENCH_FORCE_DIST", None)
    if dist:
        env.update(AA_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "cif_inter_lf", "--streams", "12", "--frames", "6", "--steps", "3",
                        "--warmup", "1", "--key-ahead", "3", "--depth", "2", "--small-batches", "1,4"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] > 0 and line["n_gpus"] == 1 and line["steps"] == 3 and line["scaling"] == "weak" and line["dtype"] == "u8"
    assert line["verified_bit_exact_vs_reference"]["bit_exact"] is True
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert line["cpu_baseline"]["kind"] in ("reference", "port") and line["cpu_baseline"]["cores"] == 1
    assert "1" in line["small_batches"] and "4" in line["small_batches"]
    if dist:
        assert line["entry_state_handoff"]["continuations_agree"] is True
    else:
        assert line["entry_state_handoff"] is None
