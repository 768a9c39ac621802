"""The N>1 path on CPU: world_size 2, gloo.  Streams are sharded disjointly across ranks, and the ExCamera-style
entry-state hand-off (DecoderState blob broadcast + continuation) is exercised at the host-parser level: every rank
continues the shared GOP from the broadcast state and must parse exactly what a straight parse yields.  (The device half
of the hand-off -- raster broadcast with RCCL -- runs in bench.py --gpus N on the GPU box.)"""
import hashlib
import os
import subprocess
import sys

import pytest

from conftest import ROOT

WORKER = r'''
import hashlib, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import torch, torch.distributed as dist
import alfalfa_amd as aa
from alfalfa_amd import sharding
from conftest import golden_frames
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
ids = sharding.stream_ids(rank, world, 3)
all_ids = [None] * world
dist.all_gather_object(all_ids, ids)
flat = [i for part in all_ids for i in part]
assert len(set(flat)) == len(flat) == 3 * world and flat == list(range(100, 100 + 3 * world))
w, h, frames = golden_frames("synth_96x80_s1")
blob = b""
if rank == 0:
    head = aa.Parser(w, h); head.parse(frames[0]); blob = head.export_state()
blob = sharding.broadcast_bytes(dist, blob, 0)
cont = aa.Parser(w, h); cont.import_state(blob)
hh = hashlib.sha256()
for fr in frames[1:]:
    hdr, mb, cf = cont.parse(fr); hh.update(mb.tobytes()); hh.update(cf.tobytes())
straight = aa.Parser(w, h); hs = hashlib.sha256()
for i, fr in enumerate(frames):
    hdr, mb, cf = straight.parse(fr)
    if i: hs.update(mb.tobytes()); hs.update(cf.tobytes())
assert hh.digest() == hs.digest(), "continuation differs from straight parse"
assert sharding.digests_agree(dist, hh.digest())
# the same hand-off in the REFERENCE's wire format (DecoderState::serialize, what a reference-built rank would send)
wire = b""
if rank == 0:
    wire = head.serialize_state()
wire = sharding.broadcast_bytes(dist, wire, 0)
cont2 = aa.Parser(w, h); cont2.deserialize_state(wire)
h2 = hashlib.sha256()
for fr in frames[1:]:
    hdr, mb, cf = cont2.parse(fr); h2.update(mb.tobytes()); h2.update(cf.tobytes())
assert h2.digest() == hs.digest(), "continuation from the reference-format state differs"
dist.barrier(); dist.destroy_process_group()
open(os.path.join(sys.argv[2], "rank%d.ok" % rank), "w").write("ok")
'''


@pytest.mark.timeout(600)
def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", str(script), ROOT, str(tmp_path)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=580)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()


def test_stream_ids():
    from alfalfa_amd import sharding
    assert sharding.stream_ids(0, 8, 4) == [100, 101, 102, 103]
    assert sharding.stream_ids(7, 8, 4) == [128, 129, 130, 131]
    with pytest.raises(ValueError):
        sharding.stream_ids(8, 8, 4)
