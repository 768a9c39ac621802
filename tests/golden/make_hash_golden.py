"""Writes tests/golden/hash_golden.json: for every golden stream, per frame, the reference's DecoderState::hash, the three
reference rasters' hashes, DecoderHash::hash and Decoder::minihash, as computed by the reference itself
(oracle/_ref/ref_hash; boost::hash_combine = the pre-1.81 formula, see oracle/ref_shims/boost/functional/hash.hpp).
Run where /root/reference exists:  make -C oracle ref && python tests/golden/make_hash_golden.py"""
import glob
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
out = {}
for ivf in sorted(glob.glob(os.path.join(HERE, "*.ivf"))):
    name = os.path.basename(ivf)[:-4]
    lines = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_hash"), ivf], check=True, capture_output=True, text=True).stdout.splitlines()
    rows = [json.loads(l) for l in lines]
    out[name] = {k: [r[k] for r in rows] for k in ("state", "last", "golden", "alternative", "hash", "minihash")}
json.dump(out, open(os.path.join(HERE, "hash_golden.json"), "w"), indent=0, sort_keys=True)
print("wrote", len(out), "streams")
