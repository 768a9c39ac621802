#!/bin/bash
cd $GRAFT_REPO_ROOT
export ALFALFA_AMD_PARSE_TIMEOUT_S=120 ALFALFA_AMD_TOKEN_PROFILE=1
mkdir -p gpurun_out
( echo "== 480 streams x 12, default shape"; timeout 300 python tools/parse_probe.py --streams 480 --reps 2
  echo "== 2200 streams x 12, default shape"; timeout 300 python tools/parse_probe.py --streams 2200 --reps 1
  echo "== 2200 streams x 12, 6 waves per CU"; ALFALFA_AMD_WGS_PER_CU=6 timeout 300 python tools/parse_probe.py --streams 2200 --reps 1
  echo "== 2200 streams x 12, 4 waves per CU, 16 lanes"; ALFALFA_AMD_MAX_LANES=16 timeout 300 python tools/parse_probe.py --streams 2200 --reps 1
  echo "== 1 stream x 12"; timeout 300 python tools/parse_probe.py --streams 1 --reps 2
) > gpurun_out/r03g_probe.log 2>&1
