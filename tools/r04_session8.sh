#!/bin/bash
# Round 4, eighth GPU session: which fill policy is the default?  20-step runs of the headline config, twice each.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
S="--steps 20 --warmup 5 --secondary= --small-batches= --no-cpu-baseline --lanes-only-steps 0 --no-device-half"
for tag in nourgent1:--no-urgent-host urgent1: nourgent2:--no-urgent-host urgent2: lanesonly:"--no-urgent-host --host-share-ms 0"; do
  name=${tag%%:*}; flags=${tag#*:}
  timeout 150 python bench.py $S $flags > gpurun_out/r04h_$name.log 2> gpurun_out/r04h_$name.err
  echo "$name rc=$? $(grep 'timed region' gpurun_out/r04h_$name.err | cut -c1-260)"
done
