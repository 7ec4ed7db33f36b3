#!/bin/bash
# Samples rocm-smi power / sclk while one GEMM shape loops: is the dominant kernel power-(clock-)limited?
# usage (GPU box): tools/power_probe.sh <shape-index> <tile> [extra gemm_bench args]
SHAPE=${1:-3}; TILE=${2:-0}; shift 2
python tools/gemm_bench.py --math f16x3 --only "$SHAPE" --tile "$TILE" --iters 1500 "$@" > /tmp/pp_bench.log 2>&1 &
BP=$!
sleep 4   # import + warm-up
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Average Graphics Package Power|Current Socket Graphics Package Power|sclk clock level|Power \(W\)" | tr '\n' ' ' | sed 's/=\+//g'
  echo
  sleep 0.3
done
wait $BP
tail -1 /tmp/pp_bench.log
