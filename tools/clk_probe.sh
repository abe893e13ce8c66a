#!/bin/bash
# Sample the shader clock / power while the INT8 GEMM micro-benchmark loops (usage: tools/clk_probe.sh [lib.so ...])
cd $GRAFT_REPO_ROOT
cp gemmul8_amd/lib/libgemmul8.so /tmp/keep.so
for f in "$@"; do
  cp $f gemmul8_amd/lib/libgemmul8.so
  echo "== $f"
  python tools/gemm_bench.py --iters 600 > /tmp/gb.log 2>&1 &
  PID=$!
  sleep 4
  for i in 1 2 3; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power (W)\|Average Graphics\|Socket" | tr -s ' ' | head -4
    sleep 0.5
  done
  wait $PID
  grep gemm_i8 /tmp/gb.log
done
cp /tmp/keep.so gemmul8_amd/lib/libgemmul8.so
