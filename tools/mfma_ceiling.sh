#!/bin/bash
# Power-limited MFMA ceiling of the board: run tools/ubench/mfma_peak for each case while sampling rocm-smi (clock, socket power).
cd $GRAFT_REPO_ROOT
for args in "i8" "i8 zero" "f8" "f8 zero"; do
  ./tools/ubench/mfma_peak 5 $args > /tmp/mp.log 2>&1 &
  PID=$!
  sleep 2.5
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Power (W)" | tr -s ' ' | tr '\n' ' '
  echo
  wait $PID
  cat /tmp/mp.log | grep -v amdgpu
done
