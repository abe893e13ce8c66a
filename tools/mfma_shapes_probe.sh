#!/bin/bash
# Power-limited INT8 MFMA ceilings by instruction shape and operand distribution (tools/ubench/mfma_shapes.hip), with the
# shader clock / socket power sampled mid-run.  usage (GPU box): bash tools/mfma_shapes_probe.sh > gpurun_out/mfma_shapes.txt
cd ${GRAFT_REPO_ROOT:-.}
for shape in ${SHAPES:-32 16 16w f32 f16}; do
  for data in rand res small sparse zero; do
    case $shape in f*) case $data in small|sparse) continue;; esac;; esac
    tools/ubench/mfma_shapes 4 $shape $data > /tmp/ms.log 2>&1 &
    PID=$!
    sleep 2.5
    rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket" | tr -s ' \t' ' ' | tr '\n' ' '
    echo
    wait $PID
    cat /tmp/ms.log
  done
done
