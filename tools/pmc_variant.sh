#!/bin/bash
# PMC pass (tools/pmc_lds_ta.txt: LDS / wait / TA / TCP counters, two per pass) of the INT8 GEMM micro-benchmark for one or more builds of
# libgemmul8.so.   usage: tools/pmc_variant.sh <outtag> lib_a.so [lib_b.so ...]   -> gpurun_out/<outtag>_<lib>.txt
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
TAG=$1; shift
mkdir -p $(dirname $O/$TAG)
cp $R/gemmul8_amd/lib/libgemmul8.so /tmp/keep_product.so
cd /tmp && export TMPDIR=/tmp
for f in "$@"; do
  b=$(basename $f .so)
  cp $R/$f $R/gemmul8_amd/lib/libgemmul8.so
  timeout 600 rocprofv3 -i $R/tools/pmc_lds_ta.txt --kernel-trace --output-format csv -d $O/${TAG}_$b -o p -- python $R/tools/gemm_bench.py --iters 3 --warmup 1 > $O/${TAG}_$b.log 2>&1
  python $R/tools/pmc_summary.py $O/${TAG}_$b gemm_i8 > $O/${TAG}_$b.txt
  find $O/${TAG}_$b -type f \( -name "*.db" -o -name "*_trace.csv" -o -size +4M \) -delete
  echo "== $b"; cat $O/${TAG}_$b.txt
done
cp /tmp/keep_product.so $R/gemmul8_amd/lib/libgemmul8.so
