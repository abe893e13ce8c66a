#!/bin/bash
# rocprofv3 evidence for config 3 (SGEMM 16384^3, 6 moduli, FP8 backend) -- run on the GPU box through gpurun; summaries land in gpurun_out/<tag>_*:
#   1. kernel-trace stats of `bench.py --config 3 --lean`                       -> <tag>_config3_kernel_stats.csv + the bench line
#   2. MFMA-busy / wave-state counters of gemm_f6_kernel                         -> <tag>_pmc_config3_summary.txt
#   3. FETCH_SIZE / WRITE_SIZE (separate passes) of every kernel of a call       -> <tag>_pmc_traffic_config3.json (tools/pmc_traffic.py ... fp8)
# Counter passes run with --kernel-trace only (never combined with sys/hip/hsa tracing on this pool).
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_c3_stats -o s -- python $R/bench.py --config 3 --lean --steps 5 --warmup 2 > $O/${TAG}_config3_bench_under_rocprof.json 2> $O/${TAG}_c3_stats.log
cp $(find $O/${TAG}_c3_stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_config3_kernel_stats.csv
rocprofv3 -i $R/tools/pmc_f6.txt --kernel-trace --output-format csv -d $O/${TAG}_c3_pmc -o p -- python $R/bench.py --config 3 --lean --steps 2 --warmup 1 > $O/${TAG}_c3_pmc.log 2>&1
python $R/tools/pmc_summary.py $O/${TAG}_c3_pmc gemm_f > $O/${TAG}_pmc_config3_summary.txt
rocprofv3 -i $R/tools/pmc_fetch.txt --kernel-trace --output-format csv -d $O/${TAG}_c3_fetch -o p -- python $R/bench.py --config 3 --lean --steps 2 --warmup 1 > $O/${TAG}_c3_fetch.log 2>&1
rocprofv3 -i $R/tools/pmc_write.txt --kernel-trace --output-format csv -d $O/${TAG}_c3_write -o p -- python $R/bench.py --config 3 --lean --steps 2 --warmup 1 > $O/${TAG}_c3_write.log 2>&1
python $R/tools/pmc_traffic.py $O/${TAG}_c3_fetch $O/${TAG}_c3_write $O/${TAG}_pmc_traffic_config3.json 16384 6 fp8 > $O/${TAG}_pmc_traffic_config3.txt
find $O/${TAG}_c3_stats $O/${TAG}_c3_pmc $O/${TAG}_c3_fetch $O/${TAG}_c3_write -type f \( -name "*.db" -o -name "*_trace.csv" -o -size +4M \) -delete
cat $O/${TAG}_pmc_config3_summary.txt; cat $O/${TAG}_pmc_traffic_config3.txt
