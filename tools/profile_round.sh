#!/bin/bash
# rocprofv3 evidence for one round (run on the GPU box through gpurun; summaries land in gpurun_out/<tag>_*, copy them to profiles/):
#   1. kernel-trace stats of the default bench.py run (config 2)            -> <tag>_config2_kernel_stats.csv + the bench line
#   2. MFMA-busy / shader-cycle / LDS / L2 counters of the INT8 GEMM kernel  -> <tag>_pmc_mfma_summary.txt
#   3. FETCH_SIZE and WRITE_SIZE (separate passes) of every kernel of a step -> <tag>_pmc_fetch / <tag>_pmc_write (tools/pmc_traffic.py)
# Counter passes run with --kernel-trace only (never combined with sys/hip/hsa tracing on this pool).
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o s -- python $R/bench.py --no-cpu > $O/${TAG}_config2_bench_under_rocprof.json 2> $O/${TAG}_stats.log
cp $(find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1) $O/${TAG}_config2_kernel_stats.csv
python - <<PY > $O/${TAG}_config2_kernel_stats_note.txt
import csv, glob
f = glob.glob("$O/${TAG}_stats/**/*kernel_trace.csv", recursive=True)[0]
t = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if "gemm_i8_kernel<0" in r["Kernel_Name"]]
print("rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu")
print("oz2::gemm_i8_kernel<0, ...>: %d launches; the first 16 in order (ms):" % len(t), ", ".join(f"{x:.3f}" for x in t[:16]))
print("launches 1-3 = warm-up, 4-13 = the 10 timed steps: mean %.3f ms; the rest = the settled-figure protocol (300 untimed + 30 event-timed calls: "
      "mean of the last 30 of them %.3f ms) and the fast-mode calls of the other_mode line" % (sum(t[3:13]) / 10, sum(t[313:343]) / 30 if len(t) >= 343 else float("nan")))
PY
rocprofv3 -i $R/tools/pmc_mfma.txt --kernel-trace --output-format csv -d $O/${TAG}_pmc_mfma -o p -- python $R/tools/gemm_bench.py --iters 4 --warmup 1 > $O/${TAG}_pmc_mfma.log 2>&1
python $R/tools/pmc_summary.py $O/${TAG}_pmc_mfma gemm_i8 > $O/${TAG}_pmc_mfma_summary.txt
rocprofv3 -i $R/tools/pmc_fetch.txt --kernel-trace --output-format csv -d $O/${TAG}_pmc_fetch -o p -- python $R/bench.py --no-cpu --steps 3 --warmup 1 > $O/${TAG}_pmc_fetch.log 2>&1
rocprofv3 -i $R/tools/pmc_write.txt --kernel-trace --output-format csv -d $O/${TAG}_pmc_write -o p -- python $R/bench.py --no-cpu --steps 3 --warmup 1 > $O/${TAG}_pmc_write.log 2>&1
python $R/tools/pmc_traffic.py $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write $O/${TAG}_pmc_traffic.json > $O/${TAG}_pmc_traffic.txt
# keep the merge small: drop raw traces, keep the counter CSVs
find $O/${TAG}_stats $O/${TAG}_pmc_mfma $O/${TAG}_pmc_fetch $O/${TAG}_pmc_write -type f \( -name "*.db" -o -name "*_trace.csv" -o -size +4M \) -delete
tail -3 $O/${TAG}_config2_kernel_stats_note.txt; cat $O/${TAG}_pmc_mfma_summary.txt; cat $O/${TAG}_pmc_traffic.txt
