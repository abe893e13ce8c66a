cd $GRAFT_REPO_ROOT
for f in gemmul8_amd/lib/lib_*.so; do cp gemmul8_amd/lib/libgemmul8.so /tmp/keep.so 2>/dev/null; cp $f gemmul8_amd/lib/libgemmul8.so; echo $f; python tools/gemm_bench.py --iters 5; done
