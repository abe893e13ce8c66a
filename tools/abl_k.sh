cd $GRAFT_REPO_ROOT
cp gemmul8_amd/lib/libgemmul8.so /tmp/keep.so
for f in /tmp/keep.so gemmul8_amd/lib/lib_*.so; do cp $f gemmul8_amd/lib/libgemmul8.so; echo $f; for k in 256 512 8192; do python tools/gemm_bench.py --iters 7 --k $k 2>&1 | grep gemm_i8; done; done
cp /tmp/keep.so gemmul8_amd/lib/libgemmul8.so
