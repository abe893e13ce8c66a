#!/bin/bash
# PMC pass over the complex low-precision phase (ZGEMM 8192^3, 20 moduli: tools/cplx_ab.py): the X / Y launches are gemm_i8_kernel<0, ..>, the combine
# launch gemm_i8_kernel<2, ..> -- per-kernel counters of both from ONE run.   -> gpurun_out/r04/pmc_cplx.txt
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 -i $R/tools/pmc_cplx.txt --kernel-trace --output-format csv -d $O/pmc_cplx -o p -- python $R/tools/cplx_ab.py $R/gemmul8_amd/lib/libgemmul8.so > $O/pmc_cplx.log 2>&1
python $R/tools/pmc_summary.py $O/pmc_cplx gemm_i8_kernel > $O/pmc_cplx.txt
find $O/pmc_cplx -type f \( -name "*.db" -o -name "*_trace.csv" -o -size +4M \) -delete
cat $O/pmc_cplx.txt; tail -3 $O/pmc_cplx.log
