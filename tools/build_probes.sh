#!/bin/bash
# Build LABORATORY variants of libgemmul8.so (gemmul8_amd/lib/lib_<tag>.so; never shipped): only one source (default oz2_gemm_i8.hip) is
# recompiled with the given -D flags -- WITHOUT -DOZ2_PRODUCT_BUILD and with the probe hooks of tools/experiments/probes/lab_hooks.hpp
# available (-DOZ2_PROBE=<bits>, -DOZ2_KSTAG=<n>) -- the other objects come from the regular build.
# usage: tools/build_probes.sh tag1="-DOZ2_PROBE=4" tag2="-DOZ2_SLEEP_A=3" ...     (SRC=oz2_scale etc. selects another source)
set -e
cd "$(dirname "$0")/../gemmul8_amd/csrc"
make -j8 >/dev/null
FLAGS="-std=c++20 -O3 -fPIC --offload-arch=gfx950 -ffp-contract=off -DOCML_BASIC_ROUNDED_OPERATIONS -Wno-unused-function -fvisibility=hidden"
for spec in "$@"; do
  tag="${spec%%=*}"; defs="${spec#*=}"
  src=${SRC:-oz2_gemm_i8}
  if [ "$src" = "oz2_gemm_i8" ] || [ "$src" = "oz2_gemm_f8" ] || [ "$src" = "oz2_gemm_f6" ]; then
    /opt/rocm/bin/hipcc $FLAGS '-DOZ2_LAB_HOOKS="../../tools/experiments/probes/lab_hooks.hpp"' $defs -c $src.hip -o build/${src}_$tag.o
  else
    /opt/rocm/bin/hipcc $FLAGS $defs -c $src.hip -o build/${src}_$tag.o
  fi
  objs=""
  for o in oz2_gemm_i8 oz2_gemm_i8_small oz2_gemm_f8 oz2_gemm_f6 oz2_scale oz2_crt oz2_driver oz2_api oz2_hook oz2_dist; do
    [ "$o" = "$src" ] && objs="$objs build/${src}_$tag.o" || objs="$objs build/$o.o"
  done
  /opt/rocm/lib/llvm/bin/clang++ -shared -fPIC -o ../lib/lib_$tag.so $objs -ldl -lpthread
  echo "built lib_$tag.so ($defs)"
done
