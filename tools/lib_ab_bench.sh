#!/bin/bash
# run `python bench.py <args>` once per library build given after "--" and print one line each (crude A/B across builds)
cd $GRAFT_REPO_ROOT
ARGS=(); while [ "$1" != "--" ]; do ARGS+=("$1"); shift; done; shift
cp gemmul8_amd/lib/libgemmul8.so /tmp/keep.so
for round in 1 2; do for f in "$@"; do cp $f gemmul8_amd/lib/libgemmul8.so 2>/dev/null || cp /tmp/keep.so gemmul8_amd/lib/libgemmul8.so; echo -n "$round $(basename $f): "; python bench.py "${ARGS[@]}" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'TFLOPS', d.get('phase_ms') or d['roofline'].get('launch_ms'))"; done; done
cp /tmp/keep.so gemmul8_amd/lib/libgemmul8.so
