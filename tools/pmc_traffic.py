#!/usr/bin/env python3
"""Turn two rocprofv3 counter-collection passes (FETCH_SIZE, WRITE_SIZE -- separate passes, tools/pmc_fetch.txt and
tools/pmc_write.txt) into a per-kernel HBM-side traffic summary: profiles/<tag>_pmc_traffic.json.

Units/corrections (guides/MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE counts 64 B per
128-B request for 16-B-per-lane streaming reads, so it is DOUBLED for kernels whose reads are all 16 B/lane (the LDS-DMA
GEMM kernels; calibrated here on stage_kmajor_kernel<double,0>, 8-B/lane reads of a 512 MiB operand = 524,411 KiB
reported, i.e. no correction for narrower reads).  WRITE_SIZE matched the known byte count of the GEMM output exactly.

usage: tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json
"""
import collections
import csv
import glob
import json
import sys

X2 = ("gemm_i8_kernel", "gemm_f8_kernel")  # all global reads are global_load_lds_dwordx4 (16 B/lane)


def collect(root, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "oz2::" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch = collect(sys.argv[1], "FETCH_SIZE")
write = collect(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    f_kib, w_kib = fetch.get(k, 0.0), write.get(k, 0.0)
    corr = 2.0 if any(x in k for x in X2) else 1.0
    out[k] = {"FETCH_SIZE_KiB": f_kib, "WRITE_SIZE_KiB": w_kib, "fetch_correction": corr,
              "hbm_side_bytes_per_launch": (f_kib * corr + w_kib) * 1024.0}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    print(f"{k[:70]:70s} {v['hbm_side_bytes_per_launch'] / 1e9:8.3f} GB/launch")
