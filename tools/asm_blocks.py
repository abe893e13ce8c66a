#!/usr/bin/env python3
"""Per-basic-block instruction census of one kernel in a -save-temps .s file: where the MFMAs, the scratch (spill) traffic, the FP64
FMAs and the LDS-DMA instructions sit.  usage: tools/asm_blocks.py <file.s> <substring of the mangled kernel name>"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
want = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and want in l.split(":")[0])
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith("\t.section") or lines[i].startswith(".Lfunc_end"))
keys = ["v_mfma", "scratch_load", "scratch_store", "v_fma_f64", "global_load_lds", "s_barrier", "v_writelane", "v_readlane"]
blk, stats = "entry", {}
order = []
for l in lines[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blk = m.group(1)
    if blk not in stats:
        stats[blk] = dict.fromkeys(keys, 0)
        stats[blk]["n"] = 0
        order.append(blk)
    if l.startswith("\t") and not l.startswith("\t."):
        stats[blk]["n"] += 1
    for k in keys:
        if k in l:
            stats[blk][k] += 1
print(lines[start].split(":")[0], "lines", end - start)
for b in order:
    st = stats[b]
    if any(st[k] for k in keys):
        print(f"  {b:12s} n={st['n']:5d} " + " ".join(f"{k}={st[k]}" for k in keys if st[k]))
