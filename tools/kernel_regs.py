#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS / kernel-argument bytes of every kernel of one .hip file of gemmul8_amd/csrc (compiled with the product flags and
-save-temps into a scratch directory).  usage: tools/kernel_regs.py oz2_scale.hip [name filter]"""
import os, re, subprocess, sys, tempfile
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "gemmul8_amd", "csrc")
d = tempfile.mkdtemp()
cmd = ["/opt/rocm/bin/hipcc", "-std=c++20", "-O3", "-fPIC", "-Wno-invalid-offsetof", "--offload-arch=gfx950", "-ffp-contract=off",
       "-DOCML_BASIC_ROUNDED_OPERATIONS", "-DOZ2_PRODUCT_BUILD", "-w", "-I" + root, "-c", os.path.join(root, src), "-o", os.path.join(d, "x.o"), "-save-temps=obj"]
subprocess.run(cmd, check=True, cwd=d)
asm = [f for f in os.listdir(d) if f.endswith(".s") and "gfx950" in f][0]
t = open(os.path.join(d, asm)).read()
for b in t.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)
    dn = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    if flt in dn:
        print(f"{dn[:80]:80s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>4s} lds {g('group_segment_fixed_size'):>6s} kernarg {g('kernarg_segment_size')}")
