#!/usr/bin/env python3
"""Extract the reference's only known-answer vectors (sample/dgemm_cuBLAS_int8.cu:24-38: hA 4x5,
hB 5x3, hC_exact 4x3, hex-float literals, column-major) into tests/golden/kat_dgemm_4x5x3.json.
Runs only in the build container (needs /root/reference); the JSON (data only) is committed."""
import json, os, re
src = open("/root/reference/GEMMul8/sample/dgemm_cuBLAS_int8.cu").read()
def vec(name):
    body = src[src.index(name + " = {"):]
    body = body[:body.index("};")]
    return re.findall(r"-?0x[0-9a-f.]+p[+-]?\d+", body)
out = dict(source="GEMMul8/sample/dgemm_cuBLAS_int8.cu:24-38 (N=15 accurate INT8); dgemm_cuBLASLt_fp8.cu:16 (N=13 FP8)",
           m=4, n=3, k=5, A=vec("hA"), B=vec("hB"), C_exact=vec("hC_exact"))
assert len(out["A"]) == 20 and len(out["B"]) == 15 and len(out["C_exact"]) == 12
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
json.dump(out, open(os.path.join(root, "tests/golden/kat_dgemm_4x5x3.json"), "w"), indent=1)
print("ok")
