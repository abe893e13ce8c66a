#!/usr/bin/env python3
"""Regression golden for the reference's known-answer sample (tests/golden/kat_dgemm_4x5x3.json): the oracle's result BIT
patterns, shifts and residue planes for INT8 N=15 / FP8 N=13 in both modes -> tests/golden/kat_result_bits.json.
This pins today's restatement against drift (a regression pin, not reference output: the reference cannot run here)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol
d = json.load(open(os.path.join(ROOT, "tests/golden/kat_dgemm_4x5x3.json")))
A = np.array([float.fromhex(x) for x in d["A"]]).reshape((4, 5), order="F")
B = np.array([float.fromhex(x) for x in d["B"]]).reshape((5, 3), order="F")
out = {"source": "oracle/oz2_oracle.c on tests/golden/kat_dgemm_4x5x3.json (tools/make_kat_result_bits.py)"}
for name, be, N in (("INT8", ol.INT8, 15), ("FP8", ol.FP8, 13)):
    for fast in (False, True):
        C, it = ol.gemm(A, B, N, fastmode=fast, backend=be, want_intermediates=True)
        out[f"{name}_N{N}_{'fast' if fast else 'accurate'}"] = {
            "C": [float(x).hex() for x in C.flatten(order="F")], "sftA": it["sftA"].tolist(), "sftB": it["sftB"].tolist(),
            "C_mid": it["C_mid"].flatten().tolist()}
json.dump(out, open(os.path.join(ROOT, "tests/golden/kat_result_bits.json"), "w"), indent=1)
print("ok")
