#!/usr/bin/env python3
"""Check tools/gen_tables.py's big-integer tables against the reference's literals.

Runs ONLY in the build container (reads /root/reference/GEMMul8/src/table.hpp:12-838); the GPU
box never sees /root/reference.  Nothing is copied: the reference file is parsed for numeric
literals, each is compared bit-for-bit with the value generated from the derivation rules, and a
one-line-per-table report is printed (committed as tests/golden/tables_vs_reference.txt).
"""
import re
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_tables as G

REF = "/root/reference/GEMMul8/src/table.hpp"


def hexfloats(text):
    return [float.fromhex(m.replace("F", "")) for m in re.findall(r"-?0x[01]\.[0-9a-f]+p[+-]?\d+F?", text)]


def block(src, start_pat, end_pat="};"):
    i = src.index(start_pat)
    j = src.index(end_pat, i)
    return src[i:j]


def main():
    src = open(REF).read()
    ns = {}
    # split into namespace INT8 / FP8 chunks in order of appearance per table kind
    report = []
    ok_all = True

    def cmp(name, got, exp):
        nonlocal ok_all
        ok = len(got) == len(exp) and all(a == b for a, b in zip(got, exp))
        ok_all &= ok
        report.append(f"{name:28s} {len(exp):4d} values  {'bit-identical' if ok else 'MISMATCH'}")
        if not ok:
            for i, (a, b) in enumerate(zip(got, exp)):
                if a != b:
                    report.append(f"    first mismatch at {i}: generated {a.hex()} reference {b.hex()}")
                    break

    tabs = {be: G.build(be) for be in ("INT8", "FP8")}

    # moduli
    for be in ("INT8", "FP8"):
        ref = [int(x) for x in re.findall(r"moduli<gemmul8::Backend::%s, \d+>\s*=\s*(\d+);" % be, src)]
        cmp(f"moduli {be}", [float(x) for x in G.MODULI[be]], [float(x) for x in ref])

    # P (double-double, negative)
    pb = [m.start() for m in re.finditer(r"constexpr double2 P\[19\]", src)]
    for be, st in zip(("INT8", "FP8"), pb):
        vals = hexfloats(src[st:src.index("};", st)])
        exp = []
        for h, l in zip(tabs[be]["P_hi"], tabs[be]["P_lo"]):
            exp += [h, l]
        cmp(f"P (hi,lo) {be}", exp, vals)

    ib = [m.start() for m in re.finditer(r"constexpr double invP\[19\]", src)]
    for be, st in zip(("INT8", "FP8"), ib):
        vals = hexfloats(src[st:src.index("};", st)])
        cmp(f"invP {be}", tabs[be]["invP"], vals)

    for be in ("INT8", "FP8"):
        vals = [float.fromhex(x) for x in re.findall(r"log2P<gemmul8::Backend::%s, \d+>\s*=\s*(0x[0-9a-fp.+-]+)F;" % be, src)]
        cmp(f"log2P {be}", [float(x) for x in tabs[be]["log2P"]], vals)

    qb = [m.start() for m in re.finditer(r"inline constexpr double qPi_1\[19\]\[20\]", src)]
    for be, st in zip(("INT8", "FP8"), qb):
        end = src.index("\n};", st)
        vals = hexfloats(src[st:end])
        exp = []
        for N in range(2, 21):
            exp += tabs[be]["qpi1"][N - 2][:N]
        cmp(f"qPi_1 {be}", exp, vals)

    qb2 = [m.start() for m in re.finditer(r"inline constexpr double2 qPi_2\[\d+\]\[20\]", src)]
    for be, st in zip(("INT8", "FP8"), qb2):
        end = src.index("\n};", st)
        vals = hexfloats(src[st:end])
        exp = []
        for N in range(G.P_IS_DOUBLE[be] + 1, 21):
            for t in range(N):
                exp += [tabs[be]["qpi2h"][N - 2][t], tabs[be]["qpi2l"][N - 2][t]]
        cmp(f"qPi_2 (h,l) {be}", exp, vals)

    # mod_pow2: INT8 rows i<->p_{i+1}, entries 2^(j+7); FP8 row0<->p0, row i<->p_{i+1}, 2^(j+8)
    st = src.index("constexpr int8_t mod_pow2_h[19][57]")
    vals = [int(x) for x in re.findall(r"-?\d+", src[src.index("{", st):src.index("\n};", st)])]
    exp = []
    for i in range(19):
        exp += [tabs["INT8"]["pow2"][i + 1][j + 7] for j in range(57)]
    cmp("mod_pow2 INT8", [float(x) for x in exp], [float(x) for x in vals])
    st = src.index("constexpr int16_t mod_pow2_h[19][64]")
    vals = [int(x) for x in re.findall(r"-?\d+", src[src.index("{", st):src.index("\n};", st)])]
    exp = []
    for i in range(19):
        row = 0 if i == 0 else i + 1
        exp += [G.sym(pow(2, j + 8, G.MODULI["FP8"][row]), G.MODULI["FP8"][row]) for j in range(64)]
    cmp("mod_pow2 FP8", [float(x) for x in exp], [float(x) for x in vals])

    print("\n".join(report))
    print("ALL TABLES BIT-IDENTICAL" if ok_all else "TABLE MISMATCH")
    return 0 if ok_all else 1


if __name__ == "__main__":
    sys.exit(main())
