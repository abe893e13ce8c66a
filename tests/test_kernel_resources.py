"""Compile-time properties of the MFMA kernels that their speed depends on and that a source change can silently lose ("the
epilogue's scratch"): register budget, scratch size, no spill traffic in any block that holds the MFMA loop, no vector-memory wait inside the
K loops of the consumer waves.  Compiles the two GEMM translation units with the product flags and -save-temps (about 40 s); no GPU needed."""
import os
import re
import shutil
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gemmul8_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["-std=c++20", "-O3", "-fPIC", "-Wno-invalid-offsetof", "--offload-arch=gfx950", "-ffp-contract=off", "-DOCML_BASIC_ROUNDED_OPERATIONS",
         "-DOZ2_PRODUCT_BUILD", "-w", "-I" + CSRC]

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")


def _asm(src):
    d = tempfile.mkdtemp()
    try:
        subprocess.run([HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", os.path.join(d, "x.o"), "-save-temps=obj"], check=True, cwd=d)
        name = [f for f in os.listdir(d) if f.endswith(".s") and "gfx950" in f][0]
        return open(os.path.join(d, name)).read()
    finally:
        shutil.rmtree(d, ignore_errors=True)


def _kernels(text):
    """{demangled-ish name: (vgprs, scratch bytes, [basic blocks as lists of instruction lines])}"""
    meta = {}
    for b in text.split("  - .agpr_count:")[1:]:
        g = lambda k: re.search(r"\." + k + r":\s+(\S+)", b).group(1)
        meta[g("name")] = (int(g("vgpr_count")), int(g("private_segment_fixed_size")))
    out = {}
    lines = text.split("\n")
    for name, (vg, sc) in meta.items():
        start = next(i for i, l in enumerate(lines) if l.startswith(name + ":"))
        end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
        blocks, cur = [], []
        for l in lines[start:end]:
            if re.match(r"^\.LBB\d+_\d+:", l):
                blocks.append(cur)
                cur = []
            elif l.startswith("\t") and not l.startswith("\t.") and not l.lstrip().startswith(";"):
                cur.append(l.strip())
        blocks.append(cur)
        out[name] = (vg, sc, blocks)
    return out


@pytest.fixture(scope="module")
def i8():
    return _kernels(_asm("oz2_gemm_i8.hip"))


@pytest.fixture(scope="module")
def f8():
    return _kernels(_asm("oz2_gemm_f8.hip"))


def _mfma_blocks(blocks):
    return [b for b in blocks if sum("v_mfma" in l for l in b) >= 32]


def test_int8_kernels_registers_and_scratch(i8):
    ks = {n: v for n, v in i8.items() if "gemm_i8_kernel" in n}
    assert len(ks) == 7, sorted(ks)   # MOD / CPLX x {K-step barrier small-K, K-step barrier, ping-pong} + the bound GEMM (a -DOZ2_MOD256=1 build adds EPI_MOD256)
    for n, (vg, sc, blocks) in ks.items():
        assert vg <= 168, (n, vg)     # 12 waves per CU = 3 per SIMD
        residue = "ILi0E" in n or "ILi2E" in n or "ILi3E" in n
        # the residue kernels keep a handful of kernel-lifetime values in scratch (entry / tile start); the epilogues spill nothing
        assert sc <= (32 if residue else 256), (n, sc)
        loops = _mfma_blocks(blocks)
        assert loops, n
        for b in loops:
            assert not any("scratch_" in l for l in b), (n, "spill traffic in an MFMA block")
            # consumer waves: no vector-memory wait in the K loop (the producers' waits sit in their own blocks, without MFMAs)
            assert not any(re.match(r"s_waitcnt vmcnt", l) for l in b), (n, "vmcnt wait in an MFMA block")


def test_int8_complex_combine_loads_are_not_serialised(i8):
    """All sixteen X / Y loads of a tile are issued before the first of them is waited for, and no scratch reload sits between them and the stores."""
    for n, (vg, sc, blocks) in i8.items():
        if "gemm_i8_kernelILi2E" not in n:
            continue
        flat = [l for b in blocks for l in b]
        loads = [i for i, l in enumerate(flat) if l.startswith("global_load_dwordx4")]
        assert len(loads) % 16 == 0 and loads, (n, len(loads))   # one epilogue per consumer half
        for g0 in range(0, len(loads), 16):
            first, last = loads[g0], loads[g0 + 15]
            between = flat[first:last + 1]
            assert not any(l.startswith("s_waitcnt vmcnt") or "scratch_" in l or l.startswith("global_store") for l in between), n


def test_fp8_kernels_registers_and_scratch(f8):
    ks = {n: v for n, v in f8.items() if "gemm_f8_kernel" in n}
    assert len(ks) == 7, sorted(ks)
    for n, (vg, sc, blocks) in ks.items():
        assert vg <= 256 and sc == 0, (n, vg, sc)   # 8 waves per CU = 2 per SIMD
        for b in _mfma_blocks(blocks):
            assert not any("scratch_" in l for l in b), n


@pytest.fixture(scope="module")
def f6():
    return _kernels(_asm("oz2_gemm_f6.hip"))


def test_fp6_kernels_registers_scratch_and_fragment_reads(f6):
    """The FP6 residue-GEMM kernel (round 5): 8 self-pipelined waves of 256 registers.  Its K-step block -- 32 MFMAs around the one barrier -- must
    hold no spill traffic, every MFMA must read six-register fragments (cbsz = blgp = 2), and no two 8-byte fragment reads may fuse into a
    ds_read2_b64 (its four registers would have to be waited for and copied in the middle of the MFMA stream: the reads are inline assembly)."""
    ks = {n: v for n, v in f6.items() if "gemm_f6_kernel" in n}
    assert len(ks) == 5, sorted(ks)   # EPI_PART, EPI_FINAL, EPI_FINAL_CPLX (three-launch form), EPI_FUSED, EPI_FUSED_CPLX (one three-segment tile loop)
    for n, (vg, sc, blocks) in ks.items():
        assert vg <= 256, (n, vg)
        # the last K-step of a tile is scheduled together with the head of the epilogue, which keeps a few values of the tile in scratch (per tile, not per K-step)
        assert sc <= (0 if ("ILi0E" in n or "ILi7E" in n) else 256), (n, sc)
        flat = [l for b in blocks for l in b]
        mf = [l for l in flat if "v_mfma_scale" in l]
        assert mf and all("cbsz:2 blgp:2" in l for l in mf), n
        for l in mf:
            a, b = re.findall(r"v\[(\d+):(\d+)\]", l)[1:3]
            assert int(a[1]) - int(a[0]) == 5 and int(b[1]) - int(b[0]) == 5, (n, l)
        steady = [b for b in blocks if sum("v_mfma" in l for l in b) >= 32 and any(l.startswith("s_barrier") for l in b) and not any("scratch_" in l for l in b)]
        assert steady, (n, "no clean K-step block")
        for b in steady:
            assert sum(l.startswith("s_barrier") for l in b) == 1, n   # ONE workgroup barrier per K-step
            assert not any("ds_read2_b64" in l for l in b), n         # (a tile's LAST K-step uses plain loads, which may fuse: once per tile)
            # the inline-asm 8-byte reads are invisible to the compiler's waitcnt pass: their destinations must be the upper third of an operand tuple
            # directly (a register copy placed between the read and the K-step's own wait would copy stale registers), and the LDS-DMA must use
            # SGPR base + VGPR offset addressing
            dst = set()
            for l in b:
                m = re.match(r"ds_read_b64 v\[(\d+):(\d+)\]", l)
                if m:
                    dst.update(range(int(m.group(1)), int(m.group(2)) + 1))
            assert dst, n
            for l in b:
                if l.startswith("v_mov_b32") or l.startswith("v_mov_b64") or l.startswith("v_accvgpr"):
                    srcs = [int(x) for x in re.findall(r"v(\d+)", l.split(",", 1)[1])] if "," in l else []
                    assert not (set(srcs) & dst), (n, l)
            tops = set()
            for l in b:
                if "v_mfma_scale" in l:
                    for lo, hi in re.findall(r"v\[(\d+):(\d+)\]", l)[1:3]:
                        tops.update((int(hi) - 1, int(hi)))
            assert dst <= tops, (n, sorted(dst - tops))
            dma = [l for l in b if l.startswith("global_load_lds")]
            assert dma and all(re.match(r"global_load_lds_dwordx4 v\d+, s\[", l) for l in dma), (n, dma[:2])


def test_fp6_inline_asm_reads_are_never_spilled_or_copied_in_flight():
    """The FP6 kernel's fragment reads are inline assembly (see F6_FRAG): the compiler does not know that their destination registers are in flight until
    the kernel's own lgkmcnt wait.  It may therefore spill or copy such a register right behind the read -- storing what the register held BEFORE the data
    landed (found in round 5: the fused complex kernel stored the aH fragments of a tile's last K-step, which runs into the epilogue, to scratch one
    instruction after reading them: a tenth of the imaginary parts wrong, differently on every run).  Walk every f6 kernel's instruction stream: between an
    inline-asm ds_read and the next lgkmcnt(0) wait no scratch store, register move or non-MFMA VALU instruction may read its destination."""
    text = _asm("oz2_gemm_f6.hip")
    lines = text.split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_ZN3oz2\w*gemm_f6_kernel\w+:", l)]
    assert len(starts) == 5
    for st in starts:
        end = next(i for i in range(st + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
        in_asm, inflight, nreads = False, set(), 0
        for l in lines[st:end]:
            t = l.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith(";") or t.startswith("."):
                continue
            if t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                inflight.clear()
                continue
            m = re.match(r"ds_read_b(64|128) v\[(\d+):(\d+)\]", t)
            if m and in_asm:
                inflight.update(range(int(m.group(2)), int(m.group(3)) + 1))
                nreads += 1
                continue
            if t.startswith("v_mfma") or t.startswith("s_") or t.startswith("ds_read") or t.startswith("global_load_lds"):
                continue   # (the MFMAs behind the kernel's partial waits read landed fragments: the region order the waits count is checked on the GPU, bit for bit)
            used = set()
            ops = t.split(None, 1)[1] if " " in t else ""
            if t.startswith("scratch_store") or t.startswith("global_store"):
                srcs = ops
            else:
                srcs = ops.split(",", 1)[1] if "," in ops else ""
            for a, b in re.findall(r"v\[(\d+):(\d+)\]", srcs):
                used.update(range(int(a), int(b) + 1))
            used.update(int(x) for x in re.findall(r"\bv(\d+)\b", srcs))
            assert not (used & inflight), (lines[st][:60], t)
        assert nreads >= 32, (lines[st][:60], nreads)


def test_fp6_quantise_lane_kernel_uses_the_hardware_pack_and_two_waves():
    """quantise_f6_pair_kernel (csrc/oz2_scale.hip, round 5): a lane converts its 32 values with v_cvt_scalef32_2xpk16_fp6_f32 -- no sign-magnitude codes
    built by hand, no fragment operand copied into place (the 16-float operands must be allocated where the values are produced) --, the even-modulus tie
    fix is a branch and not 32 selects per modulus, nothing spills, and the register budget leaves two waves per SIMD."""
    ks = _kernels(_asm("oz2_scale.hip"))
    names = [n for n in ks if "quantise_f6_pair_kernel" in n]
    assert len(names) == 2, names  # float, double
    for n in names:
        vg, sc, blocks = ks[n]
        assert sc == 0, (n, sc)
        assert vg <= 256, (n, vg)
        cvt = [b for b in blocks if any("v_cvt_scalef32_2xpk16_fp6_f32" in l for l in b)]
        assert cvt, n
        for b in cvt:
            assert sum(l.startswith("v_mov_b32") for l in b) <= 8, (n, "fragment operands are copied into place")
        hot = [b for b in blocks if sum(l.startswith("v_pk_fma_f32") or l.startswith("v_fma_f32") for l in b) >= 32 and not any("v_cmp_eq_f32" in l for l in b)]
        assert hot, (n, "no residue block without the tie compare: the even-modulus fix runs for every modulus")


def test_inline_asm_stores_carry_their_wait_states(i8):
    """store16_cols (oz2_gemm_i8_epi.hpp) issues its 16-byte residue stores from an asm block: invisible to the compiler's hazard recognizer, which on gfx950 must keep
    two wait states between a store of more than 8 bytes and a VALU write to its data registers.  Round 6 found that hazard live (a kernel re-used the registers
    right behind the block; non-temporal stores wrote half-overwritten data).  Every such store in every INT8 kernel must be followed by `s_nop 1` inside its block."""
    n_stores = 0
    for n, (vg, sc, blocks) in i8.items():
        flat = [l for b in blocks for l in b]
        for i, l in enumerate(flat):
            if l.startswith("global_store_dwordx4") and i >= 3 and flat[i - 1].startswith("s_and_b64 exec"):   # the asm block's store (exec narrowed right before it)
                n_stores += 1
                assert flat[i + 1].split() == ["s_nop", "1"], (n, flat[i:i + 3])
    assert n_stores >= 7 * 16, n_stores
