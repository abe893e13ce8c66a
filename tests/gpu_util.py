"""Helpers for the -m gpu parity tests: run the HIP path through the C ABI and fetch intermediates."""
import ctypes as C

import numpy as np
import torch

import gemmul8_amd as g
import oracle_lib as ol

NP2T = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
        np.dtype(np.complex64): torch.complex64, np.dtype(np.complex128): torch.complex128}


SAFE, REFERENCE = ol.FP8_BOUND_SAFE, ol.FP8_BOUND_REFERENCE


def select_fp8_bound_mode(mode):
    """Select the FP8 accurate-mode bound inflation on BOTH sides: the product (gemmul8_set_fp8_bound_mode) and the oracle.  SAFE = the
    product's default (engine-safe), REFERENCE = the oracle's default, (k+1)*2^-24 of src/find_max.hpp:82-96."""
    assert g.lib().gemmul8_set_fp8_bound_mode(int(mode)) >= 0
    ol.set_fp8_bound_mode(int(mode))


def restore_fp8_bound_defaults():
    g.lib().gemmul8_set_fp8_bound_mode(SAFE)
    ol.set_fp8_bound_mode(REFERENCE)


def setknob(monkeypatch, name, value):
    """Set (value=None: unset) one of the library's testing knobs for the rest of the test.  The library parses its knobs once
    (csrc/oz2_knobs.hpp), so the environment change is followed by gemmul8_reload_knobs; tests/conftest.py reloads again after the
    test, when monkeypatch has restored the environment."""
    if value is None:
        monkeypatch.delenv(name, raising=False)
    else:
        monkeypatch.setenv(name, str(value))
    g.lib().gemmul8_reload_knobs()


def to_dev(M):
    """numpy (rows x cols) matrix -> column-major device tensor of shape (cols, rows)."""
    return torch.from_numpy(np.ascontiguousarray(M.T)).cuda()


def from_dev(T):
    return T.cpu().numpy().T


def f6_plane_values(buf, rows, rows_img, kp):
    """Integer values [rows, kp] of one FP6 panel-image plane of the FP8 backend (gemmul8_layout.lo_format == 1; layout in
    csrc/oz2_gemm_f6.hip): row blocks of 256 (the last one holds rows_img - 256 * nb rows rounded up to 16), K-steps of 128 elements,
    panel = X region (16-byte slot q * Rp + r) + Y region (8-byte slot (q >> 1) * 2 Rp + 2 r + (q & 1)); a fragment packs 32 codes
    sign << 5 | |v| of 6 bits, little-endian."""
    out = np.zeros((rows, kp), np.int8)
    kt_n = kp // 128
    nb = (rows_img + 255) // 256
    blockbytes = 256 * (kp // 4 * 3)
    for tb in range(nb):
        nr = min(256, rows_img - 256 * tb)
        rp = 256 if tb < nb - 1 else (nr + 15) // 16 * 16
        for kt in range(kt_n):
            off = tb * blockbytes + kt * rp * 96
            panel = buf[off:off + rp * 96]
            X = panel[:64 * rp].reshape(4, rp, 16)
            Y = panel[64 * rp:].reshape(2, rp, 2, 8)
            frag = np.concatenate([X, np.stack([Y[0, :, 0], Y[0, :, 1], Y[1, :, 0], Y[1, :, 1]])], axis=2)  # [q][r][24]
            # four 6-bit codes per three bytes, little-endian bit order (c0 = b0[5:0], c1 = b1[3:0] b0[7:6], c2 = b2[1:0] b1[7:4], c3 = b2[7:2])
            b = frag.reshape(4, rp, 8, 3).astype(np.uint16)
            codes = np.stack([b[..., 0] & 63, (b[..., 0] >> 6 | b[..., 1] << 2) & 63, (b[..., 1] >> 4 | b[..., 2] << 4) & 63, b[..., 2] >> 2], axis=3).reshape(4, rp, 32).astype(np.int16)
            vals = np.where(codes & 32, -(codes & 31), codes & 31).astype(np.int8)  # [q][r][32]
            vals = vals.transpose(1, 0, 2).reshape(rp, 128)
            r0, r1 = 256 * tb, min(rows, 256 * tb + rp)
            if r1 > r0:
                out[r0:r1, 128 * kt:128 * kt + 128] = vals[:r1 - r0]
    return out


def f6_plane_image(vals, rows_img, kp, nbytes):
    """Inverse of f6_plane_values: the FP6 panel-image bytes (length nbytes) of integers vals[rows, kp], |v| <= 16; rows beyond vals are zero."""
    buf = np.zeros(nbytes, np.uint8)
    rows = vals.shape[0]
    kt_n = kp // 128
    nb = (rows_img + 255) // 256
    blockbytes = 256 * (kp // 4 * 3)
    for tb in range(nb):
        nr = min(256, rows_img - 256 * tb)
        rp = 256 if tb < nb - 1 else (nr + 15) // 16 * 16
        blk = np.zeros((rp, kp), np.int16)
        r0, r1 = 256 * tb, min(rows, 256 * tb + rp)
        if r1 > r0:
            blk[:r1 - r0] = vals[r0:r1]
        codes = (np.where(blk < 0, 32, 0) | np.abs(blk)).astype(np.uint8)                       # [rp][kp]
        bits = ((codes[..., None] >> np.arange(6, dtype=np.uint8)) & 1).astype(np.uint8)          # [rp][kp][6]
        frag = np.packbits(bits.reshape(rp, kt_n, 4, 32 * 6), axis=3, bitorder="little")          # [rp][kt][q][24]
        for kt in range(kt_n):
            f = frag[:, kt].transpose(1, 0, 2)                                                    # [q][rp][24]
            X = f[:, :, :16]                                                                      # 16-byte slot q * rp + r
            Y = np.zeros((2, rp, 2, 8), np.uint8)                                                 # 8-byte slot (q >> 1) * 2 rp + 2 r + (q & 1)
            for q in range(4):
                Y[q >> 1, :, q & 1] = f[q, :, 16:]
            off = tb * blockbytes + kt * rp * 96
            buf[off:off + 64 * rp] = X.ravel()
            buf[off + 64 * rp:off + 96 * rp] = Y.ravel()
    return buf


def e4m3_of_ints(v):
    """e4m3 byte (OCP, +0 for zero) of integers |v| <= 16: what the reference's planes hold (mod.hpp:159-189)."""
    a = np.abs(v.astype(np.int32))
    e = np.where(a > 0, np.floor(np.log2(np.maximum(a, 1))).astype(np.int32), 0)
    mant = np.where(a > 0, (a * 8) // (1 << e) - 8, 0)
    byte = np.where(a > 0, ((e + 7) << 3) | mant, 0) | np.where((v < 0) & (a > 0), 0x80, 0)
    return byte.astype(np.uint8)


def read_intermediates(work, code, backend, m, n, k, N):
    """Shifts, operand planes (decoded to one byte per element) and C_mid of a finished call, read from its workspace through gemmul8_get_layout."""
    cplx = code >= 2
    L = g.Layout()
    g.check(g.lib().gemmul8_get_layout(code, backend, m, n, k, N, work.data_ptr(), None, None, 0, 0, C.byref(L)))
    w = work.cpu().numpy()
    base = work.data_ptr()
    parts, nm = L.parts, L.num_mat

    def region(ptr, nbytes):
        off = ptr - base
        return w[off:off + nbytes]

    sftA = region(L.sftA, 2 * m).view(np.int16).copy()
    sftB = region(L.sftB, 2 * n).view(np.int16).copy()
    A_lo = np.zeros((parts, nm, m, k), np.uint8)
    B_lo = np.zeros((parts, nm, n, k), np.uint8)
    for p in range(parts):
        for q in range(nm):
            if L.lo_format == 1:  # FP6 panel images: decode to the integers and re-encode as the e4m3 bytes the oracle holds (zero as +0)
                va = f6_plane_values(region(L.A_lo + p * L.part_strideA + q * L.sizeA, L.sizeA), m, L.mp, L.kp)
                vb = f6_plane_values(region(L.B_lo + p * L.part_strideB + q * L.sizeB, L.sizeB), n, n, L.kp)
                assert not va[:, k:].any() and not vb[:, k:].any(), "k-padding of the FP6 planes must be zero"
                A_lo[p, q] = e4m3_of_ints(va[:, :k])
                B_lo[p, q] = e4m3_of_ints(vb[:, :k])
                continue
            pa = region(L.A_lo + p * L.part_strideA + q * L.sizeA, L.sizeA).reshape(L.mp, L.kp)
            A_lo[p, q] = pa[:m, :k]
            assert not pa[:m, k:].any(), "k-padding of A_lo must be zero"
            pb = region(L.B_lo + p * L.part_strideB + q * L.sizeB, L.sizeB).reshape(n, L.kp)
            B_lo[p, q] = pb[:, :k]
            assert not pb[:, k:].any(), "k-padding of B_lo must be zero"
    mid_dt = np.int8 if backend == g.INT8 else np.int16
    comps = 2 if cplx else 1
    isz = np.dtype(mid_dt).itemsize * comps
    Cm = np.zeros((N, n, m, comps), mid_dt)
    for t in range(N):
        pc = region(L.C_mid + t * L.sizeC * isz, L.sizeC * isz).view(mid_dt).reshape(n, L.mp, comps)
        Cm[t] = pc[:, :m, :]
    if not cplx:
        Cm = Cm[..., 0]
    return dict(sftA=sftA, sftB=sftB, A_lo=A_lo, B_lo=B_lo, C_mid=Cm, lo_format=int(L.lo_format))


def hip_gemm(A, B, N, fastmode=False, backend=g.INT8, opA="N", opB="N", alpha=1.0, beta=0.0, C0=None, want_intermediates=False,
             timers=False):
    """A, B: numpy arrays as stored (before op), like oracle_lib.gemm.  Returns C (numpy m x n) [, intermediates]."""
    dA, dB = to_dev(A), to_dev(B)
    dC = to_dev(C0.copy()) if C0 is not None else None
    m, k = (A.shape if opA == "N" else A.shape[::-1])
    n = B.shape[1] if opB == "N" else B.shape[0]
    cplx = A.dtype.kind == "c"
    tot, _, _ = g.work_size(cplx, backend, m, n, k, N)
    work = torch.zeros(tot, dtype=torch.uint8, device="cuda")
    Cd, tm, work = g.gemm(dA, dB, N, fastmode=fastmode, backend=backend, opA=opA, opB=opB, alpha=alpha, beta=beta, C_out=dC,
                          work=work, timers=timers)
    torch.cuda.synchronize()
    Cn = from_dev(Cd)
    if not want_intermediates:
        return (Cn, tm) if timers else Cn
    inter = read_intermediates(work, g._dtype_code(dA.dtype), backend, m, n, k, N)
    return (Cn, inter, tm) if timers else (Cn, inter)


def bits_equal(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    return a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8))


def shifts_close(dev, orc, what=""):
    """Device vs oracle shifts: identical except for rare +-1 at floor boundaries of the log2 approximation."""
    d = dev.astype(int) - orc.astype(int)
    bad = np.abs(d) > 1
    assert not bad.any(), f"{what}: shift differs by more than 1 at {np.nonzero(bad)[0][:5]}"
    nd = int((d != 0).sum())
    assert nd <= max(1, 0.02 * d.size), f"{what}: {nd}/{d.size} shifts differ from the oracle (log2 boundary cases should be rare)"
    return nd


def bounds_case(A, B, N, opA="N", opB="N", backend=g.INT8, skip_layout=False, bound_mode=SAFE):
    """Accurate-mode scaling phase, first half, BIT-EXACT against the oracle (rows a3 / a4 of SURVEY.md section 8):
    the 7-bit (INT8) / e4m3 round-up (FP8) bound planes of both operands, the preliminary shifts sft0 = maxUFP - ilogb(amax)
    and the row / column maxima of the bound product (int32; FP8: inflated floats), read from the workspace right after
    gemmul8_scale_bounds (scaling_accu_real.hpp:23-136,415-432, scaling.hpp:3-94, find_max.hpp:67-114).
    skip_layout: carve the workspace with enable_skip_scalA/B = 1, where the bound planes have their own slot behind the
    residue planes (gemmul8_real.hpp:101-104) instead of aliasing plane 0.
    bound_mode (FP8 backend): SAFE = the product's engine-safe inflation, REFERENCE = (k+1)*2^-24 of find_max.hpp:82-96; selected on both sides."""
    if backend == g.FP8:
        select_fp8_bound_mode(bound_mode)
    dA, dB = to_dev(A), to_dev(B)
    m, k = (A.shape if opA == "N" else A.shape[::-1])
    n = B.shape[1] if opB == "N" else B.shape[0]
    cplx = A.dtype.kind == "c"
    en = int(skip_layout)
    tot, _, _ = g.work_size(cplx, backend, m, n, k, N, en, en)
    work = torch.full((tot,), 0x5A, dtype=torch.uint8, device="cuda")
    L = g.Layout()
    code = g._dtype_code(dA.dtype)
    lib = g.lib()
    g.check(lib.gemmul8_get_layout(code, backend, m, n, k, N, work.data_ptr(), None, None, en, en, C.byref(L)))
    st = torch.cuda.current_stream().cuda_stream
    g.check(lib.gemmul8_scale_bounds(st, code, backend, g.OPS[opA], g.OPS[opB], m, n, k, dA.data_ptr(), dA.shape[1], dB.data_ptr(),
                                     dB.shape[1], N, 0, n, C.byref(L), 0, 0), "scale_bounds")
    torch.cuda.synchronize()
    w = work.cpu().numpy()
    base = work.data_ptr()
    parts = 3 if cplx else 1

    def planes(ptr, rows, rows_alloc, plane_bytes):
        out = np.zeros((parts, rows, k), np.uint8)
        for p in range(parts):
            off = ptr - base + p * plane_bytes
            pl = w[off:off + rows_alloc * L.kp].reshape(rows_alloc, L.kp)
            out[p] = pl[:rows, :k]
            assert not pl[:rows, k:].any(), "k-padding of a bound plane must be zero"
        return out
    Ab = planes(L.A_bound, m, L.mp, L.sizeA)
    Bb = planes(L.B_bound, n, n, L.sizeB)
    s0A = w[L.sftA - base:L.sftA - base + 2 * m].view(np.int16)
    s0B = w[L.sftB - base:L.sftB - base + 2 * n].view(np.int16)
    np_ = (n + 255) // 256 * 256
    mx = w[L.scratch - base:L.scratch - base + 4 * (L.mp + np_)]
    mdt = np.int32 if backend == g.INT8 else np.float32
    rmax, cmax = mx[:4 * m].view(mdt), mx[4 * L.mp:4 * L.mp + 4 * n].view(mdt)
    oA, o0A = ol.extract_bounds(A, opA, True, backend)
    oB, o0B = ol.extract_bounds(B, opB, False, backend)
    assert np.array_equal(Ab, oA), f"A bound planes differ in {np.sum(Ab != oA)} bytes"
    assert np.array_equal(Bb, oB), f"B bound planes differ in {np.sum(Bb != oB)} bytes"
    assert np.array_equal(s0A, o0A) and np.array_equal(s0B, o0B), "preliminary shifts sft0 differ"
    orm, ocm = ol.bound_maxima(oA, oB, backend)
    if backend == g.INT8:
        assert np.array_equal(rmax, orm), f"row maxima of the bound GEMM differ at {np.nonzero(rmax != orm)[0][:5]}"
        assert np.array_equal(cmax, ocm), f"column maxima of the bound GEMM differ at {np.nonzero(cmax != ocm)[0][:5]}"
        return 0
    # FP8: the accumulation of the e4m3 products belongs to the ENGINE (the reference leaves it to the vendor's FP8 GEMM and
    # inflates by (k+1)*2^-24, find_max.hpp:82-96, an IEEE FP32 summation bound).  gfx950's v_mfma_scale_f32_16x16x128_f8f6f4 does
    # not add like that (tools/ubench/f8_accum.hip, profiles/archive/r02_f8_mfma_accumulation.txt): the 128 products of an instruction are
    # summed in groups of 8, inside a group everything is aligned to the largest product and bits below 2^-13 of it are dropped;
    # group sums and the accumulator are added with ~22 bits, truncating.  A non-negative sum therefore comes out equal to the
    # exact one or LOW (1.2e-3 seen in a 1500-seed fuzz sweep), or above it by a few ulp only (<= 15 ulp over 18 K-steps in the
    # round-4 24000-case stress sweep: the accumulator add is not a pure truncation, at most about one ulp up per K-step;
    # tools/f8_bound_stress_dbg.py prints such cases).  The product's default inflation
    # ku = 7*2^-13 + 4(k+1)*2^-24 (gemmul8_set_fp8_bound_mode) covers that loss; what is asserted for real types is the GUARANTEE:
    # exact un-inflated maximum <= device value <= the oracle's inflated value of the exactly accumulated sum (+ one ulp per K-step).
    if bound_mode == REFERENCE:
        # the reference's formula on this engine: nothing is guaranteed one-sidedly (that is why it is not the product's default) -- the device's
        # maxima sit within the engine's accumulation loss (1.2e-3 seen; 2^-9 allowed) of the oracle's inflated exactly-accumulated values
        for d, o, what in ((rmax, orm, "row"), (cmax, ocm, "column")):
            d64, o64 = d.astype(np.float64), o.astype(np.float64)
            assert np.all(np.abs(d64 - o64) <= 2.0 ** -9 * np.abs(o64)), f"{what} maxima of the FP8 bound GEMM (reference formula) off by {np.max(np.abs(d64 - o64) / np.maximum(np.abs(o64), 1e-300))}"
        return int((rmax != orm).sum() + (cmax != ocm).sum())
    if not cplx:
        ex_r, ex_c = ol.bound_maxima_f8_exact(oA, oB)
        for d, o, ex, what in ((rmax, orm, ex_r, "row"), (cmax, ocm, ex_c, "column")):
            d64 = d.astype(np.float64)
            assert np.all(d64 >= ex), f"{what} maxima of the FP8 bound GEMM BELOW the exact sum by {np.min((d64 - ex) / np.maximum(ex, 1e-300))}"
            assert np.all(d64 <= o.astype(np.float64) * (1 + (L.kp // 128 + 4) * 2.0 ** -23)), f"{what} maxima of the FP8 bound GEMM above the inflated exact sum"
        return int((rmax != orm).sum() + (cmax != ocm).sum())
    # complex (round 4): T = C0 + C1 with C0 = (|Ar|-|Ai|)(|Br|-|Bi|) of both signs.  The default combination inflates C0 by
    # ku (|C0| + 2 s12) (oz2_gemm_f8.hip bound_ku), which covers the engine's loss on the MAGNITUDES of C0's terms, so the same one-sided
    # GUARANTEE holds: exact un-inflated max(T, C1) <= device value.  Above, the device may exceed the oracle's inflated value of the exact
    # sums by what the engine loses on C0's NEGATIVE terms (at most eps (T + C1), inflated again): a 2^-9 band.
    ex_r, ex_c = ol.bound_maxima_f8_exact_cplx(oA, oB)
    for d, o, ex, what in ((rmax, orm, ex_r, "row"), (cmax, ocm, ex_c, "column")):
        d64 = d.astype(np.float64)
        assert np.all(d64 >= ex), f"{what} maxima of the complex FP8 bound BELOW the exact sum by {np.min((d64 - ex) / np.maximum(ex, 1e-300))}"
        assert np.all(d64 <= o.astype(np.float64) * (1 + 2.0 ** -9)), f"{what} maxima of the complex FP8 bound above the oracle's inflated value"
    return int((rmax != orm).sum() + (cmax != ocm).sum())


def parity_case(A, B, N, fastmode, opA="N", opB="N", alpha=1.0, beta=0.0, C0=None, backend=g.INT8, bound_mode=SAFE):
    """Full bit-exact parity of one case: accurate mode's bound planes / sft0 / bound maxima (exact), shifts (tolerant),
    planes, C_mid, C (exact given the device's shifts).  FP8 backend: `bound_mode` is selected on BOTH sides (SAFE = the product's
    default inflation, REFERENCE = the reference's formula = the oracle's default)."""
    if backend == g.FP8:
        select_fp8_bound_mode(bound_mode)
    if not fastmode:
        bounds_case(A, B, N, opA=opA, opB=opB, backend=backend, bound_mode=bound_mode)
    Cd, it = hip_gemm(A, B, N, fastmode=fastmode, backend=backend, opA=opA, opB=opB, alpha=alpha, beta=beta, C0=C0, want_intermediates=True)
    # oracle with its own shifts -> compare shifts
    _, ito = ol.gemm(A, B, N, fastmode=fastmode, backend=backend, opA=opA, opB=opB, alpha=alpha, beta=beta, C0=C0, want_intermediates=True)
    nd = shifts_close(it["sftA"], ito["sftA"], "sftA") + shifts_close(it["sftB"], ito["sftB"], "sftB")
    # oracle fed with the device's shifts -> everything downstream is bit-exact
    Co, ito = ol.gemm(A, B, N, fastmode=fastmode, backend=backend, opA=opA, opB=opB, alpha=alpha, beta=beta, C0=C0,
                      sftA_in=it["sftA"], sftB_in=it["sftB"], want_intermediates=True)
    if it["lo_format"] == 1:  # the FP6 images are compared as INTEGERS (f6_plane_values: a zero decodes to 0 whichever sign bit its code carries -- the four-k writer emits +0, the
        # hardware pack of the lane-per-fragment writer keeps the sign of a -0): the oracle's -0 byte (rint of a small negative quotient) compares as +0
        for key in ("A_lo", "B_lo"):
            ito[key] = np.where(ito[key] == 0x80, 0, ito[key]).astype(np.uint8)
    assert np.array_equal(it["A_lo"], ito["A_lo"]), "A_lo planes differ"
    assert np.array_equal(it["B_lo"], ito["B_lo"]), "B_lo planes differ"
    assert np.array_equal(it["C_mid"], ito["C_mid"]), "C_mid planes differ"
    assert bits_equal(Cd, Co), f"final C differs in {np.sum(Cd != Co)} elements"
    return nd


def embed(M, ld_extra, base_off, rng, tail=5):
    """Column-major matrix M (numpy rows x cols) placed inside a larger 1-D buffer of its element type: base offset `base_off` ELEMENTS
    (so the base pointer is only element-aligned), leading dimension rows + ld_extra, `tail` elements behind the last column; every element outside the window holds
    random finite garbage.  Returns (buffer, ld)."""
    rows, cols = M.shape
    ld = rows + ld_extra
    size = base_off + ld * cols + tail
    buf = (rng.random(size) * 1e3 - 500).astype(M.dtype)
    if M.dtype.kind == "c":
        buf = (buf + 1j * (rng.random(size) * 1e3 - 500)).astype(M.dtype)
    win = buf[base_off:base_off + ld * cols].reshape(cols, ld)
    win[:, :rows] = M.T
    return buf, ld


def window_mask(size, base_off, ld, rows, cols):
    """Boolean mask over a buffer of `embed`: True inside the rows x cols window."""
    mk = np.zeros(size, bool)
    mk[base_off:base_off + ld * cols].reshape(cols, ld)[:, :rows] = True
    return mk


def parity_case_embedded(A, B, C0, N, fastmode, opA, opB, alpha, beta, backend, ld_extra, base_off, rng, bound_mode=SAFE):
    """`parity_case` on sub-matrix views: A, B, C0 (numpy, as stored) are embedded in larger buffers with ld = rows + ld_extra and a base
    pointer `base_off` elements into the allocation; the HIP path (C ABI, explicit lda / ldb / ldc) and the oracle
    (oracle_lib.gemm_embedded) run on the same bytes.  Asserted: shifts (tolerant), then with the device's shifts planes / C_mid and
    the WHOLE C buffer bit for bit -- i.e. the m x n window equals the oracle's and every byte outside it is untouched; A's and B's
    buffers unchanged.  beta != 0 updates C in place inside the larger matrix."""
    if backend == g.FP8:
        select_fp8_bound_mode(bound_mode)
    m, k = (A.shape if opA == "N" else A.shape[::-1])
    n = B.shape[1] if opB == "N" else B.shape[0]
    exa, exb, exc = (ld_extra if np.ndim(ld_extra) else (ld_extra,) * 3)
    offa, offb, offc = (base_off if np.ndim(base_off) else (base_off,) * 3)
    bufA, lda = embed(A, exa, offa, rng)
    bufB, ldb = embed(B, exb, offb, rng)
    bufC, ldc = embed(C0, exc, offc, rng)
    dt = A.dtype
    cplx = dt.kind == "c"
    code = ol.DT[dt]
    dA, dB, dC = (torch.from_numpy(x.copy()).cuda() for x in (bufA, bufB, bufC))
    tot, _, _ = g.work_size(cplx, backend, m, n, k, N)
    work = torch.zeros(tot, dtype=torch.uint8, device="cuda")
    al, be = np.array([alpha], dtype=dt), np.array([beta], dtype=dt)
    isz = dt.itemsize
    rc = g.lib().gemmul8_gemm(torch.cuda.current_stream().cuda_stream, code, backend, g.OPS[opA], g.OPS[opB], m, n, k, al.ctypes.data,
                              dA.data_ptr() + offa * isz, lda, dB.data_ptr() + offb * isz, ldb, be.ctypes.data, dC.data_ptr() + offc * isz, ldc, N,
                              int(fastmode), work.data_ptr(), None, None, 0, 0, 0, 0, None)
    g.check(rc, "gemmul8_gemm")
    torch.cuda.synchronize()
    it = read_intermediates(work, code, backend, m, n, k, N)
    assert bits_equal(dA.cpu().numpy(), bufA) and bits_equal(dB.cpu().numpy(), bufB), "an operand buffer was written"
    Cdev = dC.cpu().numpy()
    # the oracle with its own shifts (shift comparison), then with the device's (everything downstream bit-exact)
    oC = bufC.copy()
    ito = ol.gemm_embedded(bufA, offa, lda, bufB, offb, ldb, oC, offc, ldc, m, n, k, N, fastmode, backend, opA, opB, alpha, beta)
    nd = shifts_close(it["sftA"], ito["sftA"], "sftA") + shifts_close(it["sftB"], ito["sftB"], "sftB")
    oC = bufC.copy()
    ito = ol.gemm_embedded(bufA, offa, lda, bufB, offb, ldb, oC, offc, ldc, m, n, k, N, fastmode, backend, opA, opB, alpha, beta,
                           sftA_in=it["sftA"], sftB_in=it["sftB"])
    if it["lo_format"] == 1:
        for key in ("A_lo", "B_lo"):
            ito[key] = np.where(ito[key] == 0x80, 0, ito[key]).astype(np.uint8)
    assert np.array_equal(it["A_lo"], ito["A_lo"]), "A_lo planes differ"
    assert np.array_equal(it["B_lo"], ito["B_lo"]), "B_lo planes differ"
    assert np.array_equal(it["C_mid"], ito["C_mid"]), "C_mid planes differ"
    mk = window_mask(bufC.size, offc, ldc, m, n)
    assert bits_equal(Cdev[~mk], bufC[~mk]), "bytes of the enclosing C buffer outside the m x n window were written"
    assert bits_equal(oC[~mk], bufC[~mk])
    assert bits_equal(Cdev[mk], oC[mk]), f"final C differs in {np.sum(Cdev[mk] != oC[mk])} elements"
    return nd, it["lo_format"]
