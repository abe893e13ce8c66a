"""Host model of the workgroup -> tile map of the persistent GEMM kernels (oz2_gemm_common.hpp map_tile): for every shape, plane count
and column-block width the virtual workgroup ids 0 .. total-1 must hit every (plane, tile-row, tile-column) exactly once.  The GPU
parity tests run the real thing (GEMMUL8_MAP_COLBLOCK forces a width); this pins the arithmetic without a GPU."""
import pytest


def map_tile(bid, nwg, tiles_m, tiles_n, colblock):
    tpp = tiles_m * tiles_n
    xcd, idx = bid & 7, bid >> 3
    fc = (nwg >> 3) >> 5                      # full chunks of 256
    if idx < fc * 32:
        bid = (idx >> 5) * 256 + xcd * 32 + (idx & 31)
    else:
        rem = nwg - fc * 256
        q, r = rem >> 3, rem & 7
        i2 = idx - fc * 32
        bid = fc * 256 + (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + i2
    plane = bid // tpp
    rem = bid - plane * tpp
    tn0, w = 0, tiles_n
    if colblock > 0 and tiles_n > colblock:
        per_block = tiles_m * colblock
        b = rem // per_block
        rem -= b * per_block
        tn0 = b * colblock
        w = min(colblock, tiles_n - tn0)
    GM = 8
    g = rem // (GM * w)
    first_m = g * GM
    gm = min(GM, tiles_m - first_m)
    rem -= g * GM * w
    return plane, first_m + rem % gm, tn0 + rem // gm


@pytest.mark.parametrize("colblock", [0, 1, 2, 3, 8, 12, 20, 32, 64])
def test_every_tile_exactly_once(colblock):
    for tiles_m, tiles_n, planes in [(64, 64, 2), (32, 32, 3), (64, 70, 2), (13, 45, 3), (5, 33, 2), (3, 129, 1), (100, 9, 2), (1, 1, 5), (2, 5, 14)]:
        total = planes * tiles_m * tiles_n
        # the persistent loop visits vb = blockIdx + round * grid for a grid that is a multiple of 8 (or the whole range in one round)
        seen = {map_tile(vb, total, tiles_m, tiles_n, colblock) for vb in range(total)}
        assert len(seen) == total, (tiles_m, tiles_n, planes)
        assert all(0 <= p < planes and 0 <= tm < tiles_m and 0 <= tn < tiles_n for p, tm, tn in seen)


def test_column_blocks_keep_a_block_together():
    # 64 x 64 tiles, blocks of 32 columns: the first 64 * 32 tiles of a plane stay inside columns 0-31 (all row groups of block 0 first)
    tiles = [map_tile(vb, 64 * 64, 64, 64, 32) for vb in range(64 * 64)]
    first_half = {t for t in tiles if t[2] < 32}
    assert len(first_half) == 64 * 32
    # chunk c (256 consecutive canonical tiles) = 8 tile-rows x 32 tile-columns of ONE block
    canon = {}
    for vb in range(64 * 64):
        xcd, idx = vb & 7, vb >> 3
        canon[(idx >> 5) * 256 + xcd * 32 + (idx & 31)] = tiles[vb]
    for c in range(16):
        rows = {canon[c * 256 + i][1] for i in range(256)}
        cols = {canon[c * 256 + i][2] for i in range(256)}
        assert len(rows) == 8 and len(cols) == 32 and (max(cols) < 32) == (c < 8), (c, rows, cols)


# ---- round 4: the same map with its divisions on the scalar unit (oz2_gemm_common.hpp TileMapArgs / udivmod_magic / map_tile(bid, nwg, args))
def magic(d):
    return 0xFFFFFFFF if d <= 1 else (1 << 32) // d


def udivmod_magic(x, d, M):
    q = (x * M) >> 32                       # s_mul_hi_u32
    r = x - q * d
    if r >= d:
        q, r = q + 1, r - d
    return q, r


def test_magic_division_is_exact():
    import random
    rng = random.Random(5)
    ds = list(range(1, 600)) + [rng.randrange(1, 1 << 24) for _ in range(3000)] + [(1 << 31) - 1, 1 << 31, (1 << 32) - 1]
    for d in ds:
        M = magic(d)
        xs = [0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, (1 << 32) - 1, (1 << 31), (1 << 31) - 1] + [rng.randrange(0, 1 << 32) for _ in range(40)]
        xs += [k * d + e for k in (1, 7, 1000, ((1 << 32) - 1) // d) for e in (-1, 0, 1)]
        for x in xs:
            if 0 <= x < (1 << 32):
                assert udivmod_magic(x, d, M) == divmod(x, d), (x, d)


def map_tile_magic(bid, nwg, tiles_m, tiles_n, colblock):
    """mirror of the device function: make_tile_map on the host + map_tile(bid, nwg, TileMapArgs)"""
    cb = colblock if (colblock > 0 and tiles_n > colblock) else 0
    tpp = tiles_m * tiles_n
    wfull = cb if cb else tiles_n
    wtail = tiles_n % cb if cb else 0
    pb = tiles_m * wfull
    m_tpp, m_pb, m_full, m_tail = magic(tpp), magic(pb), magic(8 * wfull), magic(8 * (wtail if wtail else wfull))
    xcd, idx = bid & 7, bid >> 3
    fc = (nwg >> 3) >> 5
    if idx < fc * 32:
        bid = (idx >> 5) * 256 + xcd * 32 + (idx & 31)
    else:
        rem = nwg - fc * 256
        q, r = rem >> 3, rem & 7
        bid = fc * 256 + (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + (idx - fc * 32)
    plane, rem = udivmod_magic(bid, tpp, m_tpp)
    tn0, w, m_gs = 0, tiles_n, m_full
    if cb:
        b, rem = udivmod_magic(rem, pb, m_pb)
        tn0 = b * cb
        if tiles_n - tn0 < cb:
            w, m_gs = tiles_n - tn0, m_tail
        else:
            w = cb
    g, rem = udivmod_magic(rem, 8 * w, m_gs)
    first_m = g * 8
    if tiles_m - first_m >= 8:
        return plane, first_m + (rem & 7), tn0 + (rem >> 3)
    gm = tiles_m - first_m
    return plane, first_m + rem % gm, tn0 + rem // gm


@pytest.mark.parametrize("colblock", [0, 1, 2, 3, 8, 12, 20, 32, 64])
def test_scalar_unit_map_equals_the_division_form(colblock):
    for tiles_m, tiles_n, planes in [(64, 64, 2), (32, 32, 3), (64, 70, 2), (13, 45, 3), (5, 33, 2), (3, 129, 1), (100, 9, 2), (1, 1, 5), (2, 5, 14), (8, 8, 1), (9, 65, 2)]:
        total = planes * tiles_m * tiles_n
        for vb in range(total):
            assert map_tile_magic(vb, total, tiles_m, tiles_n, colblock) == map_tile(vb, total, tiles_m, tiles_n, colblock), (vb, tiles_m, tiles_n, planes, colblock)
