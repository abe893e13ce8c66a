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
