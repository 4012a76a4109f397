"""Host-side invariants of the scan's launch plan (csrc/atlas_hip.hip make_plan), through a hook of the tuning build (no GPU needed:
the plan is host arithmetic). The kernel trusts the plan for its 32-bit byte offsets, its 26-bit virtual candidate rows and the
one-descriptor pool; sizes that cannot be tested on a GPU in this environment (up to what fits 288 GB of HBM, and beyond) are
covered here."""
import ctypes

import numpy as np
import pytest

from atlas_amd import _lib

ROWB = 768 * 2


@pytest.fixture(scope="module")
def plan():
    T = _lib.lib(tuning=True)
    T.atlas_tune_scan_plan.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
    T.atlas_tune_scan_plan.restype = None
    T.atlas_tune_set_scan_pool.argtypes, T.atlas_tune_set_scan_pool.restype = [ctypes.c_int, ctypes.c_int], None

    def f(N, k=40, cus=256):
        out = (ctypes.c_int64 * 8)()
        T.atlas_tune_scan_plan(N, k, cus, out)
        return dict(zip(("G", "rows_per_wg", "pool_begin", "pool_rows", "pool_tiles", "tile", "pool_tile", "supported"), [int(x) for x in out]))

    yield f, T
    T.atlas_tune_set_scan_pool(60, 32)


def _check(N, p):
    G, R, tile = p["G"], p["rows_per_wg"], p["tile"]
    assert 1 <= G <= 1024 and R % 16 == 0 and tile == 256 and p["pool_tile"] == 240
    if p["pool_tiles"] == 0:
        # static split: the ranges cover [0, N) and nothing else
        assert p["pool_rows"] == 0 and G * R >= N and G * R - N < G * 16 + 16    # (trailing workgroups may be left without rows: ntiles == 0)
    else:
        assert R % tile == 0 and R >= tile                          # whole static tiles, at least one (the sample)
        assert p["pool_begin"] == G * R and p["pool_begin"] + p["pool_rows"] == N and p["pool_rows"] > 0
        assert p["pool_tiles"] == -(-p["pool_rows"] // p["pool_tile"])
        assert p["pool_tiles"] >= G                                   # every workgroup's pre-assigned first pool tile exists
    if p["supported"]:
        assert (R + 2 * tile) * ROWB < 0xfff00000                    # 32-bit byte offsets inside a static range
        assert p["pool_rows"] * ROWB < 0xfff00000                    # one descriptor spans the pool
        assert R + 2 * tile + p["pool_rows"] < (1 << 26)             # 26-bit virtual candidate rows


def test_plan_invariants_over_sizes(plan):
    f, T = plan
    rng = np.random.default_rng(7)
    sizes = [0, 1, 15, 16, 17, 255, 256, 4097, 65535, 65536, 524287, 524288, 524289, 1_000_000, 4_000_000, 32_000_000,
             100_000_000, 187_000_000, 715_000_000, 2**31 - 1, 2**32 - 2]
    sizes += [int(x) for x in rng.integers(1, 200_000_000, size=300)] + [int(x) for x in rng.integers(1, 3_000_000, size=300)]
    for permille, cap in ((60, 16), (0, 16), (500, 64), (950, 255), (1000, 4096)):
        T.atlas_tune_set_scan_pool(permille, cap)
        for N in sizes:
            for cus in (256, 304, 64):
                _check(N, f(N, cus=cus))
    T.atlas_tune_set_scan_pool(60, 32)
    # what fits one MI355X (288 GB: 187M rows) is supported with the product setting, pooled, and the pool stays small
    for N in (1_000_000, 32_000_000, 100_000_000, 187_000_000):
        p = f(N)
        assert p["supported"] == 1 and p["pool_tiles"] > 0 and p["pool_rows"] <= 33 * 256 * 256 + 65536
    assert f(100_000)["pool_tiles"] == 0 and f(524_287)["pool_tiles"] == 0 and f(524_288)["pool_tiles"] > 0


def test_dma_staged_scan_tiles_cover_every_row_exactly_once(plan):
    """csrc/dscan_kernel.h walks the SAME plan with its own tile shapes: the static part of the slab, [0, pool_begin), in tiles of 256 rows DEALT to the
    workgroups (workgroup g: tiles g, g + G, ...; the last tile of the part may be partial), then the pool in tiles of `pool_tile_rows` rows (256 in the
    product; 128 / 64 through the tuning knob) -- pool tile g pre-assigned to workgroup g, tiles G + t by ticket t, a ticket past the last tile ends a
    workgroup's sequence (atlas_hip.hip sets ScanParams::pool_tiles = ceil(pool_rows / pool_tile_rows) for it). Simulated here over the plan hook: every
    slab row lies in exactly one tile; every virtual candidate row (26 bits: 256 c + r for row r of a workgroup's c-th static tile, pool rows from
    rows_per_wg on) maps back to its slab row the way the kernel's global_row() does; with a pool every workgroup has the same number of static tiles,
    rows_per_wg / 256 of them, so static virtual rows stay below the pool's."""
    f, T = plan
    rng = np.random.default_rng(11)
    sizes = [65536, 65537, 70_001, 524_287, 524_288, 524_527, 700_001, 1_000_000, 4_000_000, 32_000_000, 128_000_000] + [int(x) for x in rng.integers(65536, 40_000_000, size=60)]
    for N in sizes:
        p = f(N)
        G, R = p["G"], p["rows_per_wg"]
        if not (p["supported"] and 64 <= G <= 256):
            continue                                              # (not a coop shape: scan_kernel.h takes it)
        static_end = p["pool_begin"]
        assert static_end == (G * R if p["pool_tiles"] > 0 else N)
        n_static = -(-static_end // 256)
        vpool = R if p["pool_tiles"] > 0 else 1 << 26
        for ptr in (256, 128, 64):
            covered = np.zeros(N, dtype=np.uint8)
            for g in range(G):
                ntiles = (n_static - g + G - 1) // G if n_static > g else 0
                if p["pool_tiles"] > 0:
                    assert ntiles == R // 256 >= 1, (N, g)
                for c in range(ntiles):
                    r0 = (c * G + g) * 256
                    rem = min(256, static_end - r0)
                    assert rem > 0
                    for r in (0, rem - 1):                        # global_row() of the tile's first and last virtual row
                        v = c * 256 + r
                        assert v < vpool and v < (1 << 26) and (((v >> 8) * G + g) << 8 | (v & 255)) == r0 + r
                    covered[r0: r0 + rem] += 1
            pool_tiles = -(-p["pool_rows"] // ptr) if p["pool_rows"] > 0 else 0
            if p["pool_tiles"] > 0:
                assert pool_tiles >= G
            for pt in range(pool_tiles):                          # (whoever draws it: pre-assigned or by ticket, each tile index is handed out once)
                rem = min(ptr, p["pool_rows"] - pt * ptr)
                r0 = p["pool_begin"] + pt * ptr
                vrow0 = R + pt * ptr
                assert rem > 0 and vrow0 >= vpool and vrow0 + rem < (1 << 26) and vrow0 + (p["pool_begin"] - vpool) == r0
                covered[r0: r0 + rem] += 1
            assert covered.min() == 1 and covered.max() == 1, (N, ptr, int(np.argmin(covered)), int(np.argmax(covered)))
