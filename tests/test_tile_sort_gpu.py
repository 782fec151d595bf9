"""csrc/tile_sort.hip on caller-made spans (gs2m_debug_tile_sort): every tile's span must come out in (depth, Gaussian id)
order -- the order the reference's 45-bit radix sort of id-ordered keys produces inside a tile (rasterizer_impl.cu:288-296) --
and the four quadrant lists / gradient rows must be the order-preserving split of it.  Spans of every length class (one wave
with 8 or 16 elements per lane, a workgroup over LDS, a workgroup over global memory), with many exactly equal depths."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(lengths, max_tile, seed, tie_levels):
    import gs2m_native
    rng = np.random.default_rng(seed)
    tiles = len(lengths)
    starts = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint32)
    n = int(starts[-1])
    ranges = np.stack([starts[:-1], starts[1:]], 1).astype(np.uint32)
    ranges[np.asarray(lengths) == 0] = 0
    P = 1 << 20
    depth = np.empty(max(n, 1), np.uint32); val = np.empty(max(n, 1), np.uint32); row = np.empty(max(n, 1), np.uint32)
    for t, L in enumerate(lengths):
        if L == 0:
            continue
        lo = int(starts[t])
        d = rng.integers(0, tie_levels, L) if tie_levels else rng.integers(0, 1 << 31, L)
        depth[lo:lo + L] = (np.float32(2.0) + d.astype(np.float32) * np.float32(0.001)).view(np.uint32) if tie_levels else (d.astype(np.uint32) | np.uint32(0x40000000))
        gid = rng.choice(P, L, replace=False).astype(np.uint32)          # a Gaussian appears once per tile
        val[lo:lo + L] = gid | (rng.integers(0, 16, L).astype(np.uint32) << np.uint32(28))
        row[lo:lo + L] = rng.integers(0, 1 << 20, L)
    wave_rowbase = rng.integers(0, 1 << 24, P // 64).astype(np.uint32)
    dev = "cuda"
    T = lambda a: torch.from_numpy(a.astype(np.uint32).view(np.int32).copy()).to(dev)
    t_ranges, t_depth, t_val, t_row, t_wrb = T(ranges.reshape(-1)), T(depth), T(val), T(row), T(wave_rowbase)
    nn = max(n, 1)
    o_pl, o_tk = torch.zeros(nn, dtype=torch.int32, device=dev), torch.zeros(nn, dtype=torch.int32, device=dev)
    o_ql = torch.zeros(8 * nn, dtype=torch.int32, device=dev)
    o_qr = torch.zeros(4 * nn, dtype=torch.int32, device=dev)
    o_qc = torch.full((4 * tiles,), -1, dtype=torch.int32, device=dev)
    L = gs2m_native.lib()
    rc = L.gs2m_debug_tile_sort(tiles, int(max_tile), t_ranges.data_ptr(), t_depth.data_ptr(), t_val.data_ptr(), t_row.data_ptr(), t_wrb.data_ptr(),
                                o_pl.data_ptr(), o_tk.data_ptr(), o_ql.data_ptr(), o_qr.data_ptr(), o_qc.data_ptr(), gs2m_native.stream_ptr())
    gs2m_native.check(rc, "gs2m_debug_tile_sort")
    torch.cuda.synchronize()
    U = lambda t: t.cpu().numpy().view(np.uint32)
    pl, tk, ql, qr, qc = U(o_pl), U(o_tk), U(o_ql).reshape(-1, 2), U(o_qr), U(o_qc).reshape(tiles, 4)
    for t, Ln in enumerate(lengths):
        lo = int(starts[t])
        if Ln == 0:
            assert np.all(qc[t] == 0), t
            continue
        d, v, r = depth[lo:lo + Ln], val[lo:lo + Ln], row[lo:lo + Ln]
        order = np.lexsort((v & np.uint32(0x0FFFFFFF), d))   # by depth, ties by Gaussian id
        assert np.array_equal(pl[lo:lo + Ln], v[order]), f"tile {t} (length {Ln}): sorted values"
        assert np.all(tk[lo:lo + Ln] == t)
        sv, sr = v[order], r[order] + wave_rowbase[(v[order] & np.uint32(0x0FFFFFFF)) >> 6]
        m = sv >> 28
        for q in range(4):
            sel = np.nonzero((m >> q) & 1)[0]
            assert qc[t, q] == len(sel), (t, q)
            base = 4 * lo + q * Ln
            assert np.array_equal(ql[base:base + len(sel), 0], sv[sel]) and np.array_equal(ql[base:base + len(sel), 1], sel.astype(np.uint32)), (t, q)
            below = np.array([bin(int(x) & ((1 << q) - 1)).count("1") for x in m[sel]], dtype=np.uint32)
            assert np.array_equal(qr[base:base + len(sel)], (sr[sel] + below).astype(np.uint32)), (t, q)


@pytest.mark.parametrize("tie_levels", [0, 7, 300])
def test_one_wave_per_tile_up_to_512(tie_levels):
    _run([0, 1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 331, 427, 511, 512, 5, 0, 400], 512, 1 + tie_levels, tie_levels)


@pytest.mark.parametrize("tie_levels", [0, 7, 300])
def test_one_wave_per_tile_up_to_1024(tie_levels):
    _run([513, 1, 0, 700, 1023, 1024, 64, 900, 512, 600, 33], 1024, 11 + tie_levels, tie_levels)


@pytest.mark.parametrize("tie_levels", [0, 5, 2000])
def test_workgroup_per_tile_lds_and_global(tie_levels):
    _run([1025, 10, 2047, 2048, 2049, 4095, 4096, 4097, 0, 9000, 513, 1024, 300, 20000], 20000, 21 + tie_levels, tie_levels)


def test_many_tiles_like_a_frame():
    rng = np.random.default_rng(5)
    _run(list(rng.integers(250, 430, 3000)), 430, 31, 0)
    _run(list(rng.integers(0, 900, 500)), 900, 32, 40)
