"""csrc/tile_sort.hip on caller-made spans (gs2m_debug_tile_sort): every tile's span must come out in (depth, Gaussian id)
order -- the order the reference's 45-bit radix sort of id-ordered keys produces inside a tile (rasterizer_impl.cu:288-296) --
and the four quadrant lists / gradient rows must be the order-preserving split of it.  Spans of every length class (one wave
with 8 or 16 elements per lane, a workgroup with 2, 4 or 8 per lane, a workgroup over LDS, a workgroup over global memory), with many exactly equal depths."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[0, 1, 2, 3], ids=["auto", "workgroups", "waves", "waves-mid-by-workgroup"], autouse=True)
def policy(request):
    """every arrangement of who sorts which span (gs2m_set_tile_sort_policy): by tile count; a workgroup per tile; a wave per tile;
    a wave per tile with the spans of 513 .. 1024 entries left to the workgroup kernel (a frame of short spans on average)"""
    import gs2m_native
    gs2m_native.set_tile_sort_policy(request.param)
    yield request.param
    gs2m_native.set_tile_sort_policy(0)


def _run(lengths, max_tile, seed, tie_levels):
    """spans as the stable tile sort leaves them: per tile the emission slots of its instances in Gaussian-index order"""
    import gs2m_native
    rng = np.random.default_rng(seed)
    tiles = len(lengths)
    starts = np.concatenate([[0], np.cumsum(lengths)]).astype(np.uint32)
    n = int(starts[-1])
    raw = np.zeros((tiles, 2), np.uint32)
    for t, L in enumerate(lengths):
        if L:
            raw[t] = (~np.uint32(starts[t]), starts[t + 1])
    P = 1 << 20
    nn = max(n, 1)
    depth_key = rng.integers(0x40000000, 0x41000000, P).astype(np.uint32)
    if tie_levels:
        depth_key = (np.float32(2.0) + rng.integers(0, tie_levels, P).astype(np.float32) * np.float32(0.001)).view(np.uint32)
    val = np.zeros(nn, np.uint32); row = np.zeros(nn, np.uint32)
    for t, L in enumerate(lengths):
        if L == 0:
            continue
        lo = int(starts[t])
        gid = np.sort(rng.choice(P, L, replace=False)).astype(np.uint32)   # a Gaussian appears once per tile; index order
        val[lo:lo + L] = gid | (rng.integers(0, 16, L).astype(np.uint32) << np.uint32(28))
        row[lo:lo + L] = rng.integers(0, 1 << 20, L)
    slot = rng.permutation(nn).astype(np.uint32)          # the emission slots the sorted values point at
    e_rec = np.zeros((nn, 4), np.uint32)
    e_rec[slot, 0] = val; e_rec[slot, 1] = row; e_rec[slot, 2] = depth_key[val & np.uint32(0x0FFFFFFF)]
    wave_rowbase = rng.integers(0, 1 << 24, P // 64).astype(np.uint32)
    dev = "cuda"
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a).astype(np.uint32).view(np.int32).reshape(-1).copy()).to(dev)
    t_raw, t_slot, t_erec, t_wrb = T(raw), T(slot), T(e_rec), T(wave_rowbase)
    Z = lambda k, fill=0: torch.full((k,), fill, dtype=torch.int32, device=dev)
    o_rg, o_pl, o_tmp, o_ql, o_qr, o_qc = Z(2 * tiles, -1), Z(nn), Z(nn), Z(8 * nn), Z(4 * nn), Z(4 * tiles, -1)
    L = gs2m_native.lib()
    rc = L.gs2m_debug_tile_sort(tiles, t_raw.data_ptr(), o_rg.data_ptr(), t_slot.data_ptr(), t_erec.data_ptr(), t_wrb.data_ptr(),
                                o_pl.data_ptr(), o_tmp.data_ptr(), o_ql.data_ptr(), o_qr.data_ptr(), o_qc.data_ptr(),
                                gs2m_native.stream_ptr())
    gs2m_native.check(rc, "gs2m_debug_tile_sort")
    torch.cuda.synchronize()
    U = lambda t: t.cpu().numpy().view(np.uint32)
    rg, pl, ql, qr, qc = U(o_rg).reshape(tiles, 2), U(o_pl), U(o_ql).reshape(-1, 2), U(o_qr), U(o_qc).reshape(tiles, 4)
    for t, Ln in enumerate(lengths):
        lo = int(starts[t])
        if Ln == 0:
            assert np.all(qc[t] == 0) and np.all(rg[t] == 0), t
            continue
        assert rg[t, 0] == lo and rg[t, 1] == lo + Ln, t
        v, r = val[lo:lo + Ln], row[lo:lo + Ln]
        d = depth_key[v & np.uint32(0x0FFFFFFF)]
        order = np.lexsort((v & np.uint32(0x0FFFFFFF), d))   # by depth, ties by Gaussian id
        assert np.array_equal(pl[lo:lo + Ln], v[order]), f"tile {t} (length {Ln}): sorted values"
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
    _run([0, 1, 2, 3, 63, 64, 65, 127, 128, 129, 130, 255, 256, 257, 300, 331, 427, 511, 512, 5, 0, 400], 512, 1 + tie_levels, tie_levels)


@pytest.mark.parametrize("tie_levels", [0, 7, 300])
def test_one_wave_per_tile_up_to_1024(tie_levels):
    _run([513, 1, 0, 700, 1023, 1024, 64, 900, 512, 600, 33, 1025, 1500, 2047, 2048], 2048, 11 + tie_levels, tie_levels)


@pytest.mark.parametrize("tie_levels", [0, 5, 2000])
def test_workgroup_per_tile_lds_and_global(tie_levels):
    _run([1025, 10, 2047, 2048, 2049, 4095, 4096, 4097, 0, 9000, 513, 1024, 300, 20000], 20000, 21 + tie_levels, tie_levels)


def test_many_tiles_like_a_frame():
    rng = np.random.default_rng(5)
    _run(list(rng.integers(250, 430, 3000)), 430, 31, 0)
    _run(list(rng.integers(0, 900, 500)), 900, 32, 40)
    _run(list(rng.integers(200, 700, 4200)), 700, 33, 25)   # a big frame: one wave per tile
