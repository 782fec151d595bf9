"""Size-independent properties at BASELINE.json's full sizes (1M Gaussians, 1920x1080), beside the full-size comparison
with the oracle in tests/test_configs_gpu.py::test_config_c3_full_size_against_oracle:
  * alpha channel: with features[:, 0] = 1 the blended buffer[0] equals 1 - final transmittance, and
    colour(bg = 1) - colour(bg = 0) equals that transmittance (compositing identity);
  * the backward is linear in the upstream gradients;
  * structural invariants of the private tile lists (sorted by tile, depth order inside a tile, ranges
    partition [0, R), quadrant lists = the order-preserving split of the tile list by the quadrant masks);
  * bitwise reproducibility;
  * `observe` and `radii` consistency (observe > 0 only where radii > 0)."""
import numpy as np
import pytest
import torch

import helpers as Hh

pytestmark = pytest.mark.gpu

P, W, H, FC = 1_000_000, 1920, 1080, 9


@pytest.fixture(scope="module")
def big():
    sc = Hh.make_scene(P, W, H, seed=0, fc=FC)
    return sc


def _forward(sc, bg):
    from diff_gaussian_rasterization import GaussianRasterizer
    sc = dict(sc); sc["bg"] = torch.tensor(bg, dtype=torch.float32)
    g = {k: v.cuda() for k, v in sc["g"].items()}
    m2 = torch.zeros(g["means3D"].shape[0], 4, device="cuda")
    with torch.no_grad():
        return GaussianRasterizer(Hh.settings_for(sc, "cuda"))(g["means3D"], m2, g["opacities"], shs=g["shs"],
                                                              scales=g["scales"], rotations=g["rotations"],
                                                              features=g["features"])


def test_compositing_identities(big):
    c0, radii, obs, b0 = _forward(big, (0.0, 0.0, 0.0))
    c1, _, _, b1 = _forward(big, (1.0, 1.0, 1.0))
    T = (c1 - c0)  # = final_T per channel
    assert torch.allclose(T[0], T[1], atol=2e-6) and torch.allclose(T[0], T[2], atol=2e-6)
    assert float(T.min()) >= -1e-6 and float(T.max()) <= 1 + 1e-6
    assert torch.allclose(b0[0] + T[0], torch.ones_like(T[0]), atol=2e-5), "sum of blend weights = 1 - T"
    assert torch.equal(b0, b1), "the G-buffer has no background term"
    assert torch.all(b0[FC:] == 0)
    assert int((radii > 0).sum()) > 0.7 * P
    assert bool(torch.all((obs > 0) <= (radii > 0))) and int(obs.sum()) > 0


def test_backward_linear_in_upstream_gradients(big):
    sc1 = dict(big); sc2 = dict(big); sc3 = dict(big)
    g = torch.Generator().manual_seed(7)
    Gc2, Gb2 = torch.randn(3, H, W, generator=g), torch.randn(10, H, W, generator=g)
    sc2["Gc"], sc2["Gb"] = Gc2, Gb2
    sc3["Gc"], sc3["Gb"] = 0.5 * big["Gc"] - 2.0 * Gc2, 0.5 * big["Gb"] - 2.0 * Gb2
    _, g1 = Hh.run_hip(sc1)
    _, g2 = Hh.run_hip(sc2)
    _, g3 = Hh.run_hip(sc3)
    # fp32 rounding is not linear: the error of an element scales with the magnitude of the terms that
    # were combined (a few ill-conditioned Gaussians out of 1M carry gradients ~1e6 with ~1e-3 relative
    # noise in the cov2D backward), so the check is elementwise against that scale.
    for k in ("means3D", "opacities", "shs", "scales", "rotations", "features"):
        a, b = g1[k].astype(np.float64), g2[k].astype(np.float64)
        lin = 0.5 * a - 2.0 * b
        scale = 0.5 * np.abs(a) + 2.0 * np.abs(b) + 1e-6 * (np.abs(a).mean() + np.abs(b).mean())
        err = np.abs(g3[k] - lin) / scale
        assert np.median(err) < 1e-6, (k, float(np.median(err)))
        assert (err > 1e-3).mean() < 1e-4, (k, float((err > 1e-3).mean()), float(err.max()))
    _, g1b = Hh.run_hip(sc1)
    for k in g1:
        assert np.array_equal(g1[k], g1b[k]), f"{k} not bitwise reproducible"


@pytest.mark.parametrize("reference_binning", [False, True])
def test_tile_list_invariants(big, reference_binning):
    """the per-tile (depth, id) order, the tile ranges and the quadrant lists of both binning modes (2.7 M / 3.8 M instances)"""
    import gs2m_native
    import diff_gaussian_rasterization as dgr
    gs2m_native.set_reference_binning(reference_binning)
    g = {k: v.cuda() for k, v in big["g"].items()}
    st = Hh.settings_for(big, "cuda")
    e = torch.Tensor([])
    R, color, radii, observe, buffer, geomB, binB, imgB = dgr._C.rasterize_gaussians(
        st.bg, g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, g["features"], st.viewmatrix,
        st.projmatrix, st.tanfovx, st.tanfovy, H, W, g["shs"], 3, st.campos, False, FC)
    torch.cuda.synchronize()
    gs2m_native.set_reference_binning(False)
    lay = gs2m_native.debug_layout(P, R, W, H)
    al = lambda t: (-t.data_ptr()) % 256
    view = lambda t, off, n, dt: t[al(t) + off: al(t) + off + n * np.dtype(dt).itemsize].cpu().numpy().view(dt)
    tk = view(binB, lay.tile_keys, R, np.uint32).astype(np.int64)
    pl = view(binB, lay.point_list, R, np.uint32) & np.uint32(0x0FFFFFFF)  # list-driven kernels: quadrant mask above the id
    dk = view(geomB, lay.depth_key, P, np.uint32).astype(np.int64)
    assert np.all(np.diff(tk) >= 0), "instances sorted by tile"
    key = (tk << 32) | dk[pl]
    assert np.all(np.diff(key) >= 0), "depth order inside every tile"
    same = np.diff(key) == 0
    assert np.all(np.diff(pl.astype(np.int64))[same] > 0), "ties keep Gaussian-id order"
    Tn = ((W + 15) // 16) * ((H + 15) // 16)
    rg = view(imgB, lay.ranges, 2 * Tn, np.uint32).reshape(Tn, 2).astype(np.int64)
    t = rg[:, 1] > rg[:, 0]
    o = np.argsort(rg[t, 0])
    assert rg[t, 0][o][0] == 0 and rg[t, 1][o][-1] == R and np.all(rg[t, 0][o][1:] == rg[t, 1][o][:-1])
    assert np.all(tk[rg[t, 0]] == np.nonzero(t)[0])
    nc = view(imgB, lay.n_contrib, W * H, np.uint32).reshape(H, W)
    lens = (rg[:, 1] - rg[:, 0]).reshape((H + 15) // 16, (W + 15) // 16).repeat(16, 0).repeat(16, 1)[:H, :W]
    assert np.all(nc <= lens)
    assert np.array_equal(radii.cpu().numpy() > 0, dk != 0xFFFFFFFF)
    # the four quadrant lists of every tile: the tile list's entries whose mask holds the quadrant, in list order, with the
    # position in the tile list; the gradient rows: dense over the view, one per set mask bit, a Gaussian's rows contiguous
    plm = view(binB, lay.point_list, R, np.uint32)
    ql = view(binB, lay.qlist, 8 * R, np.uint32).reshape(4 * R, 2)
    qr = view(binB, lay.qrow, 4 * R, np.uint32)
    qc = view(imgB, lay.qcount, 4 * Tn, np.uint32).reshape(Tn, 4)
    cnts = view(geomB, lay.counters, 64, np.uint32)
    total_rows, U = int(cnts[2]), int(cnts[3])  # the row space; heavy units (256 reserved rows each, in front of the dense rows)
    masks = plm >> 28
    set_bits = int(sum(((masks >> q) & 1).sum() for q in range(4)))
    rng = np.random.default_rng(3)
    for t in rng.choice(np.nonzero(t)[0], 40, replace=False):  # (`t` was the touched-tiles mask)
        lo, hi = int(rg[t, 0]), int(rg[t, 1])
        n = hi - lo
        for q in range(4):
            sel = np.nonzero((masks[lo:hi] >> q) & 1)[0]
            assert qc[t, q] == len(sel)
            ent = ql[4 * lo + q * n: 4 * lo + q * n + len(sel)]
            assert np.array_equal(ent[:, 0], plm[lo:hi][sel]) and np.array_equal(ent[:, 1], sel)
    rows_all = np.concatenate([qr[4 * int(rg[t, 0]) + q * int(rg[t, 1] - rg[t, 0]): 4 * int(rg[t, 0]) + q * int(rg[t, 1] - rg[t, 0]) + int(qc[t, q])]
                               for t in range(Tn) for q in range(4)])
    gids_all = np.concatenate([ql[4 * int(rg[t, 0]) + q * int(rg[t, 1] - rg[t, 0]): 4 * int(rg[t, 0]) + q * int(rg[t, 1] - rg[t, 0]) + int(qc[t, q]), 0]
                               for t in range(Tn) for q in range(4)]) & np.uint32(0x0FFFFFFF)
    gr_raw = view(geomB, lay.gauss_rows, P, np.uint32)
    tt = view(geomB, lay.tiles_touched, P, np.uint32).astype(np.int64)
    heavy = ((gr_raw & np.uint32(0x80000000)) != 0) & (tt > 0)
    ttw = np.concatenate([tt, np.zeros((-P) % 64, np.int64)]).reshape(-1, 64)
    crowded = np.repeat((ttw * (ttw < 40)).sum(1) > 320, 64)[:P]
    assert np.array_equal(heavy, (tt >= 40) | (crowded & (tt >= 8))), "heavy = at least 40 instances, 8 in a crowded wave (common.h)"
    assert len(rows_all) == set_bits and len(np.unique(rows_all)) == set_bits and int(rows_all.max()) < total_rows, "one row per set mask bit, all different"
    is_h = heavy[gids_all]
    lr, lg = rows_all[~is_h].astype(np.int64), gids_all[~is_h]
    assert total_rows == 256 * U + len(lr) and np.array_equal(np.sort(lr), np.arange(256 * U, total_rows)), "the waves' rows: dense behind the heavy units"
    o = np.argsort(lr)
    runs = 1 + int(np.count_nonzero(np.diff(lg[o].astype(np.int64))))
    assert runs == len(np.unique(lg)), "every Gaussian's rows are ONE contiguous run"
    cnt = np.bincount(lg, minlength=P)
    assert np.array_equal(cnt[~heavy], gr_raw[~heavy] * (tt[~heavy] > 0)), "gauss_rows = rows per Gaussian"
    if reference_binning:
        assert U > 0, "the reference's rectangles hold Gaussians of 40 tiles and more on this scene"
    if U > 0:
        u0 = (gr_raw & np.uint32(0x7FFFFFFF)).astype(np.int64)
        nu = (tt + 63) // 64
        hg = np.nonzero(heavy)[0]
        assert int(nu[hg].sum()) == U and np.array_equal(u0[hg], np.concatenate([[0], np.cumsum(nu[hg])[:-1]])), "units: whole, in index order"
        hr, hgid = rows_all[is_h].astype(np.int64), gids_all[is_h]
        assert np.all(hr >= 256 * u0[hgid]) and np.all(hr < 256 * (u0[hgid] + nu[hgid])), "a heavy Gaussian's rows lie in its own units"
        assert np.all((hr - 256 * u0[hgid]) // 4 < tt[hgid]), "four reserved rows per instance"
