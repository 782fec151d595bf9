"""The rasterizer on a TRAINED model of the C4 substitute (gs2m_train.c4_run): what one training view of the final model costs per
stage and how its work is distributed (instances per Gaussian, list length per tile, gradient rows per Gaussian) -- the workload the
uniform bench cloud does not show.  Run on the GPU box, two steps so that the second can sit under rocprofv3 by itself:
    python tools/c4_profile.py train /tmp/c4_call.pt [iterations]      the training run; saves what render() hands the op for view 0
    python tools/c4_profile.py run /tmp/c4_call.pt [steps]             distribution + stage times of forward + backward on that call
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch


def train(path, iters):
    import tempfile
    import gs2m_train
    from test_c4_gpu import _capture_rasterizer_call
    with tempfile.TemporaryDirectory() as tmp:
        scene = gs2m_train.c4_scene(os.path.join(tmp, "c4"))
        model, st = gs2m_train.c4_run(None, iterations=iters, scene=scene)
    print("trained:", {k: st[k] for k in ("it_per_s", "points_start", "points_max", "points_end", "psnr_end")}, flush=True)
    calls = [_capture_rasterizer_call(scene[0][v], model, geometry_stage=True) for v in (0, 17, 33)]
    torch.save(calls, path)
    # the model's geometry alone (what decides the work distribution), small enough to come back from the GPU box: later runs rebuild
    # the calls from it with random colour / feature data (`run` on an .npz)
    g = calls[0]["g"]
    out = dict(means3D=g["means3D"].numpy(), scales=g["scales"].numpy(), rotations=g["rotations"].numpy(), opacities=g["opacities"].numpy(),
               sh_dc=g["shs"][:, 0].numpy().astype(np.float16), W=calls[0]["W"], H=calls[0]["H"], fc=calls[0]["fc"], sh_degree=calls[0]["sh_degree"])
    for k, c in enumerate(calls):
        for name in ("viewmatrix", "projmatrix", "campos"):
            out[f"cam{k}_{name}"] = c["cam"][name].numpy()
        out[f"cam{k}_tan"] = np.array([c["cam"]["tanfovx"], c["cam"]["tanfovy"]])
    np.savez(os.path.join(ROOT, "gpurun_out", "c4_geom.npz"), **out)


def calls_from_geometry(path):
    import gs2m_synth as S
    z = np.load(path)
    P = z["means3D"].shape[0]
    gen = torch.Generator().manual_seed(0)
    shs = torch.cat([torch.from_numpy(z["sh_dc"].astype(np.float32)).reshape(P, 1, 3), 0.1 * torch.randn(P, 15, 3, generator=gen)], dim=1).contiguous()
    nrm = torch.nn.functional.normalize(torch.randn(P, 3, generator=gen), dim=1)
    feats = torch.cat([torch.ones(P, 1), 1.0 + 9.0 * torch.rand(P, 1, generator=gen), nrm, torch.rand(P, 5, generator=gen)], dim=1).contiguous()
    g = dict(means3D=torch.from_numpy(z["means3D"]), scales=torch.from_numpy(z["scales"]), rotations=torch.from_numpy(z["rotations"]),
             opacities=torch.from_numpy(z["opacities"]), shs=shs, features=feats)
    W, H = int(z["W"]), int(z["H"])
    calls = []
    for k in range(3):
        Gc, Gb = S.make_upstream_grads(H, W, seed=5)
        cam = dict(viewmatrix=torch.from_numpy(z[f"cam{k}_viewmatrix"]), projmatrix=torch.from_numpy(z[f"cam{k}_projmatrix"]),
                   campos=torch.from_numpy(z[f"cam{k}_campos"]), tanfovx=float(z[f"cam{k}_tan"][0]), tanfovy=float(z[f"cam{k}_tan"][1]))
        calls.append(dict(cam=cam, g=g, Gc=Gc, Gb=Gb, W=W, H=H, fc=int(z["fc"]), sh_degree=int(z["sh_degree"]), bg=torch.zeros(3)))
    return calls


def run(path, steps):
    import gs2m_native
    import helpers as Hh
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import GaussianRasterizer
    dev = "cuda"
    calls = calls_from_geometry(path) if path.endswith(".npz") else torch.load(path)
    for ci, sc in enumerate(calls):
        g = {k: v.to(dev).requires_grad_(True) for k, v in sc["g"].items()}
        P, W, H, fc = g["means3D"].shape[0], sc["W"], sc["H"], sc["fc"]
        st = Hh.settings_for(sc, dev)
        Gc, Gb = sc["Gc"].to(dev), sc["Gb"].to(dev)
        if ci == 0:
            e = torch.Tensor([])
            with torch.no_grad():
                R, color, radii, observe, buffer, geomB, binB, imgB = dgr._C.rasterize_gaussians(
                    st.bg, g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, g["features"], st.viewmatrix,
                    st.projmatrix, st.tanfovx, st.tanfovy, H, W, g["shs"], sc["sh_degree"], st.campos, False, fc)
            torch.cuda.synchronize()
            lay = gs2m_native.debug_layout(P, R, W, H)
            al = lambda t: (-t.data_ptr()) % 256
            view = lambda t, off, n, dt: t[al(t) + off: al(t) + off + n * np.dtype(dt).itemsize].cpu().numpy().view(dt)
            tt = view(geomB, lay.tiles_touched, P, np.uint32).astype(np.int64)
            rows = view(geomB, lay.gauss_rows, P, np.uint32).astype(np.int64)
            rows[tt == 0] = 0  # (written for emitting Gaussians only)
            heavy = (rows & 0x80000000) != 0  # (heavy Gaussians: the entry is their first unit, common.h)
            rows[heavy] = 0
            Tn = ((W + 15) // 16) * ((H + 15) // 16)
            rg = view(imgB, lay.ranges, Tn * 2, np.uint32).reshape(Tn, 2).astype(np.int64)
            ll = rg[:, 1] - rg[:, 0]
            q = [50, 90, 99, 99.9, 100]
            print(f"view 0: {P} Gaussians {W}x{H} fc {fc}: R {R}, visible {int((radii > 0).sum())}, emitting {int((tt > 0).sum())}, tiles {Tn}")
            print("  instances / Gaussian: mean %.2f percentiles %s %s; Gaussians with >= 64 / 512 tiles: %d / %d holding %.1f %% / %.1f %% of R" % (
                tt[tt > 0].mean(), q, np.percentile(tt[tt > 0], q).tolist(), int((tt >= 64).sum()), int((tt >= 512).sum()),
                100.0 * tt[tt >= 64].sum() / max(R, 1), 100.0 * tt[tt >= 512].sum() / max(R, 1)))
            w = tt[: (P // 64) * 64].reshape(-1, 64)
            print("  per wave of 64 Gaussians (index order): mean of max %.1f, mean of sum %.1f, max of sum %d" % (w.max(1).mean(), w.sum(1).mean(), w.sum(1).max()))
            print("  tile list length: mean %.1f percentiles %s %s; tiles <= 512: %d, 513..1024: %d, > 1024: %d, > 4096: %d" % (
                ll.mean(), q, np.percentile(ll, q).tolist(), int((ll <= 512).sum()), int(((ll > 512) & (ll <= 1024)).sum()), int((ll > 1024).sum()), int((ll > 4096).sum())))
            print("  gradient rows of the waves' own (not heavy) Gaussians: total %d = %.2f per instance of the frame; per Gaussian mean %.2f percentiles %s %s; Gaussians with >= 64 / 256 / 1024 rows: %d / %d / %d holding %.1f / %.1f / %.1f %% of the rows" % (
                rows.sum(), rows.sum() / max(R, 1), rows[rows > 0].mean(), q, np.percentile(rows[rows > 0], q).tolist(),
                int((rows >= 64).sum()), int((rows >= 256).sum()), int((rows >= 1024).sum()),
                100.0 * rows[rows >= 64].sum() / rows.sum(), 100.0 * rows[rows >= 256].sum() / rows.sum(), 100.0 * rows[rows >= 1024].sum() / rows.sum()))
            wr = rows[: (P // 64) * 64].reshape(-1, 64)
            print("  rows per wave of 64 Gaussians: mean of max %.1f, mean of sum %.1f, max of sum %d" % (wr.max(1).mean(), wr.sum(1).mean(), wr.sum(1).max()))
            del geomB, binB, imgB
        means2D = torch.zeros(P, 4, device=dev, requires_grad=True)
        rast = GaussianRasterizer(st)

        def step():
            for t in list(g.values()) + [means2D]:
                t.grad = None
            color, radii, observe, buffer = rast(g["means3D"], means2D, g["opacities"], features=g["features"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
            ((color * Gc).sum() + (buffer * Gb).sum()).backward()

        for _ in range(10):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        gs2m_native.profile_mode(2)
        for _ in range(10):
            step()
        torch.cuda.synchronize()
        stages = gs2m_native.profile_collect()
        gs2m_native.profile_mode(0)
        print(f"call {ci}: P {P}: {ms:.4f} ms per forward + backward (incl. the two torch sums);  stages us:",
              {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in stages.items() if v[1]}, flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "train":
        train(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 5000)
    else:
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 50)
