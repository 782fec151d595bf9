"""Report: the reference build (oracle/_ref/libgs2m_ref.so: the reference's kernels through hipify-perl, on this GPU) against
the CPU oracle and against the HIP path, array by array.  Test infrastructure, not product."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import helpers as Hh
from oracle import oracle, reference

oracle.build()


def cmp(name, a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape:
        print(f"  {name:16s} SHAPE {a.shape} vs {b.shape}"); return
    if a.dtype.kind in "iub":
        print(f"  {name:16s} exact={np.array_equal(a, b)} differing={int((a != b).sum())} of {a.size}")
    else:
        same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        print(f"  {name:16s} bit-identical={same} max abs {d.max() if d.size else 0:.3e} rel-to-max {d.max() / max(np.abs(b).max(), 1e-30) if d.size else 0:.3e} differing {int((a.view(np.uint32) != b.view(np.uint32)).sum())} of {a.size}")


for (P, W, H, seed, fc, hi) in () if "--cubemap" in sys.argv or "--c3-only" in sys.argv else ((3000, 160, 96, 1, 9, 0.06), (20000, 320, 200, 2, 5, 0.05), (50000, 640, 360, 3, 10, 0.03)):
    sc = Hh.make_scene(P, W, H, seed=seed, fc=fc, scale_hi=hi)
    f, gr = Hh.run_oracle(oracle, sc)
    r, rg = Hh.run_oracle(reference, sc)
    print(f"scene P={P} {W}x{H} fc={fc}: num_rendered oracle {f.num_rendered} reference {r.num_rendered}")
    for k in ("radii", "tiles_touched", "point_offsets", "clamped", "keys_sorted", "vals_sorted", "ranges", "n_contrib", "observe"):
        cmp(k, getattr(r, k), getattr(f, k))
    vis = f.radii > 0
    for k in ("depths", "means2D", "cov3D", "conic_opacity", "rgb"):
        cmp(k + "[vis]", getattr(r, k)[vis], getattr(f, k)[vis])
    for k in ("color", "buffer", "final_T"):
        cmp(k, getattr(r, k), getattr(f, k))
    for k in gr:
        cmp("d" + k, rg[k], gr[k])

if "--cubemap" in sys.argv:
    import render_utils as RU
    for res, rough in ((16, 1.0), (32, 0.5), (64, 0.385), (128, 0.27), (256, 0.155), (512, 0.04)):
        g = torch.Generator().manual_seed(res)
        x = torch.rand(6, res, res, 3, generator=g).cuda()
        G4 = torch.randn(6, res, res, 4, generator=g).cuda()
        cut = RU.ndf_cutoff(rough, 0.99)
        b = reference.specular_bounds(res, cut)
        t0 = time.perf_counter(); r4 = reference.specular_cubemap_fwd(x, b, rough, cut); t1 = time.perf_counter()
        rg = reference.specular_cubemap_bwd(x, b, G4, rough, cut); t2 = time.perf_counter()
        xx = x.clone().requires_grad_(True)
        o4 = RU._specular_cubemap.apply(xx, rough, cut)
        (o4 * G4).sum().backward()
        print(f"specular res {res} roughness {rough}: reference fwd {1e3 * (t1 - t0):.2f} ms bwd {1e3 * (t2 - t1):.2f} ms")
        cmp("raw out", o4.detach().cpu().numpy(), r4.cpu().numpy())
        cmp("weight sum", o4.detach()[..., 3].cpu().numpy(), r4[..., 3].cpu().numpy())
        cmp("ratio", (o4.detach()[..., :3] / o4.detach()[..., 3:]).cpu().numpy(), (r4[..., :3] / r4[..., 3:]).cpu().numpy())
        cmp("grad", xx.grad.cpu().numpy(), rg.cpu().numpy())
        if res >= 128:  # ill-conditioned lobes: both against the fp64 restatement on sampled output texels
            from oracle import cubemap_oracle as O
            T = 6 * res * res
            rows = np.unique(torch.randint(0, T, (24,), generator=g).numpy())
            xs = x.cpu().double().numpy().reshape(-1, 3)
            eo, er = 0.0, 0.0
            for k in range(0, len(rows), 2):
                r = rows[k:k + 2]
                W = O.specular_matrix(res, rough, cut, rows=r)
                exact = np.concatenate([W @ xs, W.sum(1, keepdims=True)], axis=1)
                eo = max(eo, float(np.abs(o4.detach().cpu().double().numpy().reshape(-1, 4)[r] - exact).max() / np.abs(exact).max()))
                er = max(er, float(np.abs(r4.cpu().double().numpy().reshape(-1, 4)[r] - exact).max() / np.abs(exact).max()))
            print(f"  against fp64 on {len(rows)} texels: HIP path {eo:.3e}, reference build {er:.3e} (relative to the largest value)")
    for res in (8, 16, 32):
        g = torch.Generator().manual_seed(res)
        x = torch.rand(6, res, res, 3, generator=g).cuda(); G = torch.randn(6, res, res, 3, generator=g).cuda()
        xx = x.clone().requires_grad_(True)
        o = RU.diffuse_cubemap(xx); (o * G).sum().backward()
        print(f"diffuse res {res}")
        cmp("out", o.detach().cpu().numpy(), reference.diffuse_cubemap_fwd(x).cpu().numpy())
        cmp("grad", xx.grad.cpu().numpy(), reference.diffuse_cubemap_bwd(x, G).cpu().numpy())
    sys.exit(0)

if "--full" in sys.argv:
    # the bench workloads (bench.py CONFIGS) through the reference build on this GPU: what a hipify port of the reference
    # delivers on MI355X, to read next to bench.py's ms_per_step for the same workloads (profiles/r03_bench*.json)
    cfgs = (("c3", (1_000_000, 1920, 1080, 9)), ("c2", (500_000, 1920, 1080, 5)), ("c5 shape", (2_000_000, 1920, 1080, 9)))
    for name, (P, W, H, fc) in cfgs[:1] if "--c3-only" in sys.argv else cfgs:
        sc = Hh.make_scene(P, W, H, seed=0, fc=fc)
        r, _ = Hh.run_oracle(reference, sc, backward=False)
        ms_ref = reference.timed_forward_backward(r, sc["Gc"].numpy(), sc["Gb"].numpy(), n=10)
        line = f"{name}: {P} Gaussians {W}x{H} fc {fc}: num_rendered {r.num_rendered}; reference build (hipify-perl, gfx950, -ffp-contract=off) {ms_ref:.2f} ms per view forward + backward"
        del r
        fast = os.path.join(ROOT, "oracle", "_ref", "libgs2m_ref_fast.so")
        if os.path.exists(fast):  # the same sources with hipcc's default contraction (`make -C oracle/ref_build CONTRACT=fast NAME=libgs2m_ref_fast.so`)
            std = reference.LIB_PATH
            reference.use_library(fast)
            r, _ = Hh.run_oracle(reference, sc, backward=False)
            line += f"; with hipcc's default contraction {reference.timed_forward_backward(r, sc['Gc'].numpy(), sc['Gb'].numpy(), n=10):.2f} ms"
            del r
            reference.use_library(std)
        print(line)
