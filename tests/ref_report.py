"""Reports on the reference build (oracle/_ref/libgs2m_ref.so: the reference's kernels through hipify-perl, on this GPU).
Test infrastructure, not product.

    python tests/ref_report.py                     three scenes, every array of the reference build against the CPU oracle's
    python tests/ref_report.py --full [--c3-only]  its time per view at the bench workloads
    python tests/ref_report.py --cubemap           the cube-map prefilters level by level against the HIP operators
    python tests/ref_report.py --sweep N [START] [--precomputed]   N random scenes, HIP path against the reference build
                                                   (--precomputed: every other one with precomputed colours and covariances)
    python tests/ref_report.py --arbitrate 32,89   sweep cases with the CPU oracle between the two
    python tests/ref_report.py --pixel 217         the worst pixel of a sweep case and the contributors near a threshold there"""
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


def scenes():
    """(no flag) three scenes: every array of the reference build against the CPU oracle's"""
    for (P, W, H, seed, fc, hi) in ((3000, 160, 96, 1, 9, 0.06), (20000, 320, 200, 2, 5, 0.05), (50000, 640, 360, 3, 10, 0.03)):
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

def cubemap():
    """`--cubemap`: the prefilters level by level, HIP operators against render-utils' kernels"""
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

def full(c3_only=False):
    """`--full [--c3-only]`: time the reference build at the bench workloads"""
    # the bench workloads (bench.py CONFIGS) through the reference build on this GPU: what a hipify port of the reference
    # delivers on MI355X, to read next to bench.py's ms_per_step for the same workloads (profiles/r03_bench*.json)
    cfgs = (("c3", (1_000_000, 1920, 1080, 9)), ("c2", (500_000, 1920, 1080, 5)), ("c5 shape", (2_000_000, 1920, 1080, 9)))
    for name, (P, W, H, fc) in cfgs[:1] if c3_only else cfgs:
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


sweep_scene = Hh.sweep_scene


def sweep(n, start=0, precomputed=False):
    """`--sweep N [START]`: N random scenes (helpers.sweep_scene), HIP path against the reference build (helpers.check_sweep_case:
    radii exact, observe / images with threshold proofs, blend sums and well-conditioned gradients element-wise, exceptions
    beyond the counted budget with their threshold proof); prints the failing cases.  tests/test_reference_gpu.py runs a
    slice of it under -m gpu."""
    bad = 0
    for case in range(start, start + n):
        try:
            Hh.check_sweep_case(reference, case, precomputed=precomputed and case % 2 == 1)
        except AssertionError as e:
            bad += 1
            print("FAIL", Hh.sweep_scene(case)[2], "::", str(e)[:240])
    print(f"sweep: {n} scenes from case {start}, {bad} failing")


def arbitrate(cases):
    """`--arbitrate c1,c2,...`: for sweep cases, the blend sums and images of the HIP path and of the reference build, each against
    the CPU oracle (double accumulators): which of the two fp32 evaluations is off when they disagree."""
    import gs2m_native
    for case in cases:
        sc, refbin, tag = sweep_scene(case)
        f, gr = Hh.run_oracle(oracle, sc)
        r, rg = Hh.run_oracle(reference, sc)
        gs2m_native.set_reference_binning(refbin)
        out, g = Hh.run_hip(sc); sums = Hh.run_hip_sums(sc)
        gs2m_native.set_reference_binning(False)
        print(tag)
        for k in ("means2D", "conics", "opacities", "colors", "features"):
            ff = 1e-4 if k == "conics" else 1e-5
            a = Hh.grad_stats(sums[k], gr[k].reshape(sums[k].shape), floor_frac=ff)[0]
            b = Hh.grad_stats(rg[k].reshape(sums[k].shape), gr[k].reshape(sums[k].shape), floor_frac=ff)[0]
            c = Hh.grad_stats(sums[k], rg[k].reshape(sums[k].shape), floor_frac=ff)[0]
            print(f"  sum:{k:10s} elements outside 1e-3: HIP vs oracle {a:.2e}, reference build vs oracle {b:.2e}, HIP vs reference build {c:.2e}")
        for name, x, y, z in (("color", out["color"], r.color, f.color), ("buffer[0]", out["buffer"][0], r.buffer[0], f.buffer[0])):
            print(f"  {name}: pixels off by more than 1e-4: HIP vs oracle {(np.abs(x - z) > 1e-4).mean():.2e}, reference build vs oracle {(np.abs(y - z) > 1e-4).mean():.2e}, HIP vs reference build {(np.abs(x - y) > 1e-4).mean():.2e}")



def pixel_case(case):
    """`--pixel CASE`: the worst alpha-channel pixel of a sweep case and every contributor of its tile list with its alpha and
    transmittance there (reference-order fp32 arithmetic on the reference build's own state)."""
    import gs2m_native
    sc, refbin, tag = sweep_scene(case)
    r, _ = Hh.run_oracle(reference, sc, backward=False)
    gs2m_native.set_reference_binning(refbin)
    out, _ = Hh.run_hip(sc, backward=False)
    gs2m_native.set_reference_binning(False)
    d = np.maximum(np.abs(out["buffer"][0] - r.buffer[0]), np.abs(out["color"] - r.color).max(0))
    y, x = np.unravel_index(d.argmax(), d.shape)
    print(f"{tag}: worst pixel ({x}, {y}) (alpha channel and colour): HIP alpha {out['buffer'][0][y, x]:.7f} colour {out['color'][:, y, x]} reference alpha {r.buffer[0][y, x]:.7f} "
          f"colour {r.color[:, y, x]} diff {d[y, x]:.3e}; quadrant ({(x % 16) // 8}, {(y % 16) // 8})")
    f32 = np.float32
    tile = (y // 16) * r.tiles_x + (x // 16)
    lo_, hi_ = int(r.ranges[tile, 0]), int(r.ranges[tile, 1])
    T = f32(1.0)
    for k, gid in enumerate(r.vals_sorted[lo_:hi_]):
        mx, my = r.means2D[gid]; A, B, C, op = r.conic_opacity[gid]
        dx = f32(mx - f32(x)); dy = f32(my - f32(y))
        power = f32(f32(f32(-0.5) * f32(f32(f32(A * dx) * dx) + f32(f32(C * dy) * dy))) - f32(f32(B * dx) * dy))
        if power > 0:
            continue
        alpha = min(0.99, float(op) * float(np.exp(np.float64(power))))
        if alpha < 1.0 / 255.0:
            if alpha > 0.9 / 255.0:
                print(f"   (k={k} gid={gid} alpha*255={alpha * 255:.5f}: below the threshold)")
            continue
        contrib = alpha * float(T)
        if abs(contrib - d[y, x]) < 0.5 * d[y, x] or alpha * 255 < 1.05 or contrib > 0.5 * d[y, x] and alpha * 255 < 1.3:
            print(f"   k={k} gid={gid} alpha*255={alpha * 255:.5f} T={float(T):.5f} alpha*T={contrib:.3e} radius={r.radii[gid]} mean=({mx:.1f},{my:.1f}) conic=({A:.3e},{B:.3e},{C:.3e}) opacity={op:.4f}")
        if f32(T * f32(1.0 - f32(alpha))) < f32(1e-4):
            break
        T = f32(T * f32(1.0 - f32(alpha)))


if __name__ == "__main__":
    arg = lambda flag, k=1: sys.argv[sys.argv.index(flag) + k]
    if "--cubemap" in sys.argv:
        cubemap()
    elif "--sweep" in sys.argv:
        i = sys.argv.index("--sweep")
        sweep(int(sys.argv[i + 1]), int(sys.argv[i + 2]) if len(sys.argv) > i + 2 and not sys.argv[i + 2].startswith("-") else 0,
              precomputed="--precomputed" in sys.argv)
    elif "--arbitrate" in sys.argv:
        arbitrate([int(c) for c in arg("--arbitrate").split(",")])
    elif "--pixel" in sys.argv:
        pixel_case(int(arg("--pixel")))
    elif "--full" in sys.argv:
        full("--c3-only" in sys.argv)
    else:
        scenes()
