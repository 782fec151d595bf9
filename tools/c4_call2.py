"""where the host time of one view goes (tools/c4_profile.py found a view whose step takes 3x its kernels)"""
import os, sys, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch
import c4_profile as C
import helpers as Hh
import gs2m_native
from diff_gaussian_rasterization import GaussianRasterizer
import diff_gaussian_rasterization as dgr
calls = C.calls_from_geometry(os.path.join(ROOT, "bench_data", "c4_geom.npz"))
dev = "cuda"
for ci in (0, 2, 0, 1):
    sc = calls[ci]
    g = {k: v.to(dev).requires_grad_(True) for k, v in sc["g"].items()}
    P = g["means3D"].shape[0]
    st = Hh.settings_for(sc, dev)
    Gc, Gb = sc["Gc"].to(dev), sc["Gb"].to(dev)
    means2D = torch.zeros(P, 4, device=dev, requires_grad=True)
    rast = GaussianRasterizer(st)
    def fwd():
        for t in list(g.values()) + [means2D]:
            t.grad = None
        color, radii, observe, buffer = rast(g["means3D"], means2D, g["opacities"], features=g["features"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
        return (color * Gc).sum() + (buffer * Gb).sum()
    for _ in range(10):
        fwd().backward()
    torch.cuda.synchronize()
    tf, tb = [], []
    ms0 = torch.cuda.memory_stats()
    for _ in range(100):
        t0 = time.perf_counter(); l = fwd(); torch.cuda.synchronize(); t1 = time.perf_counter(); l.backward(); torch.cuda.synchronize(); t2 = time.perf_counter()
        tf.append((t1 - t0) * 1e3); tb.append((t2 - t1) * 1e3)
    ms1 = torch.cuda.memory_stats()
    import numpy as np
    print(f"call {ci}: forward median {np.median(tf):.3f} max {max(tf):.3f} ms; backward median {np.median(tb):.3f} max {max(tb):.3f}; device allocs {ms1['num_device_alloc'] - ms0['num_device_alloc']} frees {ms1['num_device_free'] - ms0['num_device_free']}", flush=True)
    print("  forward ms:", [round(x, 2) for x in tf], flush=True)
    print("  backward ms:", [round(x, 2) for x in tb], flush=True)
