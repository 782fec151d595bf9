"""isolated host stalls (tens of ms) in a loop of forward + backward: is it Python's cyclic GC?  usage: stall_probe.py [gc|nogc|freeze]"""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import torch
import c4_profile as C
import helpers as Hh
from diff_gaussian_rasterization import GaussianRasterizer
mode = sys.argv[1] if len(sys.argv) > 1 else "gc"
calls = C.calls_from_geometry(os.path.join(ROOT, "bench_data", "c4_geom.npz"))
dev = "cuda"
sc = calls[0]
g = {k: v.to(dev).requires_grad_(True) for k, v in sc["g"].items()}
P = g["means3D"].shape[0]
st = Hh.settings_for(sc, dev)
Gc, Gb = sc["Gc"].to(dev), sc["Gb"].to(dev)
means2D = torch.zeros(P, 4, device=dev, requires_grad=True)
rast = GaussianRasterizer(st)
def step():
    for t in list(g.values()) + [means2D]:
        t.grad = None
    color, radii, observe, buffer = rast(g["means3D"], means2D, g["opacities"], features=g["features"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"])
    ((color * Gc).sum() + (buffer * Gb).sum()).backward()
for _ in range(20):
    step()
torch.cuda.synchronize()
if mode == "nogc":
    gc.disable()
elif mode == "freeze":
    gc.collect(); gc.freeze()
gcs = []
gc.callbacks.append(lambda phase, info: gcs.append((phase, info["generation"], time.perf_counter())))
ts = []
for i in range(1500):
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
import numpy as np
print(mode, "median", round(float(np.median(ts)), 3), "mean", round(float(np.mean(ts)), 3), "stalls > 5 ms:", [(i, round(t, 1)) for i, t in enumerate(ts) if t > 5.0])
g2 = [(p, gen, t) for p, gen, t in gcs if gen == 2]
print("  gen-2 collections:", len(g2) // 2, "durations ms:", [round((g2[i + 1][2] - g2[i][2]) * 1e3, 1) for i in range(0, len(g2) - 1, 2)], "objects tracked:", len(gc.get_objects()))
