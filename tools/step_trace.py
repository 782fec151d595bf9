"""Per-step GPU time of the first N bench steps (HIP events around each step): shows how long the warm-up transient lasts.
usage: python tools/step_trace.py [N]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "gs-2m_amd"))
import torch
import gs2m_synth as S
from diff_gaussian_rasterization import GaussianRasterizationSettings, rasterize_gaussians
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
P, W, H, fc = 1_000_000, 1920, 1080, 9
dev = torch.device("cuda", 0)
cam = S.make_camera(W, H)
g = S.make_gaussians(P, cam, seed=0)
Gc, Gb = S.make_upstream_grads(H, W, seed=0)
Gc, Gb = Gc.to(dev), Gb.to(dev)
prm = {k: v.to(dev).requires_grad_(True) for k, v in g.items()}
m2 = torch.zeros(P, 4, device=dev, requires_grad=True)
st = GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=torch.zeros(3, device=dev),
                                   scale_modifier=1.0, viewmatrix=cam["viewmatrix"].to(dev), projmatrix=cam["projmatrix"].to(dev), sh_degree=3,
                                   campos=cam["campos"].to(dev), prefiltered=False, feature_count=fc)
e = torch.Tensor([])
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(N)]
wall = []
if os.environ.get("REFBIN"):  # the reference's own instance list (gs2m_set_reference_binning)
    import gs2m_native
    gs2m_native.set_reference_binning(True)
if os.environ.get("PREWARM"):
    x = torch.rand(16 << 20, 32, device=dev)  # 2 GB of 128-B rows
    idx = torch.randint(0, 16 << 20, (8 << 20,), device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < float(os.environ["PREWARM"]) * 1e-3:
        for _ in range(4):
            y = x.index_select(0, idx)  # random 128-B gathers from HBM + 1 GB of streaming writes
        torch.cuda.synchronize()
    del x, y, idx
if os.environ.get("NOGC"):
    import gc
    gc.collect(); gc.disable()
torch.cuda.synchronize()
for i in range(N):
    for t in list(prm.values()) + [m2]:
        t.grad = None
    t0 = time.perf_counter()
    ev[i][0].record()
    color, radii, observe, buffer = rasterize_gaussians(prm["means3D"], m2, prm["shs"], e, prm["opacities"], prm["scales"], prm["rotations"], e, prm["features"], st, None)
    torch.autograd.backward([color, buffer], [Gc, Gb])
    ev[i][1].record()
    wall.append(time.perf_counter() - t0)
    if os.environ.get("EMPTY_AT") and i == int(os.environ["EMPTY_AT"]):
        torch.cuda.synchronize()
        del color, radii, observe, buffer
        for t in list(prm.values()) + [m2]:
            t.grad = None
        torch.cuda.empty_cache()
    if os.environ.get("REFBIN_AT") and i == int(os.environ["REFBIN_AT"]):  # switch to the reference's instance list mid-run (bench.py's second leg)
        import gs2m_native
        gs2m_native.set_reference_binning(True)
    if os.environ.get("SLEEP_AT") and i == int(os.environ["SLEEP_AT"]):
        torch.cuda.synchronize()
        time.sleep(0.5)
torch.cuda.synchronize()
if os.environ.get("STAGES"):
    import gs2m_native
    rows = []
    for i in range(N, N + 40):
        gs2m_native.profile_mode(2)
        for t in list(prm.values()) + [m2]:
            t.grad = None
        color, radii, observe, buffer = rasterize_gaussians(prm["means3D"], m2, prm["shs"], e, prm["opacities"], prm["scales"], prm["rotations"], e, prm["features"], st, None)
        torch.autograd.backward([color, buffer], [Gc, Gb])
        torch.cuda.synchronize()
        d = gs2m_native.profile_collect()
        rows.append({k: v[0] for k, v in d.items()})
    gs2m_native.profile_mode(0)
    for k in rows[0]:
        print(k, " ".join(f"{r[k]:.3f}" for r in rows[::3]))
ms = [a.elapsed_time(b) for a, b in ev]
print("gpu ms per step:", " ".join(f"{x:.3f}" for x in ms))
print("host ms per step:", " ".join(f"{x * 1e3:.2f}" for x in wall))
