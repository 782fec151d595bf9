"""Forward-only timing at the bench workload (GPU box): isolates the front end + blend forward."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd")):
    sys.path.insert(0, p)
import torch
import gs2m_synth as S
import diff_gaussian_rasterization as dgr
P, W, H, fc = 1_000_000, 1920, 1080, 9
dev = "cuda"
cam = S.make_camera(W, H)
g = {k: v.to(dev) for k, v in S.make_gaussians(P, cam, seed=0).items()}
e = torch.Tensor([])
args = (torch.zeros(3, device=dev), g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, g["features"],
        cam["viewmatrix"].to(dev), cam["projmatrix"].to(dev), cam["tanfovx"], cam["tanfovy"], H, W, g["shs"], 3,
        cam["campos"].to(dev), False, fc)
print("current stream handle:", torch.cuda.current_stream().cuda_stream)
import gs2m_native, statistics
for _ in range(50):
    dgr._C.rasterize_gaussians(*args)
res = {0: [], 1: []}
for rep in range(12):
    for on in (1, 0):  # series 1: host polls num_rendered; series 0: hipStreamSynchronize
        gs2m_native.set_spin_wait(on)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            dgr._C.rasterize_gaussians(*args)
        torch.cuda.synchronize()
        res[on].append((time.perf_counter() - t0) / 100 * 1e3)
for on in (1, 0):
    print("series %d: median %.4f ms  min %.4f" % (on, statistics.median(res[on]), min(res[on])))
