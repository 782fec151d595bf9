"""per-wave instance sums of the bench clouds (what GS2M_CROWDED_WAVE is compared with): python tools/wave_sums.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
import gs2m_native, gs2m_synth as S
import diff_gaussian_rasterization as dgr
for name, (P, W, H, fc) in {"c2": (500_000, 1920, 1080, 5), "c3": (1_000_000, 1920, 1080, 9), "c5": (2_000_000, 1920, 1080, 9)}.items():
    for refbin in (False, True):
        gs2m_native.set_reference_binning(refbin)
        cam = S.make_camera(W, H)
        g = {k: v.cuda() for k, v in S.make_gaussians(P, cam, seed=0).items()}
        e = torch.Tensor([])
        R, color, radii, observe, buffer, geomB, binB, imgB = dgr._C.rasterize_gaussians(
            torch.zeros(3, device="cuda"), g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, g["features"],
            cam["viewmatrix"].cuda(), cam["projmatrix"].cuda(), cam["tanfovx"], cam["tanfovy"], H, W, g["shs"], 3, cam["campos"].cuda(), False, fc)
        torch.cuda.synchronize()
        lay = gs2m_native.debug_layout(P, R, W, H)
        al = (-geomB.data_ptr()) % 256
        tt = geomB[al + lay.tiles_touched: al + lay.tiles_touched + 4 * P].cpu().numpy().view(np.uint32).astype(np.int64)
        cn = geomB[al + lay.counters: al + lay.counters + 256].cpu().numpy().view(np.uint32)
        w = np.concatenate([tt, np.zeros((-P) % 64, np.int64)]).reshape(-1, 64)
        for th in (48, 32, 24):
            ws = (w * (w < th)).sum(1)
            print(f"{name} refbin {refbin}: R {R} U {cn[3]}; tiles max {tt.max()}, >= {th}: {(tt >= th).sum()}; wave sums (light < {th}) mean {ws.mean():.0f} p99.9 {np.percentile(ws, 99.9):.0f} max {ws.max()}")
gs2m_native.set_reference_binning(False)
