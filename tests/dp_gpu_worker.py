"""Worker of tests/test_dp_gpu.py: the data-parallel reduction of REAL rasterizer gradients through RCCL (backend "nccl"
on ROCm) on a one-rank group -- what a single GPU allows.  Exercises the code the N-GPU bench runs: the in-place sum over the
binding's gradient arena (all-reduce and the reduce-scatter + un-waited all-gather form, which relies on RCCL's stream
order), the SH band trimming, and the densification side channels computed from the un-summed dL/dmeans2D."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import torch.distributed as dist

import helpers as Hh
from gs2m_dp import GradReducer
import gs2m_arena

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1])
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from diff_gaussian_rasterization import GaussianRasterizer

sc = Hh.make_scene(20_000, 320, 200, seed=21, fc=9)
dev = "cuda"
for mode in ("allreduce", "rs_ag"):
    g = {k: v.to(dev).requires_grad_(True) for k, v in sc["g"].items()}
    m2 = torch.zeros(20_000, 4, device=dev, requires_grad=True)
    color, radii, observe, buffer = GaussianRasterizer(Hh.settings_for(sc, dev))(
        g["means3D"], m2, g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"], features=g["features"])
    ((color * sc["Gc"].to(dev)).sum() + (buffer * sc["Gb"].to(dev)).sum()).backward()
    leaves = [g["means3D"], g["shs"], g["opacities"], g["scales"], g["rotations"], g["features"]]
    before = [t.grad.clone() for t in leaves]
    m2_before = m2.grad.clone()
    red = GradReducer(mode=mode, always_communicate=True)
    stats = red.reduce_densification_stats_async(m2.grad, radii, observe)
    pend = red.reduce_flat_async([t.grad for t in leaves], sh_active={1: 4})  # as at SH degree 1
    out = pend.wait()
    gn, ga, cnt, mr, obs = stats.wait()
    torch.cuda.synchronize()
    kinds = sorted(k for k, _ in red.last_plan)
    assert kinds == ["arena", "sh"], red.last_plan                       # in place over the arena + the packed SH bands, no copy
    info = gs2m_arena.lookup(leaves[0].grad)
    assert info is not None and all(o is t.grad for o, t in zip(out, leaves))
    for k, (a, b) in enumerate(zip(before, out)):
        if k == 1:  # SH: only the first 4 coefficients travelled (one rank: unchanged)
            assert torch.equal(a, b)
        else:
            assert torch.equal(a, b), k                                     # sum over ONE rank = the rank's own gradient
    assert torch.equal(m2.grad, m2_before), "dL/dmeans2D is not part of the summed range"
    vis = radii > 0
    assert torch.allclose(gn[:, 0], torch.norm(m2_before[:, :2], dim=-1) * vis) and torch.allclose(ga[:, 0], torch.norm(m2_before[:, 2:], dim=-1) * vis)
    assert torch.equal(cnt[:, 0] > 0, vis) and torch.equal(mr, radii) and torch.equal(obs, observe)
    print("mode", mode, "plan", red.last_plan)
dist.destroy_process_group()
print("DP_GPU_OK")
