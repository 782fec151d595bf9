"""Random-shape sweep of the widened-row operators against their torch / numpy formulations (the unit tests use fixed shapes)."""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import torch.nn.functional as F

rnd = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
fails = 0


def check(name, ok, info=""):
    global fails
    if not ok:
        fails += 1
        print("FAIL", name, info)


# ---- fused SSIM
from fused_ssim import fused_ssim
from test_ssim_gpu import torch_ssim_map
for _ in range(25):
    B, C, H, W = rnd.randint(1, 3), rnd.randint(1, 4), rnd.randint(1, 140), rnd.randint(1, 200)
    pad = rnd.choice(["same", "valid"]) if min(H, W) > 10 else "same"
    a = torch.rand(B, C, H, W, device="cuda", requires_grad=True)
    b = torch.rand(B, C, H, W, device="cuda")
    ref = torch_ssim_map(a, b, pad).mean()
    ref.backward()
    g0 = a.grad.clone(); a.grad = None
    got = fused_ssim(a, b, pad)
    got.backward()
    check("ssim", abs(got.item() - ref.item()) < 2e-5 and (a.grad.double() - g0).abs().max().item() < 1e-4 * g0.abs().max().item() + 1e-8, (B, C, H, W, pad))

# ---- fused Adam
import gs2m_optim
for _ in range(15):
    shapes = [tuple(rnd.randint(1, 40) for _ in range(rnd.randint(1, 3))) + () for _ in range(rnd.randint(1, 20))]
    shapes = [(rnd.randint(1, 5000),) + s for s in shapes]
    pa = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    lr, eps, betas = 10 ** rnd.uniform(-4, -1), 10 ** rnd.uniform(-15, -6), (rnd.uniform(0.5, 0.95), rnd.uniform(0.9, 0.9999))
    oa, ob = torch.optim.Adam(pa, lr=lr, eps=eps, betas=betas), gs2m_optim.Adam(pb, lr=lr, eps=eps, betas=betas)
    for it in range(3):
        for x, y in zip(pa, pb):
            x.grad = torch.randn_like(x) * 10 ** rnd.uniform(-3, 2)
            y.grad = x.grad.clone()
        oa.step(); ob.step()
    check("adam", all(torch.equal(x, y) for x, y in zip(pa, pb)), (len(shapes), lr, eps, betas))

# ---- texture lookups against the numpy oracle
import nvdiffrast.torch as dr
from oracle import texture_oracle as O
for _ in range(12):
    w, C = rnd.choice([1, 2, 3, 4, 8, 16]), rnd.randint(1, 4)
    n = rnd.randint(1, 300)
    tex = torch.rand(1, 6, w, w, C)
    d = torch.randn(n, 3) * torch.tensor([rnd.choice([1.0, 0.01]), 1.0, rnd.choice([1.0, 100.0])])
    got = dr.texture(tex.cuda(), d.view(1, 1, n, 3).cuda().contiguous(), filter_mode="linear", boundary_mode="cube").cpu().view(n, C)
    want = O.cube_sample([tex[0].numpy()], d.numpy())
    check("cube linear", np.abs(got.numpy() - want).max() < 5e-5, (w, C, n))
for _ in range(8):
    L = rnd.randint(2, 5)
    ws = [2 ** (L - 1 - l) * rnd.choice([1, 2]) for l in range(L)]
    ws = [ws[0] // (2 ** l) for l in range(L)] if ws[0] >= 2 ** (L - 1) else [2 ** (L - 1 - l) for l in range(L)]
    levels = [torch.rand(1, 6, w, w, 3) for w in ws]
    n = rnd.randint(1, 300)
    d, bias = torch.randn(n, 3), torch.rand(n) * (L + 1) - 1
    got = dr.texture(levels[0].cuda(), d.view(1, 1, n, 3).cuda().contiguous(), mip=[l.cuda() for l in levels[1:]], mip_level_bias=bias.view(1, 1, n).cuda(),
                     filter_mode="linear-mipmap-linear", boundary_mode="cube").cpu().view(n, 3)
    want = O.cube_sample([l[0].numpy() for l in levels], d.numpy(), bias.numpy())
    check("cube mip", np.abs(got.numpy() - want).max() < 5e-5, (ws, n))

# ---- grid_sample border
import gs2m_mvs as MV
for _ in range(20):
    C, H, W, N = rnd.randint(1, 4), rnd.randint(2, 90), rnd.randint(2, 120), rnd.randint(1, 4000)
    img = torch.randn(C, H, W, device="cuda", requires_grad=True)
    grid = (torch.rand(N, 2, device="cuda") * 2.6 - 1.3).requires_grad_(True)
    G = torch.randn(N, C, device="cuda")
    ref = F.grid_sample(img[None], grid.view(1, -1, 1, 2), mode="bilinear", padding_mode="border", align_corners=True)[0, :, :, 0].permute(1, 0)
    (ref * G).sum().backward()
    gi, gg = img.grad.clone(), grid.grad.clone(); img.grad = grid.grad = None
    got = MV.grid_sample_border(img, grid)
    (got * G).sum().backward()
    check("grid_sample", (got - ref).abs().max().item() < 1e-5 and (img.grad - gi).abs().max().item() < 1e-4 * max(1.0, gi.abs().max().item())
          and (grid.grad - gg).abs().max().item() < 2e-4 * max(1.0, gg.abs().max().item()), (C, H, W, N))
print("failures:", fails)
