"""CubemapLight.build_mips (pbr/light.py:86-99) prefilter stack at the reference's sizes: specular_cubemap on 512^2 ... 32^2
with roughness 0.04 ... 0.5, 16^2 with roughness 1, diffuse_cubemap on 16^2; forward and backward."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd")):
    sys.path.insert(0, p)
import torch
import render_utils as RU

levels = [(512, 0.04), (256, 0.155), (128, 0.27), (64, 0.385), (32, 0.5), (16, 1.0)]
tot_f = tot_b = 0.0
for N, r in levels + [(16, None)]:
    x = torch.rand(6, N, N, 3, device="cuda", requires_grad=True)
    fn = (lambda: RU.diffuse_cubemap(x)) if r is None else (lambda: RU.specular_cubemap(x, r))
    res = []
    for fb in (False, True):
        def run():
            out = fn()
            if fb:
                torch.autograd.grad(out, x, torch.ones_like(out))
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
    tot_f += res[0]; tot_b += res[1]
    print("%-10s res %4d roughness %-6s fwd %.3f ms   fwd+bwd %.3f ms" % ("diffuse" if r is None else "specular", N, r, res[0], res[1]))
print("whole stack: fwd %.3f ms, fwd+bwd %.3f ms" % (tot_f, tot_b))
