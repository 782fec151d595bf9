"""Optimizer step at the bench workload (1M Gaussians, the reference's nine groups = 64 floats per Gaussian):
torch.optim.Adam (foreach, the reference's) against gs2m_optim.Adam (one fused launch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd")):
    sys.path.insert(0, p)
import torch
import gs2m_optim

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
shapes = [(3,), (1, 3), (15, 3), (1,), (3,), (4,), (3,), (1,), (1,)]
lrs = [1.6e-4, 2.5e-3, 1.25e-4, 0.05, 5e-3, 1e-3, 0.05, 0.05, 0.05]


def make(cls, **kw):
    params = [torch.nn.Parameter(torch.randn((P,) + s, device="cuda")) for s in shapes]
    for p in params:
        p.grad = torch.randn_like(p)
    return cls([{"params": [p], "lr": lr} for p, lr in zip(params, lrs)], lr=0.0, eps=1e-15, **kw)


n_el = P * sum(int(torch.tensor(s).prod()) for s in shapes)
for name, opt in (("torch.optim.Adam (foreach)", make(torch.optim.Adam)), ("torch.optim.Adam (fused=True)", make(torch.optim.Adam, fused=True)),
                  ("gs2m_optim.Adam", make(gs2m_optim.Adam))):
    for _ in range(5):
        opt.step()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 50
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(n):
        opt.step()
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / n
    print("%-32s %.3f ms per step (GPU), %.3f ms wall; %.0f GB/s of the 28 B/element a single pass needs" % (
        name, ms, (time.perf_counter() - t0) / n * 1e3, n_el * 28 / ms / 1e6))
    del opt
    torch.cuda.empty_cache()
