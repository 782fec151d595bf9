"""a uniformly random subset of exactly k of n indices on the GPU: randperm prefix (a full sort of n random keys) against selection"""
import torch, time
dev = "cuda"
n, k = 450_000, 102_400
idx = torch.arange(n, device=dev)
def t(f, reps=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
def a(): return idx[torch.randperm(n, device=dev)[:k]]
def b():
    key = torch.rand(n, device=dev)
    th = torch.kthvalue(key, k).values
    return idx[torch.nonzero(key <= th).squeeze(1)[:k]]
def c():
    key = torch.rand(n, device=dev)
    return idx[torch.topk(key, k, largest=False, sorted=False).indices]
def d():  # Bernoulli thinning by an integer threshold on 32 random bits, then exactly k by a second draw among the kept (rare top-up omitted)
    key = torch.rand(n, device=dev)
    return idx[torch.nonzero(key < (k / n)).squeeze(1)]
for name, f in (("randperm prefix", a), ("rand + kthvalue + nonzero", b), ("rand + topk(unsorted)", c), ("rand + threshold + nonzero (size ~ k)", d)):
    print(f"{name:45s} {t(f):8.1f} us   size {f().numel()}")
