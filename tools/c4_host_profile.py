"""Where the HOST spends a C4 training iteration (the loop is launch-bound on most boxes of the pool): cProfile over a stretch of geometry-stage
iterations of the C4 substitute run.   python tools/c4_host_profile.py [first iteration] [iterations]"""
import cProfile, os, pstats, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd")):
    sys.path.insert(0, p)
import torch
import gs2m_train
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
pr = cProfile.Profile()
st = {}
def cb(it, g, cams, gts):
    if it == first:
        torch.cuda.synchronize(); st["t0"] = time.perf_counter(); pr.enable()
    if it == first + n:
        pr.disable(); torch.cuda.synchronize(); st["t1"] = time.perf_counter()
with tempfile.TemporaryDirectory() as tmp:
    scene = gs2m_train.c4_scene(os.path.join(tmp, "c4"))
    gs2m_train.c4_run(None, iterations=first + n + 1, schedule_iterations=5000, scene=scene, callback=cb)
print(f"{n} iterations from {first}: {(st['t1'] - st['t0']) / n * 1e3:.3f} ms per iteration under cProfile")
s = pstats.Stats(pr)
s.sort_stats("tottime").print_stats(45)
s.sort_stats("cumulative").print_stats(40)
