"""a short piece of the C4 substitute run (gs2m_train.c4_run) for kernel traces: python tools/c4_train_short.py [iterations] [schedule iterations]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd")):
    sys.path.insert(0, p)
import gs2m_train
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
sched = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
with tempfile.TemporaryDirectory() as tmp:
    scene = gs2m_train.c4_scene(os.path.join(tmp, "c4"))
    model, st = gs2m_train.c4_run(None, iterations=iters, schedule_iterations=sched, scene=scene)
print({k: st[k] for k in ("it_per_s", "points_start", "points_max", "points_end")})
