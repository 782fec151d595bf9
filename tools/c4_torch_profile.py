"""Which Python line launches which small kernel in a C4 training iteration: torch.profiler (with stacks) over a few geometry-stage
iterations of the C4 substitute run, started from the run's callback.   python tools/c4_torch_profile.py [first iteration] [iterations]"""
import os, sys, tempfile, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd")):
    sys.path.insert(0, p)
import torch
import gs2m_train
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
prof = torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True, record_shapes=True)
state = {"on": False}
def cb(it, g, cams, gts):
    if it == first and not state["on"]:
        torch.cuda.synchronize(); prof.__enter__(); state["on"] = True
    if it == first + n and state["on"]:
        torch.cuda.synchronize(); prof.__exit__(None, None, None); state["on"] = False
with tempfile.TemporaryDirectory() as tmp:
    scene = gs2m_train.c4_scene(os.path.join(tmp, "c4"))
    gs2m_train.c4_run(None, iterations=first + n + 1, schedule_iterations=5000, scene=scene, callback=cb)
# per CPU op (aten::...) that launched device kernels: total device time, count, and the innermost frames of this repository
ev = prof.events()
agg = collections.defaultdict(lambda: [0.0, 0, None])
for e in ev:
    if e.device_type == torch.autograd.DeviceType.CPU and e.kernels:
        dt = sum(k.duration for k in e.kernels)
        frames = [f for f in (e.stack or []) if "gs-2m_amd" in f or "bench.py" in f]
        where = " <- ".join(f.split("gs-2m_amd/")[-1] for f in frames[:3]) if frames else "(autograd engine / no repository frame)"
        key = (e.name, tuple(str(s) for s in (e.input_shapes or []))[:3], where)
        a = agg[key]; a[0] += dt; a[1] += len(e.kernels)
tot = sum(a[0] for a in agg.values())
print(f"{n} iterations from {first}: device time of kernels launched by profiled CPU ops {tot / n:.1f} us per iteration")
for (name, shapes, where), (dt, cnt, _) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:90]:
    print(f"{dt / n:8.1f} us  x{cnt / n:5.2f}  {name[:44]:44s} {str(shapes)[:60]:60s} {where[:150]}")
