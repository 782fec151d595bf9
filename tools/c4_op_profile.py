"""torch.profiler over a few iterations of the C4 run (geometry stage, ~400 k Gaussians): which aten / custom ops the device time of an
iteration belongs to.  python tools/c4_op_profile.py [first profiled iteration] [iterations profiled]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd")):
    sys.path.insert(0, p)
import torch
from torch.profiler import profile, ProfilerActivity
import gs2m_train
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1400
count = int(sys.argv[2]) if len(sys.argv) > 2 else 20
prof = {"p": None}

def cb(it, g, cams, gts):
    if it == first:
        torch.cuda.synchronize()
        prof["p"] = profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA])
        prof["p"].__enter__()
    elif it == first + count:
        torch.cuda.synchronize()
        prof["p"].__exit__(None, None, None)
        print(prof["p"].key_averages().table(sort_by="self_cuda_time_total", row_limit=45, max_name_column_width=60), flush=True)
        ev = prof["p"].key_averages()
        tot = sum(e.self_device_time_total for e in ev)
        print(f"device time per iteration: {tot / count:.1f} us over {count} iterations at {g.get_xyz.shape[0]} Gaussians")

with tempfile.TemporaryDirectory() as tmp:
    scene = gs2m_train.c4_scene(os.path.join(tmp, "c4"))
    gs2m_train.c4_run(None, iterations=first + count + 2, schedule_iterations=5000, scene=scene, callback=cb)
