"""Where a material-stage view spends its GPU time (tools/pbr_bench.py's `view`), by group of kernels."""
import os, sys, re, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gs-2m_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
src = open(os.path.join(ROOT, "tools", "pbr_bench.py")).read().split('stage("CubemapLight')[0]
exec(src)
for _ in range(3):
    view()
torch.cuda.synchronize()
N = 5
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(N):
        view()
    torch.cuda.synchronize()
groups = collections.OrderedDict((k, [0.0, 0]) for k in ("rasterizer", "render ops (fused pre/post)", "specular/diffuse prefilter", "texture lookups", "BLAS", "torch elementwise / reduce / copy"))
for e in prof.key_averages():
    n = e.key
    t = e.self_device_time_total if hasattr(e, "self_device_time_total") else e.self_cuda_time_total
    if re.search(r"blend_|preprocess_kernel|gaussian_bwd|row_reduce|rs_|emit_kernel|scan_tt|ranges_kernel|observe_kernel|zero_kernel", n): g = "rasterizer"
    elif re.search(r"pack_features|gbuffer_post|sobel_normal|activate", n): g = "render ops (fused pre/post)"
    elif re.search(r"specular_kernel|diffuse_kernel|axis_area", n): g = "specular/diffuse prefilter"
    elif re.search(r"texture_|shade_", n): g = "texture lookups"
    elif re.search(r"Cijk|gemm", n): g = "BLAS"
    else: g = "torch elementwise / reduce / copy"
    groups[g][0] += t / N / 1e3
    groups[g][1] += e.count / N
tot = sum(v[0] for v in groups.values())
for k, (ms, cnt) in groups.items():
    print("%-38s %7.3f ms  %5.1f launches per view" % (k, ms, cnt))
print("%-38s %7.3f ms" % ("total GPU time per view", tot))
