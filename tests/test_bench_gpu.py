"""bench.py's one-line JSON contract (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling /
vs_baseline / dtype / data / config.workload + roofline + cpu_baseline), on a short run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_has_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "8"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "median_ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 8 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "views/s" and d["value"] > 0 and abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and "traffic" in rf
    assert rf["kernel"] == "blend_bwd" and rf["measured"] == "timed region", "the contract workload: live bracket of the backward blend"
    assert d["render_level_ms"] > d["ms_per_step"] * 0.9 and d["train_step_ms"] > d["ms_per_step"] * 0.9  # the callers
    assert d["material_step_ms"] > d["train_step_ms"], "the material stage adds the light's prefilter and the deferred shading"
    # the same workload on the reference's own (bit-identical) instance list, beside the default mode's subset of it
    rb = d["reference_binning"]
    assert d["reference_binning_ms_per_step"] == rb["ms_per_step"] > 0 and rb["num_rendered"] > d["config"]["num_rendered"]
    assert abs(sum(d["stages_ms"].values()) - d["ms_per_step"]) < 0.25 * d["ms_per_step"], "stage totals add up to about the step"
    # the same K steps timed right behind the first warm-up (inside the clock governor's ramp), beside the headline
    cr = d["clock_ramp"]
    assert cr["ms_per_step_at_start"] > 0 and abs(cr["value_at_start"] - 1e3 / cr["ms_per_step_at_start"]) < 1e-2 * cr["value_at_start"]
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"] and cb["sample"]


def test_bench_two_ranks_share_one_gpu_over_gloo():
    """The N > 1 path of bench.py (launch contract, one collective per step over the gradient arena, blocking and
    pipelined timings, max over ranks) run functionally: two ranks on the one GPU of the box, gloo instead of RCCL."""
    # two processes share the one GPU here: the radix sort must not assume it has the device to itself
    env = dict(os.environ, GS2M_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", GS2M_BENCH_WATCHDOG_S="240")  # (bench.py switches the tile sort to tickets itself at N > 1)
    port = 29700 + os.getpid() % 200
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--gaussians", "30000", "--width", "320", "--height", "192"], capture_output=True, text=True, timeout=400, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    # north_star / BASELINE configs[4]: ONE view per rank and step on the camera ring, blocking sum of its gradients: that is `value`
    assert d["config"]["views_per_rank"] == 1 and d["views_per_step"] == 2
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"], "whole-job views/s"
    assert "one view per rank" in d["config"]["workload"] and "ring" in d["config"]["workload"] and "collective" in d["config"]["workload"]
    assert "cpu_baseline" not in d and d["median_ms_per_step"] > 0
    # beside it, labelled: pipelined, equal-work cameras, two accumulated views per rank, no collective at all
    for key, views in (("pipelined", 2), ("equal_work", 2), ("accumulate_v2", 4), ("compute_only", 2)):
        assert d[key + "_ms_per_step"] > 0 and abs(d[key + "_value"] - views * 1e3 / d[key + "_ms_per_step"]) < 1e-2 * d[key + "_value"], key
    assert 0 < d["scaling_efficiency_vs_compute_only"] <= 1.5
    rf = d["roofline"]  # small frame: either blend kernel may be the longer one; the line says which and how it was timed
    assert rf["kernel"] in ("blend_bwd", "blend_fwd") and rf["avg_launch_ms"] > 0
    assert rf["measured"] == ("timed region" if rf["kernel"] == "blend_bwd" else "stage pass (untimed, same step)")


def test_bench_gpus_2_as_typed_starts_its_own_ranks():
    """`python bench.py --gpus 2 --steps 3 --warmup 2` exactly as the driver would type it -- no launcher, WORLD_SIZE unset:
    bench.py starts `torch.distributed.run` as a child before it touches the GPU, relays rank 0's line and the child's
    exit code (BASELINE configs[4]; here both ranks share the box's one GPU over gloo)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(GS2M_DIST_BACKEND="gloo", GS2M_BENCH_WATCHDOG_S="500")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "rank 0 prints ONE line"
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 2 and d["scaling"] == "weak"
    assert d["config"]["views_per_rank"] == 1 and d["views_per_step"] == 2
    for key in ("pipelined", "equal_work", "accumulate_v2", "compute_only"):
        assert d[key + "_ms_per_step"] > 0 and d[key + "_value"] > 0, key
