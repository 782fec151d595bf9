"""Shader clock of the GPU under test (sysfs pp_dpm_sclk, current level) sampled while tools/step_trace.py's loop runs.

    python tools/clock_trace.py [steps]

The box shows every GPU of its node in sysfs; the one this process drives is taken to be the card whose clock moves the most
during the run.  Prints the step times of the loop, then the clock each time it changes.  Evidence for NOTEBOOK.md's note on
the warm-up ramp: the governor takes about half a second of continuous work to bring sclk from its idle level to 2400 MHz,
and drops it again after about half a second of idle, so the first few dozen steps of any short run are slower than the
steady state, by the clock ratio, in the ALU-bound kernels only."""
import glob, os, re, subprocess, sys, threading, time

paths = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))


def cur(p):
    try:
        for l in open(p).read().splitlines():
            if l.strip().endswith("*"):
                m = re.search(r"(\d+)Mhz", l)
                return int(m.group(1)) if m else -1
    except OSError:
        return -1
    return -1


stop = False
samples = []


def sampler():
    t0 = time.perf_counter()
    while not stop:
        samples.append((time.perf_counter() - t0, [cur(p) for p in paths]))
        time.sleep(0.002)


th = threading.Thread(target=sampler)
th.start()
steps = sys.argv[1] if len(sys.argv) > 1 else "400"
r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "step_trace.py"), steps],
                   capture_output=True, text=True)
stop = True
th.join()
print(r.stdout[-6000:])
if r.returncode:
    print(r.stderr[-2000:])
if not paths:
    print("no pp_dpm_sclk in sysfs")
    sys.exit(0)
rng = [max(s[1][i] for s in samples) - min(s[1][i] for s in samples) for i in range(len(paths))]
me = rng.index(max(rng))
print(f"{len(paths)} cards in sysfs; card under test: {paths[me]} (range {rng[me]} MHz); sampled every {1e3 * samples[-1][0] / len(samples):.1f} ms")
last = None
for t, s in samples:
    if s[me] != last:
        print(f"{t:8.3f} s  sclk {s[me]} MHz")
        last = s[me]
