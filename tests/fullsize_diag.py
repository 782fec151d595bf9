"""diagnosis of one full-size comparison: the elements of a gradient tensor outside 1e-3 relative (+ floor), HIP path / reference build / CPU oracle
usage: python tests/fullsize_diag.py P fc [refbin] [tensor]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "gs-2m_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import helpers as Hh
import gs2m_native
from oracle import oracle, reference
oracle.use_native_build()
os.environ.setdefault("OMP_NUM_THREADS", str(os.cpu_count()))
P, fc = int(sys.argv[1]), int(sys.argv[2])
refbin = len(sys.argv) > 3 and sys.argv[3] == "1"
names = sys.argv[4].split(",") if len(sys.argv) > 4 else ["shs"]
sc = Hh.make_scene(P, 1920, 1080, seed=0, fc=fc)
r, rg = Hh.run_oracle(reference, sc)
o, og = Hh.run_oracle(oracle, sc)
gs2m_native.set_reference_binning(refbin)
out, g = Hh.run_hip(sc)
sums = Hh.run_hip_sums(sc)
gs2m_native.set_reference_binning(False)
for k in names:
    a, b, c = (np.asarray(x[k], np.float64).reshape(len(x[k]), -1) for x in (g, rg, og))
    nz = b[b != 0]
    floor = 1e-5 * np.sqrt(np.mean(nz * nz))
    bad = np.abs(a - b) > 1e-3 * np.abs(b) + floor
    rows = np.nonzero(bad.any(1))[0]
    print(f"== {k}: floor {floor:.3e}; rows with an element outside: {rows.tolist()}")
    for gid in rows[:8]:
        cols = np.nonzero(bad[gid])[0]
        print(f" gid {gid}: columns {cols.tolist()}")
        for cidx in cols[:6]:
            print(f"   [{cidx}] hip {a[gid, cidx]:.9e} reference {b[gid, cidx]:.9e} oracle {c[gid, cidx]:.9e} | hip-ref {a[gid, cidx] - b[gid, cidx]:.3e} hip-oracle {a[gid, cidx] - c[gid, cidx]:.3e} ref-oracle {b[gid, cidx] - c[gid, cidx]:.3e} bound {1e-3 * abs(b[gid, cidx]) + floor:.3e}")
        for kk in ("colors", "opacities"):
            s_, rr, oo = np.asarray(sums[kk][gid], np.float64).reshape(-1), np.asarray(rg[kk][gid], np.float64).reshape(-1), np.asarray(og[kk][gid], np.float64).reshape(-1)
            print(f"   sum:{kk} hip {s_} reference {rr} oracle {oo}")
        print("   tiles touched", int(r.tiles_touched[gid]), "radius", int(r.radii[gid]), "closest threshold event", Hh.observe_event(o, int(gid), observe=False, band=1e-7))
