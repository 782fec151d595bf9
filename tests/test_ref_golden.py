"""The CPU oracle (oracle/gs2m_oracle.c) against golden vectors of the REFERENCE BUILD (tests/golden/ref_raster_*.npz, written on
the GPU box by tests/golden/make_ref_golden.py from oracle/_ref = the reference's own kernels): the pin of
tests/test_reference_gpu.py::test_oracle_is_pinned_to_the_reference_build, available where there is no GPU.  Integer artefacts
(num_rendered, radii, tiles_touched, offsets, sorted 64-bit keys and instance list incl. depth ties, ranges, observe) identical;
the per-Gaussian forward bit for bit; images to 2e-6; gradients in the two halves of helpers.assert_two_stage."""
import glob
import os

import numpy as np
import pytest

import helpers as Hh

FILES = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_raster_*.npz")))


class _Ref:  # the reference build's outputs with the attribute names of oracle.OracleForward
    pass


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def test_the_vectors_are_there():
    assert len(FILES) >= 3, "tests/golden/ref_raster_*.npz: generate with tests/golden/make_ref_golden.py on the GPU box"


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(p)[11:-4] for p in FILES])
def test_oracle_reproduces_the_reference_builds_vectors(oracle_lib, path):
    z = np.load(path)
    sc = Hh.scene_from_golden(z)
    f, gr = Hh.run_oracle(oracle_lib, sc)
    r = _Ref()
    for k in z.files:
        if k.startswith("ref_"):
            setattr(r, k[4:], z[k])
    rg = {k[8:]: z[k] for k in z.files if k.startswith("refgrad_")}
    assert f.num_rendered == int(z["num_rendered"]) > 0
    for k in ("radii", "tiles_touched", "point_offsets", "keys_sorted", "vals_sorted", "ranges", "observe"):
        assert np.array_equal(getattr(f, k), getattr(r, k)), k
    vis = r.radii > 0
    assert np.array_equal(f.clamped[vis], r.clamped[vis]), "clamped"
    for k in ("depths", "means2D", "conic_opacity", "cov3D", "rgb"):  # the whole per-Gaussian forward, bit for bit
        assert np.array_equal(_bits(getattr(f, k)[vis]), _bits(getattr(r, k)[vis])), k
    if "depth_ties" in path:  # the scene is there for this: equal depths inside a tile list, ordered by Gaussian id
        keys = r.keys_sorted
        same = keys[1:] == keys[:-1]
        assert same.sum() > 50 and np.all(np.diff(r.vals_sorted.astype(np.int64))[same] > 0)
    # blending: the device's expf against the host's, float atomics against double sums
    assert (f.n_contrib != r.n_contrib).mean() <= 1e-4
    assert np.abs(f.final_T - r.final_T).max() <= 5e-7
    assert np.abs(f.color - r.color).max() <= 2e-6
    for ch in range(10):
        assert np.abs(f.buffer[ch] - r.buffer[ch]).max() <= 2e-6 * max(1.0, float(np.abs(r.buffer[ch]).max())), f"buffer[{ch}]"
    # gradients: (A) the per-Gaussian sums the reference accumulates with atomicAdd, element-wise at 1e-3; (B) the reference's
    # cov2D / projection / SH / cov3D chain against the oracle's chain evaluated on the reference's own sums: 1e-5, no exceptions
    Hh.assert_two_stage(oracle_lib, f, gr, rg)
