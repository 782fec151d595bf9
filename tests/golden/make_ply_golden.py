"""Writes tests/golden/model_small.ply + model_small.npz: a 5-Gaussian degree-3 model in the byte layout the reference's
GaussianModel.save_ply produces (scene/gaussian_model.py:263-302 through plyfile's PlyData([el]).write: header
`ply / format binary_little_endian 1.0 / element vertex N / property float <name> ... / end_header`, then N packed
little-endian float32 rows in construct_list_of_attributes order, SH tensors channel-major).  Built here with
struct.pack, independently of gs2m_model's writer; plyfile itself is not installed in the build container, so its
header conventions are restated from its documented output, not executed.

    python tests/golden/make_ply_golden.py
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(77)
n, M = 5, 16
xyz = rng.normal(size=(n, 3)).astype(np.float32)
f_dc = rng.normal(size=(n, 1, 3)).astype(np.float32)       # (N, 1, 3) as the model stores it
f_rest = rng.normal(size=(n, M - 1, 3)).astype(np.float32)  # (N, 15, 3)
opacity = rng.normal(size=(n, 1)).astype(np.float32)
scaling = rng.normal(size=(n, 3)).astype(np.float32)
rotation = rng.normal(size=(n, 4)).astype(np.float32)
albedo = rng.normal(size=(n, 3)).astype(np.float32)
roughness = rng.normal(size=(n, 1)).astype(np.float32)
metallic = rng.normal(size=(n, 1)).astype(np.float32)

names = ["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(3 * (M - 1))] \
    + ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)] + [f"albedo_{i}" for i in range(3)] \
    + ["roughness", "metallic"]
header = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n + "".join(f"property float {a}\n" for a in names) + "end_header\n"
body = b""
for i in range(n):
    row = list(xyz[i]) + [0.0, 0.0, 0.0]
    row += [f_dc[i, 0, c] for c in range(3)]                             # transpose(1, 2).flatten: channel-major
    row += [f_rest[i, k, c] for c in range(3) for k in range(M - 1)]
    row += list(opacity[i]) + list(scaling[i]) + list(rotation[i]) + list(albedo[i]) + list(roughness[i]) + list(metallic[i])
    assert len(row) == len(names)
    body += struct.pack("<%df" % len(row), *[float(v) for v in row])
open(os.path.join(HERE, "model_small.ply"), "wb").write(header.encode("ascii") + body)
np.savez(os.path.join(HERE, "model_small.npz"), xyz=xyz, f_dc=f_dc, f_rest=f_rest, opacity=opacity, scaling=scaling,
         rotation=rotation, albedo=albedo, roughness=roughness, metallic=metallic)
print("wrote model_small.ply (%d bytes)" % (len(header) + len(body)))
