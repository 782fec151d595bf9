"""COLMAP sparse-model I/O for the loaders row: the binary `cameras.bin`, `images.bin`,
`points3D.bin` of a COLMAP reconstruction -- what scene/colmap_loader.py:123-240 reads and
scene/dataset_readers.py:49-108, 141-197 turn into cameras, an initial point cloud and the scene radius.

Own implementation on `struct` / numpy (the per-image 2-D observations and the per-point tracks, which dominate file
size, are read as numpy blocks rather than value by value); little-endian layouts as COLMAP's
Reconstruction::Write*Binary writes them.  A writer is included so that synthetic scenes can be stored in the format
(tests, `gs2m_train` datasets).  Names follow the reference (`Camera`, `Image`, `qvec2rotmat`, `read_*_binary`).
"""
import collections
import os
import struct

import numpy as np

Camera = collections.namedtuple("Camera", ["id", "model", "width", "height", "params"])
Image = collections.namedtuple("Image", ["id", "qvec", "tvec", "camera_id", "name", "xys", "point3D_ids"])

# COLMAP camera models: id -> (name, number of parameters)
CAMERA_MODELS = {0: ("SIMPLE_PINHOLE", 3), 1: ("PINHOLE", 4), 2: ("SIMPLE_RADIAL", 4), 3: ("RADIAL", 5), 4: ("OPENCV", 8),
                 5: ("OPENCV_FISHEYE", 8), 6: ("FULL_OPENCV", 12), 7: ("FOV", 5), 8: ("SIMPLE_RADIAL_FISHEYE", 4),
                 9: ("RADIAL_FISHEYE", 5), 10: ("THIN_PRISM_FISHEYE", 12)}
_MODEL_IDS = {name: (mid, n) for mid, (name, n) in CAMERA_MODELS.items()}


def qvec2rotmat(q):
    """(w, x, y, z) -> 3x3 rotation (world to camera in COLMAP's convention)."""
    w, x, y, z = q
    return np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * w * z, 2 * z * x + 2 * w * y],
                     [2 * x * y + 2 * w * z, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * w * x],
                     [2 * z * x - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * x * x - 2 * y * y]])


def rotmat2qvec(R):
    """3x3 rotation -> (w, x, y, z), w >= 0 (largest-eigenvector form, robust for any rotation)."""
    Rxx, Ryx, Rzx, Rxy, Ryy, Rzy, Rxz, Ryz, Rzz = np.asarray(R, dtype=np.float64).flat
    K = np.array([[Rxx - Ryy - Rzz, 0, 0, 0], [Ryx + Rxy, Ryy - Rxx - Rzz, 0, 0], [Rzx + Rxz, Rzy + Ryz, Rzz - Rxx - Ryy, 0],
                  [Ryz - Rzy, Rzx - Rxz, Rxy - Ryx, Rxx + Ryy + Rzz]]) / 3.0
    vals, vecs = np.linalg.eigh(K)
    q = vecs[[3, 0, 1, 2], np.argmax(vals)]
    return -q if q[0] < 0 else q


def read_intrinsics_binary(path):
    cams = {}
    with open(path, "rb") as f:
        (n,) = struct.unpack("<Q", f.read(8))
        for _ in range(n):
            cid, mid, w, h = struct.unpack("<iiQQ", f.read(24))
            name, npar = CAMERA_MODELS[mid]
            cams[cid] = Camera(cid, name, w, h, np.frombuffer(f.read(8 * npar), dtype="<f8").copy())
    return cams


def read_extrinsics_binary(path):
    images = {}
    with open(path, "rb") as f:
        data = f.read()
    (n,) = struct.unpack_from("<Q", data, 0)
    off = 8
    obs = np.dtype([("x", "<f8"), ("y", "<f8"), ("id", "<i8")])
    for _ in range(n):
        iid, qw, qx, qy, qz, tx, ty, tz, cid = struct.unpack_from("<idddddddi", data, off)
        off += 64
        end = data.index(b"\x00", off)
        name = data[off:end].decode("utf-8")
        off = end + 1
        (m,) = struct.unpack_from("<Q", data, off)
        off += 8
        block = np.frombuffer(data, dtype=obs, count=m, offset=off)
        off += 24 * m
        images[iid] = Image(iid, np.array([qw, qx, qy, qz]), np.array([tx, ty, tz]), cid, name,
                            np.column_stack([block["x"], block["y"]]).astype(np.float64).reshape(m, 2), block["id"].astype(np.int64))
    return images


def read_points3D_binary(path):
    """-> xyz (n, 3) float64, rgb (n, 3) float64 in 0..255, error (n, 1) float64 (the tracks are skipped)."""
    with open(path, "rb") as f:
        data = f.read()
    (n,) = struct.unpack_from("<Q", data, 0)
    xyz, rgb, err = np.empty((n, 3)), np.empty((n, 3)), np.empty((n, 1))
    off = 8
    for i in range(n):
        _, x, y, z, r, g, b, e = struct.unpack_from("<QdddBBBd", data, off)
        (tl,) = struct.unpack_from("<Q", data, off + 43)
        off += 51 + 8 * tl
        xyz[i], rgb[i], err[i] = (x, y, z), (r, g, b), e
    return xyz, rgb, err


def write_model(folder, cameras, images, points_xyz, points_rgb, points_error=None):
    """cameras: iterable of Camera; images: iterable of Image (xys / point3D_ids may be empty); points: (n,3) arrays."""
    os.makedirs(folder, exist_ok=True)
    with open(os.path.join(folder, "cameras.bin"), "wb") as f:
        cameras = list(cameras)
        f.write(struct.pack("<Q", len(cameras)))
        for c in cameras:
            mid, npar = _MODEL_IDS[c.model]
            assert len(c.params) == npar
            f.write(struct.pack("<iiQQ", c.id, mid, c.width, c.height) + np.asarray(c.params, dtype="<f8").tobytes())
    with open(os.path.join(folder, "images.bin"), "wb") as f:
        images = list(images)
        f.write(struct.pack("<Q", len(images)))
        for im in images:
            f.write(struct.pack("<idddddddi", im.id, *[float(v) for v in im.qvec], *[float(v) for v in im.tvec], im.camera_id))
            f.write(im.name.encode("utf-8") + b"\x00")
            m = 0 if im.xys is None else len(im.xys)
            f.write(struct.pack("<Q", m))
            if m:
                rec = np.empty(m, dtype=[("x", "<f8"), ("y", "<f8"), ("id", "<i8")])
                rec["x"], rec["y"], rec["id"] = np.asarray(im.xys)[:, 0], np.asarray(im.xys)[:, 1], np.asarray(im.point3D_ids)
                f.write(rec.tobytes())
    with open(os.path.join(folder, "points3D.bin"), "wb") as f:
        n = len(points_xyz)
        f.write(struct.pack("<Q", n))
        err = np.zeros(n) if points_error is None else np.asarray(points_error).reshape(-1)
        for i in range(n):
            r, g, b = (int(v) for v in points_rgb[i])
            f.write(struct.pack("<QdddBBBd", i + 1, *[float(v) for v in points_xyz[i]], r, g, b, float(err[i])))
            f.write(struct.pack("<Q", 0))


CameraInfo = collections.namedtuple("CameraInfo", ["uid", "R", "T", "Fx", "Fy", "image_name", "width", "height"])


def colmap_cameras(extrinsics, intrinsics):
    """readColmapCameras (scene/dataset_readers.py:72-108) without the image files: R is stored transposed (the
    rasterizer's convention), SIMPLE_PINHOLE / PINHOLE only."""
    out = []
    for key in extrinsics:
        e = extrinsics[key]
        c = intrinsics[e.camera_id]
        if c.model == "SIMPLE_PINHOLE":
            fx = fy = c.params[0]
        elif c.model == "PINHOLE":
            fx, fy = c.params[0], c.params[1]
        else:
            raise ValueError(f"Unsupported COLMAP camera model {c.model}: only undistorted (SIMPLE_)PINHOLE datasets are supported")
        out.append(CameraInfo(c.id, np.transpose(qvec2rotmat(e.qvec)), np.array(e.tvec), fx, fy, e.name, c.width, c.height))
    return out


def nerf_normalization(cam_infos):
    """getNerfppNorm (scene/dataset_readers.py:49-70): translate = -mean camera centre, radius = 1.1 x the largest
    distance of a camera centre from that mean (the `cameras_extent` densification thresholds are scaled by)."""
    from gs2m_synth import world2view   # getWorld2View2: returns float32, and the reference inverts THAT
    centres = np.stack([np.linalg.inv(world2view(c.R, c.T))[:3, 3] for c in cam_infos], axis=1)
    centre = centres.mean(axis=1, keepdims=True)
    return {"translate": -centre.flatten(), "radius": float(np.linalg.norm(centres - centre, axis=0).max() * 1.1)}
