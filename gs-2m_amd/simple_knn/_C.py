"""`distCUDA2(points)`: mean squared distance to the 3 nearest neighbours, on the HIP device.

Mirrors /root/reference/submodules/simple-knn/spatial.cu:15-24 (python name in ext.cpp:15-17):
points f32 (P,3) on the device -> f32 (P).  Backed by gs2m_knn_dist2 (include/gs2m_raster.h).
"""
import torch

import gs2m_native as _native


def distCUDA2(points):
    if not points.is_cuda:
        raise RuntimeError("distCUDA2: points must be on a HIP (cuda) device; there is no CPU path")
    pts = points.contiguous().float()
    P = pts.size(0)
    means = torch.full((P,), 0.0, dtype=torch.float32, device=pts.device)
    if P == 0:
        return means
    holder = {}

    def _alloc(nbytes, _user):
        try:
            holder["t"] = torch.empty(int(nbytes), dtype=torch.uint8, device=pts.device)
            return holder["t"].data_ptr()
        except Exception:
            return 0

    cb = _native.ALLOC_FN(_alloc)
    with torch.cuda.device(pts.device):
        rc = _native.lib().gs2m_knn_dist2(P, pts.data_ptr(), means.data_ptr(), cb, None,
                                          torch.cuda.current_stream().cuda_stream)
    _native.check(rc, "gs2m_knn_dist2")
    return means
