"""distCUDA2 (HIP) vs the exhaustive CPU oracle (exact 3-NN mean of squared distances)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(pts):
    from simple_knn._C import distCUDA2
    return distCUDA2(torch.tensor(pts).cuda()).cpu().numpy()


@pytest.mark.parametrize("P,seed", [(5, 0), (100, 1), (1025, 2), (5000, 3), (30000, 4)])
def test_dist_matches_oracle(oracle_lib, P, seed):
    g = torch.Generator().manual_seed(seed)
    pts = (torch.rand(P, 3, generator=g) * torch.tensor([4.0, 2.0, 8.0]) - torch.tensor([2.0, 1.0, 0.0])).numpy()
    d = _run(pts)
    ref = oracle_lib.knn_dist2(pts)
    assert d.shape == (P,) and d.dtype == np.float32
    assert np.array_equal(d, ref), f"max rel {np.abs(d - ref).max() / ref.max():.2e}"


def test_clustered_and_duplicate_points(oracle_lib):
    g = torch.Generator().manual_seed(9)
    centres = torch.randn(40, 3, generator=g) * 5
    pts = (centres[torch.randint(0, 40, (8000,), generator=g)] + 0.01 * torch.randn(8000, 3, generator=g)).numpy()
    pts[100:110] = pts[100]  # coincident points are neighbours at distance 0
    d = _run(pts)
    assert np.array_equal(d, oracle_lib.knn_dist2(pts))
    assert np.all(d[100:110] == 0)


def test_fewer_than_four_points(oracle_lib):
    pts = np.array([[0, 0, 0], [1, 0, 0], [0, 2, 0]], np.float32)
    d = _run(pts)
    ref = oracle_lib.knn_dist2(pts)  # unfilled slots stay FLT_MAX -> inf after the sum, as in the reference
    assert np.array_equal(np.isinf(d), np.isinf(ref)) and np.array_equal(d[np.isfinite(d)], ref[np.isfinite(ref)])
    assert distance_empty()


def distance_empty():
    from simple_knn._C import distCUDA2
    return distCUDA2(torch.zeros(0, 3).cuda()).numel() == 0


def test_scales_init_like_create_from_pcd(oracle_lib):
    """the one call site: scene/gaussian_model.py:190-191, dist2 -> log(sqrt(clamp_min(1e-7)))."""
    from simple_knn._C import distCUDA2
    g = torch.Generator().manual_seed(5)
    pts = torch.rand(4096, 3, generator=g)
    dist2 = torch.clamp_min(distCUDA2(pts.float().cuda()), 0.0000001)
    scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3).cpu()
    ref = np.log(np.sqrt(np.maximum(oracle_lib.knn_dist2(pts.numpy()), 1e-7)))
    assert torch.isfinite(scales).all() and np.allclose(scales[:, 0].numpy(), ref, rtol=1e-6)
