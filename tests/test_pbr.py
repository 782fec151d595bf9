"""CPU: the pure-PyTorch helpers of `pbr/` against outputs of the reference's own functions, and the generated
environment-BRDF table against a sub-sample of the table the reference ships (tests/golden/ref_pbr.npz, written by
tests/golden/make_golden.py in the build container)."""
import os
import types

import numpy as np
import torch

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_pbr.npz"))
T = lambda k: torch.from_numpy(G[k])


def test_shading_helpers_reproduce_the_reference_functions():
    from pbr import shade, light
    close = lambda a, k, tol=1e-6: np.testing.assert_allclose(a.numpy() if isinstance(a, torch.Tensor) else a, G[k], rtol=tol, atol=tol)
    close(shade.saturate_dot(T("dot_a"), T("dot_b")), "saturate_dot")
    x = T("tone_x")
    close(shade.aces_film(x), "aces_film")
    close(shade.aces_film(x.numpy()), "aces_film_np")
    close(shade.linear_to_srgb(x), "linear_to_srgb")
    close(shade.linear_to_srgb(x.numpy()), "linear_to_srgb_np")
    close(shade.srgb_to_linear(x), "srgb_to_linear")
    close(shade.srgb_to_linear(x.numpy()), "srgb_to_linear_np")
    x3 = x.reshape(1, 1, -1, 3).expand(1, 2, -1, 3).contiguous()
    close(shade.rgb_to_srgb(x3), "rgb_to_srgb")
    close(shade.srgb_to_rgb(x3), "srgb_to_rgb")
    x4 = torch.cat([x3, torch.full_like(x3[..., :1], 0.37)], dim=-1)
    assert torch.equal(shade.rgb_to_srgb(x4)[..., 3], x4[..., 3]) and torch.equal(shade.rgb_to_srgb(x4)[..., :3], shade.rgb_to_srgb(x3))
    close(shade.envBRDF_approx(T("env_rough"), T("env_nov")), "envBRDF_approx")
    gx, gy = T("cube_x"), T("cube_y")
    for s in range(6):
        assert np.array_equal(light.cube_to_dir(s, gx, gy).numpy(), G["cube_to_dir"][s])
    close(light.cubemap_mip.forward(types.SimpleNamespace(), T("mip_in")), "mip_out")
    assert [light.CubemapLight.LIGHT_MIN_RES, light.CubemapLight.MIN_ROUGHNESS, light.CubemapLight.MAX_ROUGHNESS] == list(G["light_consts"])
    for levels in (7, 4):
        fake = types.SimpleNamespace(MIN_ROUGHNESS=light.CubemapLight.MIN_ROUGHNESS, MAX_ROUGHNESS=light.CubemapLight.MAX_ROUGHNESS, specular=[None] * levels)
        close(light.CubemapLight.get_mip(fake, T("mip_rough")), "get_mip_%d" % levels)


def test_generated_brdf_table_is_the_reference_table_up_to_sampling_noise():
    """tools/make_brdf_lut.py (16384 samples per texel) against the shipped table: the residual is the SHIPPED table's own
    sampling noise (its second differences are as rough as a 1024-sample estimate); 3.2e-3 max, 1.3e-4 mean."""
    from pbr import get_brdf_lut
    lut = get_brdf_lut()
    assert tuple(lut.shape) == (1, 256, 256, 2) and lut.dtype == torch.float32
    idx = G["brdf_idx"]
    d = np.abs(lut[0].numpy()[np.ix_(idx, idx)] - G["brdf_sub"])
    assert d.max() < 3.5e-3 and d.mean() < 2e-4, (d.max(), d.mean())
