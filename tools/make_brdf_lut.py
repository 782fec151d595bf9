"""Generates gs-2m_amd/pbr/brdf_256_256.bin: the split-sum environment BRDF table (Karis 2013, "Real Shading in Unreal
Engine 4") the deferred shading looks up with (N.V, roughness) -- pbr/shade.py:120-127 loads a table of this layout
((1, 256, 256, 2) float32, row = roughness, column = N.V, channels = scale and bias of F0).

For every (N.V, roughness) texel centre: importance-sample the GGX lobe (alpha = roughness^2) with a Hammersley
sequence, weight by the height-correlated Smith-GGX visibility (Heitz 2014), and accumulate (1 - Fc) G_vis and Fc G_vis
with Fc = (1 - V.H)^5.  (Checked in the build container against the table the reference ships, pbr/brdf_256_256.bin: this
convention -- texel centres, alpha = roughness^2, correlated visibility -- reproduces it to 1.2e-4 mean / 3e-3 max, the shipped table's own sampling noise; the separable
k = alpha / 2 form does not.)

    python tools/make_brdf_lut.py [--samples 16384]      (5 minutes)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def radical_inverse_vdc(i):
    bits = i.astype(np.uint32)
    bits = (bits << 16) | (bits >> 16)
    bits = ((bits & 0x55555555) << 1) | ((bits & 0xAAAAAAAA) >> 1)
    bits = ((bits & 0x33333333) << 2) | ((bits & 0xCCCCCCCC) >> 2)
    bits = ((bits & 0x0F0F0F0F) << 4) | ((bits & 0xF0F0F0F0) >> 4)
    bits = ((bits & 0x00FF00FF) << 8) | ((bits & 0xFF00FF00) >> 8)
    return bits.astype(np.float64) * 2.3283064365386963e-10


def integrate(nov, roughness, samples=1024):
    """nov, roughness: arrays of the same shape -> (A, B)."""
    i = np.arange(samples)
    u1 = (i + 0.0) / samples
    u2 = radical_inverse_vdc(i)
    shape = nov.shape
    nov = nov.reshape(-1, 1)
    a = (roughness.reshape(-1, 1)) ** 2
    V = np.stack([np.sqrt(1.0 - nov * nov), np.zeros_like(nov), nov], axis=-1)  # (n, 1, 3), N = +z
    phi = 2.0 * np.pi * u1[None, :]
    cos_t = np.sqrt((1.0 - u2[None, :]) / (1.0 + (a * a - 1.0) * u2[None, :]))
    sin_t = np.sqrt(np.maximum(1.0 - cos_t * cos_t, 0.0))
    H = np.stack([sin_t * np.cos(phi), sin_t * np.sin(phi), cos_t], axis=-1)      # (n, s, 3)
    VoH = np.sum(V * H, axis=-1)
    L = 2.0 * VoH[..., None] * H - V
    NoL = np.clip(L[..., 2], 0.0, 1.0)
    NoH = np.clip(H[..., 2], 0.0, 1.0)
    VoH = np.clip(VoH, 0.0, 1.0)
    a2 = a * a
    vis = 0.5 / np.maximum(NoL * np.sqrt(nov * nov * (1.0 - a2) + a2) + nov * np.sqrt(NoL * NoL * (1.0 - a2) + a2), 1e-12)
    G_vis = np.where(NoL > 0, 4.0 * vis * VoH * NoL / np.maximum(NoH, 1e-12), 0.0)
    Fc = (1.0 - VoH) ** 5
    A = ((1.0 - Fc) * G_vis).mean(axis=1)
    B = (Fc * G_vis).mean(axis=1)
    return A.reshape(shape), B.reshape(shape)


def make(res=256, samples=1024):
    c = (np.arange(res) + 0.5) / res
    out = np.zeros((res, res, 2), dtype=np.float32)
    for y in range(res):   # row = roughness
        A, B = integrate(c, np.full(res, c[y]), samples)
        out[y, :, 0], out[y, :, 1] = A, B
    return out


if __name__ == "__main__":
    n = int(sys.argv[sys.argv.index("--samples") + 1]) if "--samples" in sys.argv else 16384
    lut = make(256, n)
    path = os.path.join(ROOT, "gs-2m_amd", "pbr", "brdf_256_256.bin")
    lut.tofile(path)
    print("wrote", path, lut.shape, lut.dtype, "A in [%.4f, %.4f], B in [%.4f, %.4f]" % (lut[..., 0].min(), lut[..., 0].max(), lut[..., 1].min(), lut[..., 1].max()))
