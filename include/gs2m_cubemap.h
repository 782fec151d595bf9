/* gs2m_cubemap.h -- C ABI of the environment-map prefilters of the deferred PBR stage, part of
 * libgs2m_raster.so.
 *
 * Replaces the reference's CUDA-only dependency submodules/render-utils for the two operators CubemapLight.build_mips
 * (pbr/light.py:86-99) runs for every training view:
 *     render_utils.diffuse_cubemap(cubemap)                       render_utils/ops.py:336-356, c_src/cubemap.cu:110-172
 *     render_utils.specular_cubemap(cubemap, roughness, cutoff)   render_utils/ops.py:358-403, c_src/cubemap.cu:174-350
 * Device pointers, fp32, channels-last cube maps (6, res, res, C).  Both backward passes are deterministic gathers (the
 * reference scatters with atomics).  The specular operator needs no bounds table: the acceleration boxes are computed
 * per thread and do not affect the result.  Asynchronous on `stream`; return GS2M_OK (0) or a negative GS2M_ERR_*. */
#ifndef GS2M_CUBEMAP_H
#define GS2M_CUBEMAP_H

#ifdef __cplusplus
extern "C" {
#endif

/* cubemap, out, dL_dout, dL_dcubemap: (6, res, res, 3).  Cosine-weighted sum over the whole sphere. */
int gs2m_diffuse_cubemap_forward(int res, const float* cubemap, float* out, void* stream);
int gs2m_diffuse_cubemap_backward(int res, const float* dL_dout, float* dL_dcubemap, void* stream);

/* Per-level table of the separable texel-area factors, `res` floats (area(x, y) = table[x] * table[y]): fill once per
 * resolution and keep (it depends on nothing else).  Needed by the specular operator. */
int gs2m_cubemap_texel_table(int res, float* table, void* stream);

/* cubemap, dL_dcubemap: (6, res, res, 3); out, dL_dout: (6, res, res, 4) = (weighted colour sum, weight sum) -- the
 * caller divides, as render_utils/ops.py:403 does.  GGX lobe of alpha = roughness^2 restricted to directions with
 * cos(angle) >= costheta_cutoff. */
int gs2m_specular_cubemap_forward(int res, float roughness, float costheta_cutoff, const float* texel_table, const float* cubemap,
                                  float* out, void* stream);
int gs2m_specular_cubemap_backward(int res, float roughness, float costheta_cutoff, const float* texel_table, const float* dL_dout,
                                   float* dL_dcubemap, void* stream);

/* The same operator INCLUDING the division by the weight sum (what `specular_cubemap` returns, render_utils/ops.py:391-403) and
 * its backward: out, dL_dout (6, res, res, 3); raw (6, res, res, 4), 16-byte aligned, is written by the forward and must be
 * handed to the backward unchanged; scratch: (6, res, res, 4) floats, 16-byte aligned. */
int gs2m_specular_cubemap_normalized_forward(int res, float roughness, float costheta_cutoff, const float* texel_table,
                                             const float* cubemap, float* raw, float* out, void* stream);
int gs2m_specular_cubemap_normalized_backward(int res, float roughness, float costheta_cutoff, const float* texel_table,
                                              const float* raw, const float* dL_dout, float* scratch, float* dL_dcubemap, void* stream);

#ifdef __cplusplus
}
#endif
#endif
