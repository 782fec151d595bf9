/* gs2m_texture.h -- C ABI of the texture lookups of the deferred PBR stage, part of
 * libgs2m_raster.so.
 *
 * Replaces, for the three modes the reference uses, `nvdiffrast.torch.texture`
 * (submodules/nvdiffrast/nvdiffrast/torch/ops.py:427; kernels in nvdiffrast/common/textureCUDA.cu) at its call sites
 *     pbr/shade.py:150-154   cube map, filter 'linear', boundary 'cube'                  (diffuse irradiance by normal)
 *     pbr/shade.py:165-169   2-D, filter 'linear', boundary 'clamp'                      (BRDF LUT)
 *     pbr/shade.py:173-179   cube map, 'linear-mipmap-linear', explicit `mip=` stack, per-pixel `mip_level_bias`,
 *                            no `uv_da` (the level is the clamped bias)                  (specular, by roughness)
 *     pbr/light.py:43-48, 111-115   cube map 'linear' (cubemap_mip backward, environment export)
 * with nvdiffrast's semantics: face selection and face coordinates, texel centres at (i + 0.5) / w, footprints that
 * continue across cube edges, the corner texel that is the average of the other three, zero output for non-finite
 * directions.  Backward: gradient with respect to the texture only (each mip level its own tensor); the reference
 * detaches the directions and the roughness that feed these lookups.
 *
 * All pointers are DEVICE pointers, fp32; textures are channels-last: cube level l is (6, width[l], width[l], C),
 * 2-D is (height, width, C); 1 <= C <= 4.  `tex`, `grad_tex`, `width` are HOST arrays of `levels` entries
 * (read before the call returns).  Gradient tensors are ACCUMULATED into (atomic adds): zero them first.
 * Asynchronous on `stream`; return GS2M_OK (0) or a negative GS2M_ERR_* code (gs2m_raster.h). */
#ifndef GS2M_TEXTURE_H
#define GS2M_TEXTURE_H

#ifdef __cplusplus
extern "C" {
#endif

#define GS2M_TEX_MAX_LEVELS 12

/* dirs: (n, 3) lookup directions (any length); mip_level_bias: (n) or NULL.  NULL: 'linear' on level 0 (levels may
 * be 1).  Non-NULL: 'linear-mipmap-linear' with level = clamp(bias, 0, levels - 1).  out: (n, C). */
int gs2m_texture_cube_forward(int n, int channels, int levels, const float* const* tex, const int* width, const float* dirs,
                              const float* mip_level_bias, float* out, void* stream);
/* image_width: the n lookups are an image of that width in row-major order (n a multiple of it; lets the backward work on 2-D
 * pixel tiles, whose footprints overlap), or 0. */
int gs2m_texture_cube_backward(int n, int channels, int levels, float* const* grad_tex, const int* width, const float* dirs,
                               const float* mip_level_bias, const float* dL_dout, int image_width, void* stream);

/* uv: (n, 2) in texture units ([0, 1] spans the texture); coordinates clamp to the centres of the edge texels. */
int gs2m_texture_2d_clamp_forward(int n, int channels, int width, int height, const float* tex, const float* uv, float* out,
                                  void* stream);
int gs2m_texture_2d_clamp_backward(int n, int channels, int width, int height, float* grad_tex, const float* uv,
                                   const float* dL_dout, void* stream);

#ifdef __cplusplus
}
#endif
#endif
