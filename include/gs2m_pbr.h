/* gs2m_pbr.h -- C ABI of the fused deferred shading, part of libgs2m_raster.so (csrc/texture.hip).
 *
 * `pbr_shading` (pbr/shade.py:130-213) as one kernel each way, for the configuration `pbr_render` (pbr/__init__.py:9-56) uses
 * (occlusion = 1, no tone mapping; gamma is applied by the caller):
 *     r   = 2 max(n.v, 0) n - v
 *     E   = texture(diffuse (6,wd,wd,3), n, 'linear', 'cube')
 *     A,B = texture(brdf_lut (H,W,2), (clamp(n.v, 1e-4, 1), roughness), 'linear', 'clamp')
 *     L   = texture(specular stack, r, mip_level_bias = CubemapLight.get_mip(roughness), 'linear-mipmap-linear', 'cube')
 *     rgb = clamp(E albedo + L (F0 A + B), 0, 1),  F0 = 0.04 (1 - metallic) + albedo metallic   (0.04 if metallic is NULL)
 * The reference runs ~30 PyTorch launches forward and ~50 backward around three `dr.texture` calls for this.
 * Backward: gradients to albedo, metallic (optional) and -- accumulated, zero them first -- the diffuse map and every level
 * of the specular stack; normals, view directions and roughness get none (pbr_render detaches them).  `image_width`: the pixels
 * are an image of that width in row-major order (n a multiple of it) -- lets the backward work on 2-D pixel tiles -- or 0.
 * Device pointers, fp32; pixel arrays are (n, 3) / (n, 1); `specular`, `dL_dspecular`, `width` are HOST arrays of `levels`
 * entries.  Asynchronous on `stream`; return GS2M_OK (0) or a negative GS2M_ERR_* code (gs2m_raster.h). */
#ifndef GS2M_PBR_H
#define GS2M_PBR_H

#ifdef __cplusplus
extern "C" {
#endif

/* diffuse_rgb, specular_rgb, diffuse_light: optional extra outputs (NULL to skip), the entries of pbr_shading's result dict. */
int gs2m_pbr_shade_forward(int n, const float* normals, const float* view_dirs, const float* albedo, const float* roughness,
                           const float* metallic, const float* brdf_lut, int lut_width, int lut_height, const float* diffuse,
                           int diffuse_width, int levels, const float* const* specular, const int* width, float min_roughness,
                           float max_roughness, float* render_rgb, float* diffuse_rgb, float* specular_rgb, float* diffuse_light,
                           void* stream);
int gs2m_pbr_shade_backward(int n, const float* normals, const float* view_dirs, const float* albedo, const float* roughness,
                            const float* metallic, const float* brdf_lut, int lut_width, int lut_height, const float* diffuse,
                            int diffuse_width, int levels, const float* const* specular, const int* width, float min_roughness,
                            float max_roughness, const float* dL_drender_rgb, float* dL_dalbedo, float* dL_dmetallic,
                            float* dL_ddiffuse, float* const* dL_dspecular, int image_width, void* stream);

/* The shading inputs from the planar G-buffer maps (pbr_render, pbr/__init__.py:25-43), one launch: normals (H, W, 3) =
 * normal_map (3, H, W) re-normalised where non-zero; albedo (H, W, 3) = clamp(albedo_map, 0, 1); roughness (H, W) =
 * roughness_map * (max - min) + min; metallic (H, W) = metallic_map, or alpha_map * clamp(1 - roughness_map, 0, 1) when
 * metallic_map is NULL.  Backward: only the albedo carries a gradient (through the clamp; the reference detaches the rest,
 * and a learnt metallic map passes its gradient through unchanged): dL_dalbedo (H, W, 3) -> dL_dalbedo_map (3, H, W). */
int gs2m_pbr_inputs_forward(int width, int height, const float* normal_map, const float* albedo_map, const float* roughness_map,
                            const float* alpha_map, const float* metallic_map, float min_roughness, float max_roughness,
                            float* normals, float* albedo, float* roughness, float* metallic, void* stream);
int gs2m_pbr_inputs_backward(int width, int height, const float* albedo_map, const float* dL_dalbedo, float* dL_dalbedo_map, void* stream);

#ifdef __cplusplus
}
#endif
#endif
