// gi_texture.h -- the texture runtime shared by the shade stage (material inputs, dome light) and by traversal (textured cutout
// opacity in the any-hit test).  Included by gi_traversal.h and gi_shading.h.
#pragma once

#include "gi_queues.h"

namespace gi {

// ------------------------------------------------------------------------------------------------
// Texture runtime (mdl_interface.glsl:8-38 apply_wrap_and_crop, :127-145 tex_lookup_float4_2d) over a software sampler:
// bilinear, REPEAT addressing, LOD 0 (the reference's single sampler, Gi.cpp:388-392, CgpuVk.cpp:1985-1990).
// Operation order == oracle sample_bilinear_repeat / tex_lookup_float4_2d.
// ------------------------------------------------------------------------------------------------
__device__ inline F4 sample_bilinear_repeat(const TextureRec& t, float u, float v)
{
  u = u - floorf(u); v = v - floorf(v);
  const float x = u * (float)t.width - 0.5f, y = v * (float)t.height - 0.5f;
  const float x0f = floorf(x), y0f = floorf(y);
  const float fx = x - x0f, fy = y - y0f;
  const int w = (int)t.width, h = (int)t.height;
  int ix0 = (int)x0f, iy0 = (int)y0f;
  if (ix0 < 0) ix0 += w;
  if (iy0 < 0) iy0 += h;
  int ix1 = ix0 + 1; if (ix1 >= w) ix1 -= w;
  int iy1 = iy0 + 1; if (iy1 >= h) iy1 -= h;
  const F4* tx = reinterpret_cast<const F4*>(t.texels);
  const F4 t00 = ld4(&tx[(size_t)iy0 * w + ix0]), t10 = ld4(&tx[(size_t)iy0 * w + ix1]);
  const F4 t01 = ld4(&tx[(size_t)iy1 * w + ix0]), t11 = ld4(&tx[(size_t)iy1 * w + ix1]);
  const float gx = 1.0f - fx, gy = 1.0f - fy;
  F4 o;
  o.x = (t00.x * gx + t10.x * fx) * gy + (t01.x * gx + t11.x * fx) * fy;
  o.y = (t00.y * gx + t10.y * fx) * gy + (t01.y * gx + t11.y * fx) * fy;
  o.z = (t00.z * gx + t10.z * fx) * gy + (t01.z * gx + t11.z * fx) * fy;
  o.w = (t00.w * gx + t10.w * fx) * gy + (t01.w * gx + t11.w * fx) * fy;
  return o;
}
__device__ __forceinline__ float apply_wrap_and_crop(float coord, uint32_t wrap, uint32_t res) // crop = (0, 1)
{
  if (wrap == TEX_WRAP_REPEAT) coord = coord - floorf(coord);
  else {
    if (wrap == TEX_WRAP_MIRRORED_REPEAT) {
      const float tmp = floorf(coord);
      if (((int)tmp & 1) != 0) coord = 1.0f - (coord - tmp); else coord = coord - tmp;
    }
    const float inv_hdim = 0.5f / (float)res;
    coord = fmin2(fmax2(coord, inv_hdim), 1.0f - inv_hdim);
  }
  return coord;
}
__device__ inline F4 tex_lookup_float4_2d(const TextureRec& t, float u, float v, uint32_t wrapU, uint32_t wrapV)
{
  if ((wrapU == TEX_WRAP_CLIP && (u < 0.0f || u > 1.0f)) || (wrapV == TEX_WRAP_CLIP && (v < 0.0f || v > 1.0f))) return F4{0.0f, 0.0f, 0.0f, 0.0f};
  u = apply_wrap_and_crop(u, wrapU, t.width);
  v = apply_wrap_and_crop(v, wrapV, t.height);
  return sample_bilinear_repeat(t, u, v);
}
// ------------------------------------------------------------------------------------------------
// The remaining texture entry points of the MDL renderer runtime (mdl_interface.glsl:45-65 tex_lookup_float4_3d, :86-105 tex_texel_float4_3d, :167-186
// tex_texel_float4_2d, :208-221 tex_resolution_2d).  Only MDL-generated code calls them -- none of the closed forms does -- so in this library they are
// reachable through giCDebugTexRuntime only; they complete the runtime and are held to the oracle bit for bit.  `valid` false is the
// reference's texture id 0 (the invalid texture).  The 3-D sampler is the 2-D one's trilinear extension (same deviation D5: software filter weights).
// ------------------------------------------------------------------------------------------------
struct TextureRec3 { const float* texels; uint32_t width, height, depth; }; // width x height x depth RGBA texels, slice by slice
__device__ inline F4 tex_texel_float4_2d(const TextureRec& t, bool valid, int x, int y)
{
  if (!valid || x < 0 || x >= (int)t.width || y < 0 || y >= (int)t.height) return F4{0.0f, 0.0f, 0.0f, 0.0f};
  return ld4(&reinterpret_cast<const F4*>(t.texels)[(size_t)y * t.width + (size_t)x]);
}
__device__ __forceinline__ void tex_resolution_2d(const TextureRec& t, bool valid, int& w, int& h)
{ w = valid ? (int)t.width : 0; h = valid ? (int)t.height : 0; }
__device__ inline F4 sample_trilinear_repeat(const TextureRec3& t, float u, float v, float w)
{
  w = w - floorf(w);
  const float z = w * (float)t.depth - 0.5f, z0f = floorf(z), fz = z - z0f, gz = 1.0f - fz;
  const int d = (int)t.depth;
  int iz0 = (int)z0f;
  if (iz0 < 0) iz0 += d;
  int iz1 = iz0 + 1; if (iz1 >= d) iz1 -= d;
  const size_t slice = (size_t)t.width * t.height * 4u;
  const TextureRec s0{t.texels + slice * (size_t)iz0, t.width, t.height}, s1{t.texels + slice * (size_t)iz1, t.width, t.height};
  const F4 a = sample_bilinear_repeat(s0, u, v), b = sample_bilinear_repeat(s1, u, v);
  return F4{a.x * gz + b.x * fz, a.y * gz + b.y * fz, a.z * gz + b.z * fz, a.w * gz + b.w * fz};
}
__device__ inline F4 tex_lookup_float4_3d(const TextureRec3& t, bool valid, float u, float v, float w, uint32_t wrapU, uint32_t wrapV, uint32_t wrapW)
{
  if (!valid || (wrapU == TEX_WRAP_CLIP && (u < 0.0f || u > 1.0f)) || (wrapV == TEX_WRAP_CLIP && (v < 0.0f || v > 1.0f)) ||
      (wrapW == TEX_WRAP_CLIP && (w < 0.0f || w > 1.0f))) return F4{0.0f, 0.0f, 0.0f, 0.0f};
  u = apply_wrap_and_crop(u, wrapU, t.width);
  v = apply_wrap_and_crop(v, wrapV, t.height);
  w = apply_wrap_and_crop(w, wrapW, t.depth);
  return sample_trilinear_repeat(t, u, v, w);
}
__device__ inline F4 tex_texel_float4_3d(const TextureRec3& t, bool valid, int x, int y, int z)
{
  if (!valid || x < 0 || x >= (int)t.width || y < 0 || y >= (int)t.height || z < 0 || z >= (int)t.depth) return F4{0.0f, 0.0f, 0.0f, 0.0f};
  return ld4(&reinterpret_cast<const F4*>(t.texels)[((size_t)z * t.height + (size_t)y) * t.width + (size_t)x]);
}
// scene_data_lookup_float4x4 (mdl_interface.glsl:476-479) is "return default_value; // TODO: not implemented" in the reference: the caller keeps its default.

// mdl_cutout_opacity of the closed forms from the raw opacity value: UsdPreviewSurface opacity with the opacityThreshold switch, OpenPBR
// geometry_opacity (== host cutoutOpacity / oracle cutout_opacity)
__device__ __forceinline__ float cutout_rule(uint32_t klass, float op, float threshold)
{
  float cl = op > 0.0f ? op : 0.0f; cl = cl < 1.0f ? cl : 1.0f;
  if (klass == 2u) return cl;
  if (threshold > 0.0f) return (op >= threshold) ? 1.0f : 0.0f;
  return cl;
}
// Cutout opacity of a candidate hit (rp_main.ahit:51-60 evaluates the material's cutout expression with the candidate's shading
// state): the constant, or -- when the opacity input is textured -- channel `channel` of texel * scale + bias at the candidate's st
// (st interpolated as setup_shading_state does, mdl_shading_state.glsl:62-65).
// UsdTransform2d between the primvar reader and a UsdUVTexture's `st` (UsdPreviewSurface specification: scale, then rotation, then translation), folded into
// six floats by the front end; fixed association, no contraction (== oracle tex_transform_st)
__device__ __forceinline__ void tex_transform_st(const TexBindingRec& b, float& u, float& v)
{
  if (!(b.mode & TEX_MODE_XFORM)) return;
  const float s = u, t = v;
  u = (b.xf[0] * s + b.xf[1] * t) + b.xf[2];
  v = (b.xf[3] * s + b.xf[4] * t) + b.xf[5];
}
__device__ inline float cutout_opacity_at(const SceneView& sc, uint32_t matWord, uint32_t triIdx, float hu, float hv)
{
  const MaterialRec* m = &sc.materials[matWord & 0x00ffffffu];
  if (!(m->flags & MAT_FLAG_OPACITY_TEX)) return m->p[MP_CUTOUT];
  const uint4 td = reinterpret_cast<const uint4*>(sc.tris)[(size_t)triIdx * 4u + 3u]; // (i0, i1, i2, prim)
  const float bx = 1.0f - hu - hv, by = hu, bz = hv;
  float u, v;
  if (sc.shadePacked) { // one line: the mesh triangle's shading record carries the three uv pairs
    const TriShade& q = sc.triShade[td.x];
    u = (bx * q.uv[0][0] + by * q.uv[1][0]) + bz * q.uv[2][0];
    v = (bx * q.uv[0][1] + by * q.uv[1][1]) + bz * q.uv[2][1];
  } else {
    u = (bx * sc.verts[td.x].u + by * sc.verts[td.y].u) + bz * sc.verts[td.z].u;
    v = (bx * sc.verts[td.x].v + by * sc.verts[td.y].v) + bz * sc.verts[td.z].v;
  }
  const TexBindingRec& b = m->tex[TEX_OPACITY];
  tex_transform_st(b, u, v);
  const F4 t = tex_lookup_float4_2d(sc.textures[b.tex - 1u], u, v, b.mode & 0xffu, (b.mode >> 8) & 0xffu);
  const uint32_t ch = (b.mode >> 16) & 3u;
  const float raw = ch == 0u
      ? t.x * b.scale[0] + b.bias[0] : (ch == 1u ? t.y * b.scale[1] + b.bias[1] : (ch == 2u ? t.z * b.scale[2] + b.bias[2] : t.w * b.scale[3] + b.bias[3]));
  return cutout_rule(m->klass, raw, m->p[15]); // p[15] = opacityThreshold
}

} // namespace gi
