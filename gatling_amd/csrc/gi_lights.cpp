// gi_lights.cpp -- light stores and setters, dome light, upload of the light records (Gi.cpp:2573-2976)
// (one of the translation units gi_c.cpp was split into in round 6; shared declarations: gi_host.h)
#include "gi_host.h"

extern "C" {
// ---------------------------------------------------------------------------------------------------------------
// lights (defaults and derived fields: Gi.cpp:2573-2976)
// ---------------------------------------------------------------------------------------------------------------
#define LIGHT_DIRTY(l) (l)->scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER

GiCSphereLight* giCCreateSphereLight(GiCScene* scene)
{
  if (!scene) return nullptr;
  std::lock_guard<std::mutex> g(scene->mutex);
  auto* l = new GiCSphereLight{scene, 0};
  SphereLightRec r{}; r.ds = packHalf2x16(1.0f, 1.0f); r.area = 1.0f; r.radius[0] = r.radius[1] = r.radius[2] = 0.5f;
  l->index = scene->sphereLights.add(l, r);
  scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER;
  return l;
}
void giCDestroySphereLight(GiCScene* scene, GiCSphereLight* l)
{
  if (!scene || !l) return;
  std::lock_guard<std::mutex> g(scene->mutex);
  scene->sphereLights.remove(l->index); scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER;
  delete l;
}
void giCSetSphereLightPosition(GiCSphereLight* l, const float* p) { std::lock_guard<std::mutex> lk(l->scene->mutex);
    memcpy(l->scene->sphereLights.recs[l->index].pos, p, 12); LIGHT_DIRTY(l); }
void giCSetSphereLightBaseEmission(GiCSphereLight* l, const float* c) { std::lock_guard<std::mutex> lk(l->scene->mutex);
    memcpy(l->scene->sphereLights.recs[l->index].em, c, 12); LIGHT_DIRTY(l); }
void giCSetSphereLightRadius(GiCSphereLight* l, float rx, float ry, float rz)
{ std::lock_guard<std::mutex> lk(l->scene->mutex);
  // Knud Thomsen ellipsoid surface approximation (Gi.cpp:2635-2651)
  float ab = powf(rx * ry, 1.6f), ac = powf(rx * rz, 1.6f), bc = powf(ry * rz, 1.6f);
  float area = float(powf((ab + ac + bc) / 3.0f, 1.0f / 1.6f) * 4.0f * M_PI);
  SphereLightRec& r = l->scene->sphereLights.recs[l->index];
  r.radius[0] = rx; r.radius[1] = ry; r.radius[2] = rz; r.area = area;
  LIGHT_DIRTY(l);
}
void giCSetSphereLightDiffuseSpecular(GiCSphereLight* l, float d, float s) { std::lock_guard<std::mutex> lk(l->scene->mutex);
    l->scene->sphereLights.recs[l->index].ds = packHalf2x16(d, s); LIGHT_DIRTY(l); }

GiCDistantLight* giCCreateDistantLight(GiCScene* scene)
{
  if (!scene) return nullptr;
  std::lock_guard<std::mutex> g(scene->mutex);
  auto* l = new GiCDistantLight{scene, 0};
  DistantLightRec r{}; r.ds = packHalf2x16(1.0f, 1.0f); r.invPdf = 1.0f;
  l->index = scene->distantLights.add(l, r);
  scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER;
  return l;
}
void giCDestroyDistantLight(GiCScene* scene, GiCDistantLight* l)
{
  if (!scene || !l) return;
  std::lock_guard<std::mutex> g(scene->mutex);
  scene->distantLights.remove(l->index); scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER;
  delete l;
}
void giCSetDistantLightDirection(GiCDistantLight* l, const float* d) { std::lock_guard<std::mutex> lk(l->scene->mutex);
    memcpy(l->scene->distantLights.recs[l->index].dir, d, 12); LIGHT_DIRTY(l); }
void giCSetDistantLightBaseEmission(GiCDistantLight* l, const float* c) { std::lock_guard<std::mutex> lk(l->scene->mutex);
    memcpy(l->scene->distantLights.recs[l->index].em, c, 12); LIGHT_DIRTY(l); }
void giCSetDistantLightAngle(GiCDistantLight* l, float angle)
{ std::lock_guard<std::mutex> lk(l->scene->mutex);
  float half = 0.5f * angle; // Gi.cpp:2723-2735
  DistantLightRec& r = l->scene->distantLights.recs[l->index];
  r.angle = angle; r.invPdf = (half > 0.0f) ? float(2.0f * M_PI * (1.0f - cosf(half))) : 1.0f;
  LIGHT_DIRTY(l);
}
void giCSetDistantLightDiffuseSpecular(GiCDistantLight* l, float d, float s) { std::lock_guard<std::mutex> lk(l->scene->mutex);
    l->scene->distantLights.recs[l->index].ds = packHalf2x16(d, s); LIGHT_DIRTY(l); }

GiCRectLight* giCCreateRectLight(GiCScene* scene)
{
  if (!scene) return nullptr;
  std::lock_guard<std::mutex> g(scene->mutex);
  auto* l = new GiCRectLight{scene, 0};
  const float t0[3] = {1, 0, 0}, t1[3] = {0, 1, 0};
  RectLightRec r{}; r.width = 1.0f; r.height = 1.0f; r.t0 = encodeDirection(t0); r.t1 = encodeDirection(t1); r.ds = packHalf2x16(1.0f, 1.0f);
  l->index = scene->rectLights.add(l, r);
  scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER;
  return l;
}
void giCDestroyRectLight(GiCScene* scene, GiCRectLight* l)
{
  if (!scene || !l) return;
  std::lock_guard<std::mutex> g(scene->mutex);
  scene->rectLights.remove(l->index); scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER;
  delete l;
}
void giCSetRectLightOrigin(GiCRectLight* l, const float* o) { std::lock_guard<std::mutex> lk(l->scene->mutex);
    memcpy(l->scene->rectLights.recs[l->index].origin, o, 12); LIGHT_DIRTY(l); }
void giCSetRectLightTangents(GiCRectLight* l, const float* t0, const float* t1)
{ std::lock_guard<std::mutex> lk(l->scene->mutex);
  RectLightRec& r = l->scene->rectLights.recs[l->index]; r.t0 = encodeDirection(t0); r.t1 = encodeDirection(t1); LIGHT_DIRTY(l);
}
void giCSetRectLightBaseEmission(GiCRectLight* l, const float* c) { std::lock_guard<std::mutex> lk(l->scene->mutex);
    memcpy(l->scene->rectLights.recs[l->index].em, c, 12); LIGHT_DIRTY(l); }
void giCSetRectLightDimensions(GiCRectLight* l, float w, float h) { std::lock_guard<std::mutex> lk(l->scene->mutex);
    RectLightRec& r = l->scene->rectLights.recs[l->index]; r.width = w; r.height = h; LIGHT_DIRTY(l); }
void giCSetRectLightDiffuseSpecular(GiCRectLight* l, float d, float s) { std::lock_guard<std::mutex> lk(l->scene->mutex);
    l->scene->rectLights.recs[l->index].ds = packHalf2x16(d, s); LIGHT_DIRTY(l); }

GiCDiskLight* giCCreateDiskLight(GiCScene* scene)
{
  if (!scene) return nullptr;
  std::lock_guard<std::mutex> g(scene->mutex);
  auto* l = new GiCDiskLight{scene, 0};
  const float t0[3] = {1, 0, 0}, t1[3] = {0, 1, 0};
  DiskLightRec r{}; r.rx = 0.5f; r.ry = 0.5f; r.t0 = encodeDirection(t0); r.t1 = encodeDirection(t1); r.ds = packHalf2x16(1.0f, 1.0f);
  l->index = scene->diskLights.add(l, r);
  scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER;
  return l;
}
void giCDestroyDiskLight(GiCScene* scene, GiCDiskLight* l)
{
  if (!scene || !l) return;
  std::lock_guard<std::mutex> g(scene->mutex);
  scene->diskLights.remove(l->index); scene->dirty |= DIRTY_LIGHTS | DIRTY_FRAMEBUFFER;
  delete l;
}
void giCSetDiskLightOrigin(GiCDiskLight* l, const float* o) { std::lock_guard<std::mutex> lk(l->scene->mutex);
    memcpy(l->scene->diskLights.recs[l->index].origin, o, 12); LIGHT_DIRTY(l); }
void giCSetDiskLightTangents(GiCDiskLight* l, const float* t0, const float* t1)
{ std::lock_guard<std::mutex> lk(l->scene->mutex);
  DiskLightRec& r = l->scene->diskLights.recs[l->index]; r.t0 = encodeDirection(t0); r.t1 = encodeDirection(t1); LIGHT_DIRTY(l);
}
void giCSetDiskLightBaseEmission(GiCDiskLight* l, const float* c) { std::lock_guard<std::mutex> lk(l->scene->mutex);
    memcpy(l->scene->diskLights.recs[l->index].em, c, 12); LIGHT_DIRTY(l); }
void giCSetDiskLightRadius(GiCDiskLight* l, float rx, float ry) { std::lock_guard<std::mutex> lk(l->scene->mutex);
    DiskLightRec& r = l->scene->diskLights.recs[l->index]; r.rx = rx; r.ry = ry; LIGHT_DIRTY(l); }
void giCSetDiskLightDiffuseSpecular(GiCDiskLight* l, float d, float s) { std::lock_guard<std::mutex> lk(l->scene->mutex);
    l->scene->diskLights.recs[l->index].ds = packHalf2x16(d, s); LIGHT_DIRTY(l); }


GiCDomeLight* giCCreateDomeLight(GiCScene* scene, const char* filePath)
{
  if (!scene) return nullptr;
  auto* l = new GiCDomeLight(); l->scene = scene; l->filePath = filePath ? filePath : "";
  // the reference decodes the file through imgio (Gi.cpp:2215-2230); here: Radiance RGBE, PFM and PNG, anything else stays unloaded
  uint32_t w = 0, h = 0; std::vector<float> px;
  if (!l->filePath.empty() && loadImage(l->filePath.c_str(), /*srgbToLinear=*/false, /*keepHdr=*/true, w, h, px)) {
    GiCTextureDesc td{w, h, px.data()};
    l->texture = giCCreateTexture(scene, &td);
    l->ownsTexture = l->texture != nullptr;
  } else if (!l->filePath.empty()) {
    fprintf(stderr, "[gatling_gi] unable to load dome light texture at '%s' (.hdr, .pfm, .png and baseline .jpg are decoded in-library, other formats through "
                    "giCSetImageLoader)\n", l->filePath.c_str());
  }
  return l;
}
void giCDestroyDomeLight(GiCDomeLight* l)
{
  if (!l) return;
  if (l->ownsTexture) giCDestroyTexture(l->texture);
  std::lock_guard<std::mutex> g(l->scene->mutex);
  l->scene->dirty |= DIRTY_FRAMEBUFFER;
  delete l;
}
void giCSetDomeLightTexture(GiCDomeLight* l, GiCTexture* t)
{
  if (!l) return;
  GiCTexture* old = nullptr;
  {
    std::lock_guard<std::mutex> lk(l->scene->mutex);
    if (l->ownsTexture) { old = l->texture; l->ownsTexture = false; }
    l->texture = t; l->scene->dirty |= DIRTY_FRAMEBUFFER;
  }
  if (old) giCDestroyTexture(old); // takes the scene mutex itself
}
void giCSetDomeLightRotation(GiCDomeLight* l, const float* q) { std::lock_guard<std::mutex> lk(l->scene->mutex); memcpy(l->rotation, q, 16);
    l->scene->dirty |= DIRTY_FRAMEBUFFER; }
void giCSetDomeLightBaseEmission(GiCDomeLight* l, const float* c) { std::lock_guard<std::mutex> lk(l->scene->mutex); memcpy(l->baseEmission, c, 12);
    l->scene->dirty |= DIRTY_FRAMEBUFFER; }
void giCSetDomeLightDiffuseSpecular(GiCDomeLight* l, float d, float s) { std::lock_guard<std::mutex> lk(l->scene->mutex); l->diffuse = d; l->specular = s;
    l->scene->dirty |= DIRTY_FRAMEBUFFER; }

} // extern "C"

// Hostile input (include/gi_c.h): a light with a non-finite field is left out of the device arrays and of the *LightCount uniforms -- light sampling would turn
// it into NaN radiance on every path that draws it (the reference uploads it as it
// is).  The stores keep it: a later setter call with a usable value brings it back.
static bool usableHalfPair(uint32_t ds) { return std::isfinite(f16ToF32((uint16_t)(ds & 0xffffu))) && std::isfinite(f16ToF32((uint16_t)(ds >> 16))); }
template <size_t N> static bool allFinite(const float (&v)[N]) { for (float x : v) if (!std::isfinite(x)) return false; return true; }
static bool usableLight(const SphereLightRec& l)
{ return allFinite(l.pos) && allFinite(l.em) && allFinite(l.radius) && std::isfinite(l.area) && usableHalfPair(l.ds); }
static bool usableLight(const DistantLightRec& l)
{ return allFinite(l.dir) && allFinite(l.em) && std::isfinite(l.angle) && std::isfinite(l.invPdf) && usableHalfPair(l.ds); }
static bool usableLight(const RectLightRec& l)
{ return allFinite(l.origin) && allFinite(l.em) && std::isfinite(l.width) && std::isfinite(l.height) && usableHalfPair(l.ds); }
static bool usableLight(const DiskLightRec& l)
{ return allFinite(l.origin) && allFinite(l.em) && std::isfinite(l.rx) && std::isfinite(l.ry) && usableHalfPair(l.ds); }
template <class Rec> static std::vector<Rec> usableLights(const std::vector<Rec>& in, const char* kind)
{
  std::vector<Rec> out; out.reserve(in.size());
  for (const Rec& r : in) if (usableLight(r)) out.push_back(r);
  if (out.size() != in.size()) fprintf(stderr, "[gatling_gi] warning: %zu %s light(s) with a non-finite field are ignored\n", in.size() - out.size(), kind);
  return out;
}

// decoded tangents and normal of a rect / disk light (gi_types.h LightFrame): decode_direction twice, then cross(t1, t0) in the device's operation order
template <class Rec> static std::vector<LightFrame> lightFrames(const std::vector<Rec>& recs)
{
  std::vector<LightFrame> out(recs.size());
  for (size_t i = 0; i < recs.size(); i++) {
    LightFrame f{};
    decodeDirection(recs[i].t0, f.t0); decodeDirection(recs[i].t1, f.t1);
    const float* a = f.t1; const float* b = f.t0; // gi_device_math.h cross(a, b)
    f.n[0] = a[1] * b[2] - a[2] * b[1]; f.n[1] = a[2] * b[0] - a[0] * b[2]; f.n[2] = a[0] * b[1] - a[1] * b[0];
    out[i] = f;
  }
  return out;
}

int uploadLights(GiCScene* s)
{
  const std::vector<SphereLightRec> sphere = usableLights(s->sphereLights.recs, "sphere");
  const std::vector<DistantLightRec> distant = usableLights(s->distantLights.recs, "distant");
  const std::vector<RectLightRec> rect = usableLights(s->rectLights.recs, "rect");
  const std::vector<DiskLightRec> disk = usableLights(s->diskLights.recs, "disk");
  const std::vector<LightFrame> rectFrames = lightFrames(rect), diskFrames = lightFrames(disk);
  s->lightCounts[0] = (uint32_t)sphere.size(); s->lightCounts[1] = (uint32_t)distant.size(); s->lightCounts[2] = (uint32_t)rect.size();
      s->lightCounts[3] = (uint32_t)disk.size();
  const uint32_t nDev = std::min<uint32_t>(sceneDeviceCount(s), (uint32_t)s->replicas.size() + 1u);
  for (uint32_t d = 0; d < nDev; d++) {
    SceneDevice& D = sceneDevice(s, d);
    HIP_TRY(hipSetDevice(g_ctx.devs[d].device));
    hipStream_t st = g_ctx.devs[d].stream;
    if (D.dSphere.upload(sphere, st) || D.dDistant.upload(distant, st) || D.dRect.upload(rect, st) || D.dDisk.upload(disk, st) ||
        D.dRectFrames.upload(rectFrames, st) || D.dDiskFrames.upload(diskFrames, st))
      return GI_C_ERROR;
    HIP_TRY(hipStreamSynchronize(st));
  }
  HIP_TRY(hipSetDevice(g_ctx.device));
  return GI_C_OK;
}

