// gi_c.cpp -- global state, initialisation, host-side arithmetic helpers, render buffers, scene options and statistics. Implements include/gi_c.h together with
// the files below. (one of the translation units gi_c.cpp was split into in round 6; shared declarations: gi_host.h)
#include "gi_host.h"

// ---------------------------------------------------------------------------------------------------------------
// global state
// ---------------------------------------------------------------------------------------------------------------
thread_local std::string t_lastError;
void setError(const std::string& e) { t_lastError = e; fprintf(stderr, "[gatling_gi] error: %s\n", e.c_str()); }
Context g_ctx;
double nowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// glm::packHalf2x16 (round to nearest even)
uint16_t f32ToF16(float f)
{
  uint32_t x; memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u, absx = x & 0x7fffffffu;
  if (absx >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((absx > 0x7f800000u) ? 0x200u : 0u));
  if (absx >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
  if (absx < 0x33000001u) return (uint16_t)sign;
  int exp = (int)(absx >> 23) - 127;
  uint32_t man = (absx & 0x7fffffu) | 0x800000u;
  if (exp < -14) {
    int shift = -14 - exp + 13;
    uint32_t r = man >> shift, rem = man & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1u))) r++;
    return (uint16_t)(sign | r);
  }
  uint32_t r = ((uint32_t)(exp + 15) << 10) | ((man >> 13) & 0x3ffu), rem = man & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1u))) r++;
  return (uint16_t)(sign | r);
}
float f16ToF32(uint16_t h)
{
  uint32_t sign = ((uint32_t)h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu, out;
  if (exp == 0) { float v = (float)man * 5.9604644775390625e-8f; return sign ? -v : v; }
  if (exp == 31) out = sign | 0x7f800000u | (man << 13); else out = sign | ((exp + 112u) << 23) | (man << 13);
  float f; memcpy(&f, &out, 4); return f;
}
uint32_t packHalf2x16(float a, float b) { return (uint32_t)f32ToF16(a) | ((uint32_t)f32ToF16(b) << 16); }

// _EncodeDirection, Gi.cpp:287-300 (glm::packUnorm2x16 rounds)
uint32_t encodeDirection(const float* vin)
{
  float x = vin[0], y = vin[1], z = vin[2];
  float inv = 1.0f / sqrtf((x * x + y * y) + z * z);
  x *= inv; y *= inv; z *= inv;
  float s = fabsf(x) + fabsf(y) + fabsf(z);
  x /= s; y /= s; z /= s;
  float px = x >= 0.0f ? 1.0f : -1.0f, py = y >= 0.0f ? 1.0f : -1.0f, ex, ey;
  if (z < 0.0f) { ex = (1.0f - fabsf(y)) * px; ey = (1.0f - fabsf(x)) * py; } else { ex = x; ey = y; }
  ex = ex * 0.5f + 0.5f; ey = ey * 0.5f + 0.5f;
  ex = std::min(std::max(ex, 0.0f), 1.0f); ey = std::min(std::max(ey, 0.0f), 1.0f);
  return (uint32_t)nearbyintf(ex * 65535.0f) | ((uint32_t)nearbyintf(ey * 65535.0f) << 16);
}

// decode_direction (common.glsl:198-207) evaluated once per vertex on the host, operation for operation what
// gi_decode_direction / the oracle execute (IEEE fp32, no contraction), so results stay bit-identical.
void decodeDirection(uint32_t e, float out[3])
{
  float ex = (float)(e & 0xffffu) / 65535.0f, ey = (float)(e >> 16) / 65535.0f;
  ex = ex * 2.0f - 1.0f; ey = ey * 2.0f - 1.0f;
  float x = ex, y = ey, z = 1.0f - fabsf(ex) - fabsf(ey);
  float t = (-z > 0.0f) ? -z : 0.0f;
  x += (x >= 0.0f) ? -t : t;
  y += (y >= 0.0f) ? -t : t;
  float inv = 1.0f / sqrtf((x * x + y * y) + z * z);
  out[0] = x * inv; out[1] = y * inv; out[2] = z * inv;
}

// Turbo colour map (A. Mikhailov's polynomial fit of Google's Turbo look-up table; the reference indexes the 256-entry table,
// Gi.cpp:338-341).  Same association as the oracle's turbo_colormap.
void turboColormap(float x, float* rgb)
{
  const float x2 = x * x, x3 = x2 * x, x4 = x2 * x2, x5 = x4 * x;
  rgb[0] = (((0.13572138f + 4.61539260f * x) + -42.66032258f * x2) + 132.13108234f * x3) + (-152.94239396f * x4 + 59.28637943f * x5);
  rgb[1] = (((0.09140261f + 2.19418839f * x) + 4.84296658f * x2) + -14.18503333f * x3) + (4.27729857f * x4 + 2.82956604f * x5);
  rgb[2] = (((0.10667330f + 12.64194608f * x) + -60.58204836f * x2) + 110.36276771f * x3) + (-89.90310912f * x4 + 27.34824973f * x5);
}

// per-material constants of the closed-form BSDFs (DESIGN.md "Materials"); same fp32 formulas as the oracle's ups_params
float cutoutOpacity(const MaterialRec& m) // same rule as the oracle's cutout_opacity
{
  float op = m.p[GI_C_P_OPACITY], th = m.p[GI_C_P_OPACITY_THRESHOLD];
  float cl = op > 0.0f ? op : 0.0f; cl = cl < 1.0f ? cl : 1.0f;
  if (m.klass == GI_C_MAT_OPEN_PBR) return cl;
  if (th > 0.0f) return (op >= th) ? 1.0f : 0.0f;
  return cl;
}

void deriveMaterialConstants(MaterialRec& m)
{
  const float cutout = cutoutOpacity(m);
  struct SetCutout { MaterialRec& m; float v; ~SetCutout() { m.p[MP_CUTOUT] = v; } } setCutout{m, cutout};
  const float* p = m.p;
  if (m.klass == GI_C_MAT_OPEN_PBR) { // same fp32 formulas as the oracle's opbr_params (open_pbr_surface.mtlx:306-373)
    float bw = p[GI_C_P_BASE_WEIGHT], sw = p[GI_C_P_SPECULAR_WEIGHT];
    float r = p[GI_C_P_ROUGHNESS], cr = p[GI_C_P_CLEARCOAT_ROUGHNESS], coat = p[GI_C_P_CLEARCOAT];
    float cior = p[GI_C_P_COAT_IOR], ior = p[GI_C_P_IOR];
    float qc = (cior - 1.0f) / (cior + 1.0f);
    float ratio = ior / cior, inv = cior / ior;
    float etaCoated = (ratio > 1.0f) ? ratio : inv;
    float etaS = etaCoated * coat + ior * (1.0f - coat);
    float q = (etaS - 1.0f) / (etaS + 1.0f);
    float f0 = sw * (q * q); f0 = f0 > 0.0f ? f0 : 0.0f; f0 = f0 < 0.99999f ? f0 : 0.99999f;
    float eps = ((etaS - 1.0f) > 0.0f ? 1.0f : ((etaS - 1.0f) < 0.0f ? -1.0f : 0.0f)) * sqrtf(f0);
    float depth = p[GI_C_P_TRANSMISSION_DEPTH];
    float out[MAT_PARAM_COUNT]; memcpy(out, p, sizeof(out));
    for (int i = 0; i < 3; i++) {
      out[MP_ALBEDO + i] = p[GI_C_P_BASE_COLOR + i] * bw;
      out[MP_F0 + i] = p[GI_C_P_SPECULAR_COLOR + i] * sw;
    }
    { // dielectric base VDF (open_pbr_surface.mtlx:220-298): absorption = extinction - scattering, shifted to be non-negative
      float ab[3];
      for (int i = 0; i < 3; i++) {
        float tc = p[GI_C_P_TRANSMISSION_COLOR + i]; tc = tc > 1e-6f ? tc : 1e-6f;
        const float ext = (depth > 0.0f) ? -logf(tc) / depth : 0.0f;
        const float sc = (depth > 0.0f) ? p[GI_C_P_TRANSMISSION_SCATTER + i] / depth : 0.0f;
        ab[i] = ext - sc;
      }
      float mn = ab[0] < ab[1] ? ab[0] : ab[1]; mn = mn < ab[2] ? mn : ab[2];
      for (int i = 0; i < 3; i++) out[MP_SIGMA_A + i] = (depth > 0.0f) ? ((0.0f > mn) ? ab[i] - mn : ab[i]) : 0.0f;
    }
    { // coat roughening (open_pbr_surface.mtlx:101-131), same fp32 operations as opbr_effective_roughness
      const float c4 = (cr * cr) * (cr * cr), r4 = (r * r) * (r * r);
      float t = 2.0f * c4 + r4; t = t < 1.0f ? t : 1.0f;
      const float ra = sqrtf(sqrtf(t));
      r = ra * coat + r * (1.0f - coat);
    }
    out[MP_ALPHA] = (r * r > 0.001f) ? r * r : 0.001f; out[MP_COAT] = coat; out[MP_COAT_ALPHA] = (cr * cr > 0.001f) ? cr * cr : 0.001f;
    out[MP_COAT_F0] = qc * qc; out[MP_ETA] = (1.0f + eps) / (1.0f - eps);
    // which optional lobes the material has, in the slot of the thin-walled switch: materials without them do not load their inputs per hit (gi_types.h
    // MP_FEATURES)
    { // volumetric subsurface medium (oracle opbr_params "Volumetric subsurface", the same fp32 operations): extinction 1 / (radius * scale), albedo 1 - s^2
      for (int i = 0; i < 3; i++) {
        float c = p[GI_C_P_SUBSURFACE_COLOR + i]; c = c > 0.0f ? c : 0.0f; c = c < 1.0f ? c : 1.0f;
        const float sq = sqrtf((9.59217f + 41.6808f * c) + (17.7126f * c) * c);
        const float sv = (4.09712f + 4.20863f * c) - sq;
        float alb = 1.0f - sv * sv; alb = alb > 0.0f ? alb : 0.0f; alb = alb < 1.0f ? alb : 1.0f;
        float rr = p[GI_C_P_SUBSURFACE_RADIUS] * p[GI_C_P_SUBSURFACE_RADIUS_SCALE + i]; rr = rr > 1e-6f ? rr : 1e-6f;
        m.sss[3 + i] = 1.0f / rr; m.sss[i] = alb * m.sss[3 + i];
      }
      m.sss[6] = 1.0f; m.sss[7] = 0.0f;
    }
    // geometry_tangent / geometry_coat_tangent (open_pbr_surface.mtlx:89, 91; 385 ... 457, 561) as the turns of the frame's tangent documents bind to them:
    // only anisotropic lobes can tell (the oracle's conditions, opbr_params; the same libm calls on the same angles).  sss[6..7]: the coat's turn, [8..9]: the base
    // lobes', [10..11]: the coat's relative to the turned base frame (k_shade turns the frame in place; a coat without a frame of its own starts from there)
    const bool specTurned = p[GI_C_P_SPECULAR_ANISOTROPY] > 0.0f && p[GI_C_P_SPECULAR_ROTATION] != 0.0f;
    const bool coatTurned = p[GI_C_P_CLEARCOAT] > 0.0f && p[GI_C_P_COAT_ANISOTROPY] > 0.0f && (p[GI_C_P_COAT_ROTATION] != 0.0f || specTurned);
    m.sss[8] = 1.0f; m.sss[9] = 0.0f; m.sss[10] = 1.0f; m.sss[11] = 0.0f;
    if (coatTurned) { const float a = 6.2831855f * p[GI_C_P_COAT_ROTATION]; m.sss[6] = cosf(a); m.sss[7] = sinf(a); }
    if (specTurned) {
      const float a = 6.2831855f * p[GI_C_P_SPECULAR_ROTATION], r = 6.2831855f * (p[GI_C_P_COAT_ROTATION] - p[GI_C_P_SPECULAR_ROTATION]);
      m.sss[8] = cosf(a); m.sss[9] = sinf(a); m.sss[10] = cosf(r); m.sss[11] = sinf(r);
    }
    out[MP_FEATURES] = (float)((p[GI_C_P_THIN_WALLED] != 0.0f ? MATF_THIN_WALLED : 0u) | (p[GI_C_P_FUZZ_WEIGHT] > 0.0f ? MATF_FUZZ : 0u) |
                               ((p[GI_C_P_THIN_WALLED] == 0.0f && p[GI_C_P_SUBSURFACE_WEIGHT] > 0.0f) ? MATF_SSS_VOLUME : 0u) |
                               ((p[GI_C_P_SPECULAR_ANISOTROPY] > 0.0f || p[GI_C_P_COAT_ANISOTROPY] > 0.0f) ? MATF_ANISOTROPY : 0u) |
                               (p[GI_C_P_THIN_FILM_WEIGHT] > 0.0f ? MATF_THIN_FILM : 0u) | (coatTurned ? MATF_COAT_ROTATION : 0u) |
                               (specTurned ? MATF_SPEC_ROTATION : 0u));
    memcpy(m.p, out, sizeof(out));
    return;
  }
  float r = p[GI_C_P_ROUGHNESS], cr = p[GI_C_P_CLEARCOAT_ROUGHNESS];
  float alpha = (r * r > 0.001f) ? r * r : 0.001f, coatAlpha = (cr * cr > 0.001f) ? cr * cr : 0.001f;
  float albedo[3], F0[3];
  if (p[GI_C_P_USE_SPECULAR_WORKFLOW] != 0.0f) {
    for (int i = 0; i < 3; i++) { F0[i] = p[GI_C_P_SPECULAR_COLOR + i]; albedo[i] = p[GI_C_P_BASE_COLOR + i]; }
  } else {
    float ior = p[GI_C_P_IOR], metal = p[GI_C_P_METALLIC];
    float q = (1.0f - ior) / (1.0f + ior), f0 = q * q;
    for (int i = 0; i < 3; i++) { F0[i] = f0 * (1.0f - metal) + p[GI_C_P_BASE_COLOR + i] * metal; albedo[i] = p[GI_C_P_BASE_COLOR + i] * (1.0f - metal); }
  }
  for (int i = 0; i < 3; i++) { m.p[MP_ALBEDO + i] = albedo[i]; m.p[MP_F0 + i] = F0[i]; }
  m.p[MP_ALPHA] = alpha; m.p[MP_COAT] = p[GI_C_P_CLEARCOAT]; m.p[MP_COAT_ALPHA] = coatAlpha;
}

void SceneDevice::releaseAll()
{
  dNodes.release(); dTris.release(); dInstances.release(); dVerts.release(); dTriFaceId.release(); dTriShade.release(); dTriGeomNormal.release();
  dTlasNodes.release(); dBlasNodes.release(); dTlasItems.release(); dFlatOfOrig.release(); dBlasTris.release(); dInstTrav.release();
  for (auto* b : dTexels) { b->release(); delete b; }
  dTexels.clear(); dTextures.release(); dMeshes.release(); dSceneData.release();
  dMaterials.release(); dSphere.release(); dDistant.release(); dRect.release(); dDisk.release(); dRectFrames.release(); dDiskFrames.release();
  slots.release(); media.release(); scratchColor.release(); neeKey.release(); pathSegments.release(); sampleBuf.release(); accum.release();
  for (uint32_t q = 0; q < Q_COUNT; q++) { qSlot[q].release(); qA[q].release(); qB[q].release(); qC[q].release(); }
  qFresh[0].release(); qFresh[1].release();
  dCounters.release();
  if (hCounters) { (void)hipHostFree(hCounters); hCounters = nullptr; }
  if (evShade) { (void)hipEventDestroy(evShade); evShade = nullptr; }
  if (evShadow) { (void)hipEventDestroy(evShadow); evShadow = nullptr; }
  if (hPoll) { (void)hipHostFree(hPoll); hPoll = nullptr; for (hipEvent_t& e : pollEvent) { (void)hipEventDestroy(e); e = nullptr; } }
  for (hipEvent_t e : eventPool) (void)hipEventDestroy(e);
  eventPool.clear();
}

// ---------------------------------------------------------------------------------------------------------------
// init
// ---------------------------------------------------------------------------------------------------------------
extern "C" {

const char* giCGetLastError(void) { return t_lastError.c_str(); }

// The device list: giCInitializeDevices' argument, else $GATLING_DEVICES ("0,1,2,3" or "all"), else the one ordinal of giCInitialize.  The same ordinal may be
// listed twice (two contexts on one GPU): that is how the multi-device path is tested on a one-GPU box.
static int initDevices(const std::vector<int>& ordinals)
{
  if (g_ctx.initialized) return GI_C_OK;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n == 0) { setError("no HIP device available: this library has no CPU fallback"); return GI_C_ERROR; }
  if (ordinals.empty()) { setError("empty device list"); return GI_C_ERROR; }
  for (int d : ordinals) if (d < 0 || d >= n) { setError("device ordinal out of range"); return GI_C_ERROR; }
  std::vector<DevCtx> devs;
  for (int d : ordinals) {
    DevCtx c; c.device = d;
    HIP_TRY(hipSetDevice(d));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, d));
    c.cuCount = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c.stream2, hipStreamNonBlocking));
    devs.push_back(c);
  }
  // the row shares travel to the primary device over xGMI: peer access both ways. The outcome is kept (giCGetDevicePeerAccess) and decides how a device's share
  // is gathered: without peer access a hipMemcpyDefault between two devices silently stages through pageable host memory -- the library then does it itself,
  // through a pinned buffer
  for (size_t i = 1; i < devs.size(); i++) {
    if (devs[i].device == devs[0].device) continue; // another context on the same GPU (tests): its memory is directly addressable
    int can01 = 0, can10 = 0;
    const hipError_t q0 = hipDeviceCanAccessPeer(&can01, devs[0].device, devs[i].device), q1 = hipDeviceCanAccessPeer(&can10, devs[i].device, devs[0].device);
    if (q0 != hipSuccess || q1 != hipSuccess) { devs[i].peer = -1; (void)hipGetLastError(); continue; }
    if (!can01 || !can10) { devs[i].peer = 0; continue; }
    (void)hipSetDevice(devs[0].device); const hipError_t e0 = hipDeviceEnablePeerAccess(devs[i].device, 0);
    (void)hipSetDevice(devs[i].device); const hipError_t e1 = hipDeviceEnablePeerAccess(devs[0].device, 0);
    auto fine = [](hipError_t e) { return e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled; };
    devs[i].peer = (fine(e0) && fine(e1)) ? 1 : -1;
    if (devs[i].peer != 1) fprintf(stderr,
        "[gatling_gi] peer access between devices %d and %d could not be enabled (%s / %s): row shares of device %d go through pinned host memory\n",
                                   devs[0].device, devs[i].device, hipGetErrorString(e0), hipGetErrorString(e1), devs[i].device);
    (void)hipGetLastError();
  }
  HIP_TRY(hipSetDevice(devs[0].device));
  g_ctx.devs = devs;
  g_ctx.device = devs[0].device; g_ctx.cuCount = devs[0].cuCount; g_ctx.stream = devs[0].stream;
  g_ctx.initialized = true;
  return GI_C_OK;
}

int giCInitialize(int deviceOrdinal)
{
  if (g_ctx.initialized) return GI_C_OK;
  std::vector<int> ordinals{deviceOrdinal};
  if (const char* e = getenv("GATLING_DEVICES")) { // "0,1,2,3", or "all"
    ordinals.clear();
    if (!strcmp(e, "all")) { int n = 0; if (hipGetDeviceCount(&n) == hipSuccess) for (int d = 0; d < n; d++) ordinals.push_back(d); }
    else for (const char* p = e; *p;) { char* end = nullptr; const long v = strtol(p, &end, 10); if (end == p) { p++; continue; } ordinals.push_back((int)v);
        p = end; }
    if (ordinals.empty()) ordinals.push_back(deviceOrdinal);
  }
  return initDevices(ordinals);
}

int giCInitializeDevices(const int32_t* deviceOrdinals, uint32_t count)
{
  if (!deviceOrdinals || count == 0) { setError("giCInitializeDevices: empty device list"); return GI_C_ERROR; }
  return initDevices(std::vector<int>(deviceOrdinals, deviceOrdinals + count));
}

uint32_t giCGetApiVersion(void) { return GI_C_API_VERSION; }
uint32_t giCGetDeviceCount(void) { return g_ctx.initialized ? (uint32_t)g_ctx.devs.size() : 0u; }
int32_t giCGetDevicePeerAccess(uint32_t index) { return (g_ctx.initialized && index < g_ctx.devs.size()) ? (int32_t)g_ctx.devs[index].peer : -1; }

void giCTerminate(void)
{
  if (!g_ctx.initialized) return;
  for (auto& w : g_ctx.workers) if (w) w->shutdown();
  g_ctx.workers.clear();
  for (DevCtx& c : g_ctx.devs) {
    (void)hipSetDevice(c.device);
    (void)hipStreamSynchronize(c.stream);
    (void)hipStreamDestroy(c.stream);
    if (c.stream2) { (void)hipStreamSynchronize(c.stream2); (void)hipStreamDestroy(c.stream2); }
  }
  g_ctx.devs.clear();
  g_ctx.stream = nullptr;
  g_ctx.initialized = false;
}

// ---------------------------------------------------------------------------------------------------------------
// render buffers (Gi.cpp:2978-3006; memory is created lazily by giRender, :1997-2034 -- here at creation so that
// giCGetRenderBufferMem is valid immediately)
// ---------------------------------------------------------------------------------------------------------------
GiCRenderBuffer* giCCreateRenderBuffer(uint32_t width, uint32_t height, int32_t format)
{
  if (!g_ctx.initialized) { setError("giCCreateRenderBuffer before giCInitialize"); return nullptr; }
  uint32_t stride = (format == GI_C_FORMAT_FLOAT32_VEC4) ? 16u : 4u;
  auto* rb = new GiCRenderBuffer{width, height, stride, (size_t)width * height * stride};
  size_t bytes = rb->size ? rb->size : 16;
  if (hipMalloc(&rb->deviceMem, bytes) != hipSuccess || hipHostMalloc(&rb->hostMem, bytes, hipHostMallocDefault) != hipSuccess) {
    setError("failed to allocate render buffer");
    if (rb->deviceMem) (void)hipFree(rb->deviceMem);
    delete rb; return nullptr;
  }
  (void)hipMemset(rb->deviceMem, 0, bytes);
  memset(rb->hostMem, 0, bytes);
  return rb;
}
void giCDestroyRenderBuffer(GiCRenderBuffer* rb)
{
  if (!rb) return;
  std::lock_guard<std::mutex> g(g_ctx.resourceMutex);
  (void)hipStreamSynchronize(g_ctx.stream);
  if (rb->deviceMem) (void)hipFree(rb->deviceMem);
  if (rb->hostMem) (void)hipHostFree(rb->hostMem);
  if (rb->stageMem) (void)hipHostFree(rb->stageMem);
  for (size_t i = 0; i < rb->replicaMem.size(); i++)
    if (rb->replicaMem[i] && i + 1 < g_ctx.devs.size()) { (void)hipSetDevice(g_ctx.devs[i + 1].device); (void)hipFree(rb->replicaMem[i]); }
  (void)hipSetDevice(g_ctx.device);
  delete rb;
}
void* giCGetRenderBufferMem(GiCRenderBuffer* rb) { return rb ? rb->hostMem : nullptr; }
void* giCGetRenderBufferDeviceMem(GiCRenderBuffer* rb) { return rb ? rb->deviceMem : nullptr; }
void giCSetRenderBufferDeviceOnly(GiCRenderBuffer* rb, int32_t deviceOnly) { if (rb) rb->deviceOnly = deviceOnly != 0; }

int giCSetSceneOption(GiCScene* scene, int32_t option, int32_t value)
{
  if (!scene) return GI_C_ERROR;
  std::lock_guard<std::mutex> g(scene->mutex);
  if (option == GI_C_SCENE_OPTION_COUNT_TRAVERSAL) { scene->countTraversal = value != 0; return GI_C_OK; }
  if (option == GI_C_SCENE_OPTION_KERNEL_TIMERS) { scene->kernelTimers = value != 0; scene->kernelTimerStride = value > 0 ? (uint32_t)value : 1u;
      return GI_C_OK; }
  if (option == GI_C_SCENE_OPTION_POOL_SLOTS) { scene->optPoolSlots = value > 0 ? (uint64_t)value : 0; return GI_C_OK; }
  if (option == GI_C_SCENE_OPTION_TWO_LEVEL) { scene->optTwoLevel = value < 0 ? -1 : (value ? 1 : 0); scene->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER;
      return GI_C_OK; }
  if (option == GI_C_SCENE_OPTION_TRACE_DYNAMIC) { scene->optTraceDyn = value < 0 ? -1 : (value > 64 ? 64 : value); return GI_C_OK; }
  if (option == GI_C_SCENE_OPTION_FUSED_PATH) { scene->optFusedPath = value < 0 ? -1 : (value > 2 ? -1 : value); return GI_C_OK; }
  if (option == GI_C_SCENE_OPTION_SAMPLE_BUFFER_MB) { scene->optSampleBufferMb = value > 0 ? (uint64_t)value : 0; return GI_C_OK; }
  if (option == GI_C_SCENE_OPTION_DEVICES) { scene->optDevices = value > 0 ? value : 0; scene->dirty |= DIRTY_BVH | DIRTY_LIGHTS | DIRTY_FRAMEBUFFER;
      /* replicas are made with the build; a NEW replica also needs the lights, which travel under DIRTY_LIGHTS only */ return GI_C_OK; }
  setError("unknown scene option"); return GI_C_ERROR;
}

int giCGetRenderStats(const GiCScene* scene, GiCRenderStats* out)
{
  if (!scene || !out) return GI_C_ERROR;
  *out = scene->stats;
  return GI_C_OK;
}

} // extern "C"
