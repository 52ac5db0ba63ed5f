/* gi_render -- a plain C99 client of the gi C ABI (include/gi_c.h): loads a flat binary scene (.gscn, written by
 * gatling_amd/scenefile.py), feeds it through giCCreate... the way hdGatling feeds Hydra prims through Gi.h, renders with
 * giCRender and writes the colour AOV.  The shape of the reference's headless standalone (README.md:69-78:
 * `gatling <scene.usd> render.png --image-width W --image-height H --spp N --max-bounces B`), with the .gscn standing in for
 * USD + Hydra; SURVEY.md section 8d asks for exactly this harness.
 *
 *   gcc -std=c99 -O2 -Iinclude tools/gi_render.c -o tools/gi_render -Lgatling_amd -lgatling_gi -Wl,-rpath,$PWD/gatling_amd -lm
 *   tools/gi_render scene.gscn out.pfm [--image-width W] [--image-height H] [--spp N] [--max-bounces B] [--rr-bounce-offset K]
 *                   [--next-event-estimation 0|1] [--medium-stack-size K] [--max-sample-value X] [--rows begin:end[:stride]]
 *                   [--device D] [--stats]
 *   tools/gi_render scene.gscn --info        parses the file and prints its inventory; needs no GPU
 *
 * Output by extension: .pfm (RGB float, bottom row first -- the buffer's own order), .raw (RGBA float32, bottom row first),
 * .ppm (8-bit sRGB, top row first).  No CPU fallback: without a device giCInitialize fails and so does this program. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gi_c.h"

typedef struct { const uint8_t* d; size_t n, p; int bad; } Rd;

static const void* rd_take(Rd* r, size_t n)
{
  if (r->bad || n > r->n - r->p) { r->bad = 1; return NULL; }
  const void* q = r->d + r->p;
  r->p += n;
  return q;
}
static uint32_t rd_u32(Rd* r) { const void* q = rd_take(r, 4); uint32_t v = 0; if (q) memcpy(&v, q, 4); return v; }
static int32_t rd_i32(Rd* r) { return (int32_t)rd_u32(r); }
static void rd_f32(Rd* r, float* out, size_t n) { const void* q = rd_take(r, n * 4); if (q) memcpy(out, q, n * 4); else memset(out, 0, n * 4); }
static char* rd_str(Rd* r) /* malloc'ed, NUL-terminated */
{
  uint32_t n = rd_u32(r);
  const void* q = rd_take(r, n);
  rd_take(r, (4u - (n & 3u)) & 3u);
  char* s = (char*)calloc((size_t)n + 1, 1);
  if (q && s) memcpy(s, q, n);
  return s;
}

typedef struct { uint32_t has, width, height; GiCRenderSettings rs; float clear[4]; } FileSettings;

static int fail(const char* what) { fprintf(stderr, "gi_render: %s\n", what); return 1; }

static int write_image(const char* path, const float* rgba, uint32_t w, uint32_t h)
{
  const char* ext = strrchr(path, '.');
  FILE* f = fopen(path, "wb");
  if (!f) return fail("cannot open the output file");
  if (ext && !strcmp(ext, ".raw")) {
    fwrite(rgba, 16, (size_t)w * h, f);
  } else if (ext && !strcmp(ext, ".ppm")) {
    fprintf(f, "P6\n%u %u\n255\n", w, h);
    for (uint32_t y = h; y-- > 0;)
      for (uint32_t x = 0; x < w; x++)
        for (int c = 0; c < 3; c++) {
          float v = rgba[((size_t)y * w + x) * 4 + c];
          v = v <= 0.0f ? 0.0f : (v >= 1.0f ? 1.0f : v);
          v = v <= 0.0031308f ? 12.92f * v : 1.055f * powf(v, 1.0f / 2.4f) - 0.055f;
          fputc((int)(v * 255.0f + 0.5f), f);
        }
  } else { /* .pfm: "PF", negative scale = little-endian, rows bottom to top */
    fprintf(f, "PF\n%u %u\n-1.0\n", w, h);
    for (size_t i = 0; i < (size_t)w * h; i++) fwrite(rgba + i * 4, 4, 3, f);
  }
  fclose(f);
  return 0;
}

int main(int argc, char** argv)
{
  if (argc < 3) { fprintf(stderr, "Usage: gi_render <scene.gscn> <render.pfm|.raw|.ppm> [options]   |   gi_render <scene.gscn> --info\n"); return 2; }
  const int infoOnly = !strcmp(argv[2], "--info");

  FILE* f = fopen(argv[1], "rb");
  if (!f) return fail("cannot open the scene file");
  fseek(f, 0, SEEK_END);
  long size = ftell(f);
  fseek(f, 0, SEEK_SET);
  uint8_t* data = (uint8_t*)malloc((size_t)size);
  if (!data || fread(data, 1, (size_t)size, f) != (size_t)size) return fail("cannot read the scene file");
  fclose(f);
  Rd r = {data, (size_t)size, 0, 0};
  const void* magic = rd_take(&r, 4);
  if (!magic || memcmp(magic, "GSCN", 4) || rd_u32(&r) != 3u) return fail("not a version-3 .gscn file");

  FileSettings fs;
  memset(&fs, 0, sizeof fs);
  fs.has = rd_u32(&r);
  if (fs.has) {
    fs.width = rd_u32(&r); fs.height = rd_u32(&r);
    const void* q = rd_take(&r, sizeof(GiCRenderSettings));
    if (q) memcpy(&fs.rs, q, sizeof(GiCRenderSettings));
    rd_f32(&r, fs.clear, 4);
  } else { /* the Hydra render-setting defaults (renderDelegate.cpp:93-110), clear colour of the delegate (:229) */
    fs.width = fs.height = 800; /* Argparse.cpp:25-26 */
    fs.rs.spp = 1; fs.rs.maxBounces = 13; fs.rs.rrBounceOffset = 3; fs.rs.rrInvMinTermProb = 0.95f; fs.rs.maxSampleValue = 10.0f;
    fs.rs.filterImportanceSampling = 1; fs.rs.jitteredSampling = 1; fs.rs.lightIntensityMultiplier = 1.0f; fs.rs.maxVolumeWalkLength = 7;
    fs.rs.metersPerSceneUnit = 1.0f; fs.rs.progressiveAccumulation = 1; fs.rs.domeLightCameraVisible = 1;
    fs.clear[0] = fs.clear[1] = fs.clear[2] = fs.clear[3] = 1.0f;
  }
  uint32_t rowBegin = 0, rowEnd = 0, rowStride = 1;
  int device = 0, stats = 0;
  for (int i = 3; i < argc; i++) {
    const char* a = argv[i];
    const char* v = i + 1 < argc ? argv[i + 1] : NULL;
    if (!strcmp(a, "--stats")) { stats = 1; continue; }
    if (!v) return fail("option without a value");
    i++;
    if (!strcmp(a, "--image-width")) fs.width = (uint32_t)atoi(v);
    else if (!strcmp(a, "--image-height")) fs.height = (uint32_t)atoi(v);
    else if (!strcmp(a, "--spp")) fs.rs.spp = (uint32_t)atoi(v);
    else if (!strcmp(a, "--max-bounces")) fs.rs.maxBounces = (uint32_t)atoi(v);
    else if (!strcmp(a, "--rr-bounce-offset")) fs.rs.rrBounceOffset = (uint32_t)atoi(v);
    else if (!strcmp(a, "--rr-inv-min-term-prob")) fs.rs.rrInvMinTermProb = (float)atof(v);
    else if (!strcmp(a, "--max-sample-value")) fs.rs.maxSampleValue = (float)atof(v);
    else if (!strcmp(a, "--next-event-estimation")) fs.rs.nextEventEstimation = atoi(v);
    else if (!strcmp(a, "--medium-stack-size")) fs.rs.mediumStackSize = (uint32_t)atoi(v);
    else if (!strcmp(a, "--depth-of-field")) fs.rs.depthOfField = atoi(v);
    else if (!strcmp(a, "--device")) device = atoi(v);
    else if (!strcmp(a, "--rows")) { if (sscanf(v, "%u:%u:%u", &rowBegin, &rowEnd, &rowStride) < 2) return fail("--rows wants begin:end[:stride]"); }
    else { fprintf(stderr, "gi_render: unknown option %s\n", a); return 2; }
  }

  GiCCameraDesc cam;
  rd_f32(&r, (float*)&cam, sizeof cam / 4);

  GiCScene* scene = NULL;
  if (!infoOnly) {
    if (giCInitialize(device) != GI_C_OK) { fprintf(stderr, "gi_render: giCInitialize: %s\n", giCGetLastError()); return 1; }
    scene = giCCreateScene();
    if (!scene) return fail("giCCreateScene failed");
  }

  /* textures */
  const uint32_t nTex = rd_u32(&r);
  if (nTex > (r.n - r.p) / 8u) return fail("corrupt .gscn file (texture count)");
  GiCTexture** textures = (GiCTexture**)calloc(nTex ? nTex : 1, sizeof *textures);
  if (!textures) return fail("out of memory");
  for (uint32_t t = 0; t < nTex && !r.bad; t++) {
    GiCTextureDesc td;
    td.width = rd_u32(&r); td.height = rd_u32(&r);
    td.rgba = (const float*)rd_take(&r, (size_t)td.width * td.height * 16);
    if (!infoOnly && td.rgba && !(textures[t] = giCCreateTexture(scene, &td))) { fprintf(stderr, "gi_render: giCCreateTexture: %s\n", giCGetLastError()); return 1; }
  }
  /* materials */
  const uint32_t nMat = rd_u32(&r);
  if (nMat > (r.n - r.p) / 8u) return fail("corrupt .gscn file (material count)");
  GiCMaterial** materials = (GiCMaterial**)calloc(nMat ? nMat : 1, sizeof *materials);
  if (!materials) return fail("out of memory");
  for (uint32_t m = 0; m < nMat && !r.bad; m++) {
    char* name = rd_str(&r);
    GiCMaterialDesc md;
    memset(&md, 0, sizeof md);
    md.klass = rd_u32(&r);
    const uint32_t nParams = rd_u32(&r);
    if (nParams != GI_C_MAT_PARAM_COUNT) return fail("material parameter block of another size");
    rd_f32(&r, md.p, nParams);
    if (!infoOnly && !(materials[m] = giCCreateMaterial(scene, name, &md))) { fprintf(stderr, "gi_render: giCCreateMaterial: %s\n", giCGetLastError()); return 1; }
    for (int slot = 0; slot < GI_C_TEX_SLOT_COUNT; slot++) {
      GiCTextureBinding b;
      memset(&b, 0, sizeof b);
      const int32_t tex = rd_i32(&r);
      b.wrapS = rd_i32(&r); b.wrapT = rd_i32(&r); b.channel = rd_i32(&r);
      rd_f32(&r, b.scale, 4); rd_f32(&r, b.bias, 4);
      const uint32_t hasXf = rd_u32(&r);
      float xf[6];
      rd_f32(&r, xf, 6);
      if (tex >= 0 && (uint32_t)tex < nTex && !infoOnly) {
        b.texture = textures[tex];
        if (giCSetMaterialTexture(materials[m], slot, &b) != GI_C_OK) { fprintf(stderr, "gi_render: giCSetMaterialTexture: %s\n", giCGetLastError()); return 1; }
        if (hasXf && giCSetMaterialTextureTransform(materials[m], slot, xf) != GI_C_OK) { fprintf(stderr, "gi_render: giCSetMaterialTextureTransform: %s\n", giCGetLastError()); return 1; }
      }
    }
    for (int slot = 0; slot < GI_C_TEX_SLOT_COUNT; slot++) {
      char* pv = rd_str(&r);
      if (pv && pv[0] && !infoOnly && giCSetMaterialPrimvarInput(materials[m], slot, pv) != GI_C_OK) return fail("giCSetMaterialPrimvarInput failed");
      free(pv);
    }
    free(name);
  }
  /* dome light */
  GiCDomeLight* dome = NULL;
  const uint32_t hasDome = rd_u32(&r);
  if (hasDome) {
    const int32_t tex = rd_i32(&r);
    float v[9];
    rd_f32(&r, v, 9);
    if (!infoOnly) {
      dome = giCCreateDomeLight(scene, "");
      if (tex >= 0 && (uint32_t)tex < nTex) giCSetDomeLightTexture(dome, textures[tex]);
      giCSetDomeLightRotation(dome, v); giCSetDomeLightBaseEmission(dome, v + 4); giCSetDomeLightDiffuseSpecular(dome, v[7], v[8]);
    }
  }
  /* meshes */
  const uint32_t nMesh = rd_u32(&r);
  if (nMesh > (r.n - r.p) / 8u) return fail("corrupt .gscn file (mesh count)");
  GiCMesh** meshes = (GiCMesh**)calloc(nMesh ? nMesh : 1, sizeof *meshes);
  if (!meshes) return fail("out of memory");
  uint64_t triangles = 0, instancedTriangles = 0;
  for (uint32_t m = 0; m < nMesh && !r.bad; m++) {
    char* name = rd_str(&r);
    GiCMeshDesc d;
    memset(&d, 0, sizeof d);
    d.vertexCount = rd_u32(&r); d.faceCount = rd_u32(&r);
    d.id = rd_i32(&r);
    const uint32_t flags = rd_u32(&r);
    const int32_t material = rd_i32(&r);
    d.maxFaceId = rd_u32(&r);
    d.isDoubleSided = (flags & 1u) != 0; d.isLeftHanded = (flags & 2u) != 0; d.name = name;
    float transform[16];
    rd_f32(&r, transform, 16);
    const uint32_t nInst = rd_u32(&r);
    const float* inst = (const float*)rd_take(&r, (size_t)nInst * 64);
    const int32_t* instIds = (flags & 16u) ? (const int32_t*)rd_take(&r, (size_t)nInst * 4) : NULL;
    d.vertices = (const GiCVertex*)rd_take(&r, (size_t)d.vertexCount * sizeof(GiCVertex));
    d.faces = (const GiCFace*)rd_take(&r, (size_t)d.faceCount * 12);
    d.faceIds = (flags & 8u) ? (const int32_t*)rd_take(&r, (size_t)d.faceCount * 4) : NULL;
    if (r.bad) break;
    triangles += d.faceCount; instancedTriangles += (uint64_t)d.faceCount * nInst;
    if (!infoOnly) {
      if (!(meshes[m] = giCCreateMesh(scene, &d))) { fprintf(stderr, "gi_render: giCCreateMesh: %s\n", giCGetLastError()); return 1; }
      giCSetMeshTransform(meshes[m], transform);
      giCSetMeshInstanceTransforms(meshes[m], nInst, inst);
      if (instIds) giCSetMeshInstanceIds(meshes[m], nInst, instIds);
    }
    for (int which = 0; which < 2; which++) { /* mesh primvars, then instancer primvars */
      const uint32_t nPv = rd_u32(&r);
      if (nPv > (r.n - r.p) / 16u) { r.bad = 1; break; }
      GiCPrimvarData* pv = (GiCPrimvarData*)calloc(nPv ? nPv : 1, sizeof *pv);
      char** names = (char**)calloc(nPv ? nPv : 1, sizeof *names);
      if (!pv || !names) return fail("out of memory");
      for (uint32_t k = 0; k < nPv && !r.bad; k++) {
        names[k] = rd_str(&r);
        pv[k].name = names[k]; pv[k].type = rd_i32(&r); pv[k].interpolation = rd_i32(&r);
        const uint32_t nFloats = rd_u32(&r);
        pv[k].data = rd_take(&r, (size_t)nFloats * 4); pv[k].dataSize = (uint64_t)nFloats * 4;
      }
      if (nPv && !r.bad && !infoOnly) {
        const int rc = which == 0 ? giCSetMeshPrimvars(meshes[m], nPv, pv) : giCSetMeshInstancerPrimvars(meshes[m], nPv, pv);
        if (rc != GI_C_OK) { fprintf(stderr, "gi_render: giCSetMesh*Primvars: %s\n", giCGetLastError()); return 1; }
      }
      for (uint32_t k = 0; k < nPv; k++) free(names[k]);
      free(names); free(pv);
    }
    if (!infoOnly) {
      if (material >= 0 && (uint32_t)material < nMat) giCSetMeshMaterial(meshes[m], materials[material]);
      giCSetMeshVisibility(meshes[m], (flags & 4u) != 0);
    }
    free(name);
  }
  /* analytic lights */
  uint32_t nLights[4] = {0, 0, 0, 0};
  void** lights[4] = {NULL, NULL, NULL, NULL};
  for (int kind = 0; kind < 4 && !r.bad; kind++) {
    static const uint32_t floats[4] = {11, 9, 16, 16};
    nLights[kind] = rd_u32(&r);
    if (nLights[kind] > (r.n - r.p) / (floats[kind] * 4u)) { r.bad = 1; break; }
    lights[kind] = (void**)calloc(nLights[kind] ? nLights[kind] : 1, sizeof(void*));
    if (!lights[kind]) return fail("out of memory");
    for (uint32_t l = 0; l < nLights[kind] && !r.bad; l++) {
      float v[16];
      rd_f32(&r, v, floats[kind]);
      if (infoOnly) continue;
      if (kind == 0) {
        GiCSphereLight* h = giCCreateSphereLight(scene);
        lights[kind][l] = h;
        giCSetSphereLightPosition(h, v); giCSetSphereLightBaseEmission(h, v + 3); giCSetSphereLightRadius(h, v[6], v[7], v[8]);
        giCSetSphereLightDiffuseSpecular(h, v[9], v[10]);
      } else if (kind == 1) {
        GiCDistantLight* h = giCCreateDistantLight(scene);
        lights[kind][l] = h;
        giCSetDistantLightDirection(h, v); giCSetDistantLightBaseEmission(h, v + 3); giCSetDistantLightAngle(h, v[6]);
        giCSetDistantLightDiffuseSpecular(h, v[7], v[8]);
      } else if (kind == 2) {
        GiCRectLight* h = giCCreateRectLight(scene);
        lights[kind][l] = h;
        giCSetRectLightOrigin(h, v); giCSetRectLightTangents(h, v + 3, v + 6); giCSetRectLightBaseEmission(h, v + 9);
        giCSetRectLightDimensions(h, v[12], v[13]); giCSetRectLightDiffuseSpecular(h, v[14], v[15]);
      } else {
        GiCDiskLight* h = giCCreateDiskLight(scene);
        lights[kind][l] = h;
        giCSetDiskLightOrigin(h, v); giCSetDiskLightTangents(h, v + 3, v + 6); giCSetDiskLightBaseEmission(h, v + 9);
        giCSetDiskLightRadius(h, v[12], v[13]); giCSetDiskLightDiffuseSpecular(h, v[14], v[15]);
      }
    }
  }
  const void* end = rd_take(&r, 4);
  if (r.bad || !end || memcmp(end, "END!", 4)) return fail("truncated or corrupt .gscn file");

  if (infoOnly) {
    printf("gscn v1: %u textures, %u materials, %u meshes, %llu triangles (%llu instanced), lights sphere/distant/rect/disk %u/%u/%u/%u, dome %u, "
           "settings %u (%ux%u spp %u bounces %u)\n", nTex, nMat, nMesh, (unsigned long long)triangles, (unsigned long long)instancedTriangles,
           nLights[0], nLights[1], nLights[2], nLights[3], hasDome, fs.has, fs.width, fs.height, fs.rs.spp, fs.rs.maxBounces);
    return 0;
  }

  /* one giCRender call; the colour AOV arrives in the render buffer's host memory (Gi.cpp:2492-2502) */
  GiCRenderBuffer* rb = giCCreateRenderBuffer(fs.width, fs.height, GI_C_FORMAT_FLOAT32_VEC4);
  if (!rb) { fprintf(stderr, "gi_render: giCCreateRenderBuffer: %s\n", giCGetLastError()); return 1; }
  GiCAovBinding binding;
  memset(&binding, 0, sizeof binding);
  binding.aovId = GI_C_AOV_COLOR;
  memcpy(binding.clearValue, fs.clear, 16);
  binding.renderBuffer = rb;
  GiCRenderParams p;
  memset(&p, 0, sizeof p);
  p.aovBindings = &binding; p.aovBindingCount = 1; p.camera = cam; p.domeLight = dome; p.renderSettings = fs.rs; p.scene = scene;
  p.rowBegin = rowBegin; p.rowEnd = rowEnd ? rowEnd : fs.height; p.rowStride = rowStride ? rowStride : 1;
  if (giCRender(&p) != GI_C_OK) { fprintf(stderr, "gi_render: giCRender: %s\n", giCGetLastError()); return 1; }
  if (stats) {
    GiCRenderStats st;
    if (giCGetRenderStats(scene, &st) == GI_C_OK)
      printf("render %.1f ms (bvh build %.1f ms, upload %.1f ms), %.1f Msamples/s, %llu samples, %llu segments, %u bvh8 nodes, %u triangles\n", st.renderMs,
             st.bvhBuildMs, st.uploadMs, (double)st.samples / (st.renderMs * 1e3), (unsigned long long)st.samples, (unsigned long long)st.segments,
             st.nodeCount, st.triangleCount);
  }
  const int rc = write_image(argv[2], (const float*)giCGetRenderBufferMem(rb), fs.width, fs.height);

  giCDestroyRenderBuffer(rb);
  for (uint32_t m = 0; m < nMesh; m++) giCDestroyMesh(meshes[m]);
  for (uint32_t m = 0; m < nMat; m++) giCDestroyMaterial(materials[m]);
  if (dome) giCDestroyDomeLight(dome);
  for (uint32_t t = 0; t < nTex; t++) giCDestroyTexture(textures[t]);
  for (uint32_t l = 0; l < nLights[0]; l++) giCDestroySphereLight(scene, (GiCSphereLight*)lights[0][l]);
  for (uint32_t l = 0; l < nLights[1]; l++) giCDestroyDistantLight(scene, (GiCDistantLight*)lights[1][l]);
  for (uint32_t l = 0; l < nLights[2]; l++) giCDestroyRectLight(scene, (GiCRectLight*)lights[2][l]);
  for (uint32_t l = 0; l < nLights[3]; l++) giCDestroyDiskLight(scene, (GiCDiskLight*)lights[3][l]);
  for (int kind = 0; kind < 4; kind++) free(lights[kind]);
  giCDestroyScene(scene);
  giCTerminate();
  free(meshes); free(materials); free(textures); free(data);
  return rc;
}
