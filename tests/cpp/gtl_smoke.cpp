// Drives the gi core through the reference's C++ API shape (include/gtl/gi/Gi.h): one quad + one emissive quad described as
// MaterialX strings, rendered at 32x18.  Prints a checksum; exits non-zero on any API failure.  Built by tests/test_gtl_shim.py.
#include <gtl/gi/Gi.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

using namespace gtl;

static GiMesh* quad(GiScene* scene, float z, float half, GiMaterial* mat, int id, const float* displayColor = nullptr)
{
  std::vector<GiVertex> v(4);
  const float p[4][2] = {{-half, -half}, {half, -half}, {half, half}, {-half, half}};
  for (int i = 0; i < 4; i++) {
    v[i] = GiVertex{{p[i][0], p[i][1], z}, 0.0f, {0, 0, 1}, 0.0f, {1, 0, 0}, 1.0f};
  }
  std::vector<GiFace> f = {{{0, 1, 2}}, {{0, 2, 3}}};
  std::vector<int> faceIds = {0, 0};
  std::vector<GiPrimvarData> primvars;
  if (displayColor) { // constant displayColor primvar, as hdGatling hands it over (mesh.cpp primvar sync)
    GiPrimvarData pv{"displayColor", GiPrimvarType::Vec3, GiPrimvarInterpolation::Constant, {}};
    pv.data.resize(12); memcpy(pv.data.data(), displayColor, 12);
    primvars.push_back(pv);
  }
  GiMeshDesc d{2, f, faceIds, id, true, false, "quad", 0, primvars, 4, v};
  GiMesh* m = giCreateMesh(scene, d);
  const float I[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  giSetMeshTransform(m, &I[0][0]);
  giSetMeshInstanceTransforms(m, 1, &I);
  int ids[1] = {0};
  giSetMeshInstanceIds(m, 1, ids);
  giSetMeshMaterial(m, mat);
  return m;
}

int main()
{
  std::vector<std::string> noPaths;
  GiInitParams init{"", "", noPaths, nullptr, ""};
  if (giInitialize(init) != GiStatus::Ok) { fprintf(stderr, "giInitialize failed\n"); return 2; }
  GiScene* scene = giCreateScene();
  // diffuseColor comes from a primvar reader (UsdPrimvarReader_float3 -> MaterialX geompropvalue): scene_data_lookup_float3
  const char* floorMtlx = "<materialx version=\"1.38\"><geompropvalue name=\"dc\" type=\"color3\"><input name=\"geomprop\" type=\"string\" value=\"displayColor\" /></geompropvalue>"
                          "<UsdPreviewSurface name=\"SR\" type=\"surfaceshader\">"
                          "<input name=\"diffuseColor\" type=\"color3\" nodename=\"dc\" /><input name=\"roughness\" type=\"float\" value=\"0.4\" />"
                          "</UsdPreviewSurface></materialx>";
  const char* lampMtlx = "<materialx version=\"1.39\"><open_pbr_surface name=\"L\" type=\"surfaceshader\">"
                         "<input name=\"emission_luminance\" type=\"float\" value=\"5.0\" /><input name=\"emission_color\" type=\"color3\" value=\"1, 0.9, 0.8\" />"
                         "</open_pbr_surface></materialx>";
  GiMaterial* floorMat = giCreateMaterialFromMtlxStr(scene, "floor", floorMtlx);
  GiMaterial* lampMat = giCreateMaterialFromMtlxStr(scene, "lamp", lampMtlx);
  GiMaterial* unsupported = giCreateMaterialFromMtlxStr(scene, "x", "<materialx><disney_principled name=\"s\"/></materialx>"); // no closed form: hdGatling's fallback material takes over
  GiMaterial* translated = giCreateMaterialFromMtlxStr(scene, "y", "<materialx><standard_surface name=\"s\"/></materialx>");  // read onto the OpenPBR closed forms
  if (!floorMat || !lampMat || unsupported || !translated) { fprintf(stderr, "material creation mismatch\n"); return 3; }
  giDestroyMaterial(translated);
  const float green[3] = {0.1f, 0.8f, 0.1f};
  GiMesh* a = quad(scene, 0.0f, 2.0f, floorMat, 1, green);
  GiMesh* b = quad(scene, 1.5f, 0.4f, lampMat, 2);
  // optional third quad whose diffuseColor is an image file (UsdUVTexture node -> giCCreateTextureFromFile inside the shim)
  GiMaterial* texMat = nullptr; GiMesh* c = nullptr;
  if (const char* png = getenv("GTL_SMOKE_PNG")) {
    const std::string texMtlx = std::string("<materialx version=\"1.38\"><UsdUVTexture name=\"img\" type=\"multioutput\"><input name=\"file\" type=\"filename\" value=\"") + png +
                                "\" /><input name=\"wrapS\" type=\"string\" value=\"repeat\" /></UsdUVTexture><UsdPreviewSurface name=\"T\" type=\"surfaceshader\">"
                                "<input name=\"diffuseColor\" type=\"color3\" nodename=\"img\" output=\"rgb\" /></UsdPreviewSurface></materialx>";
    texMat = giCreateMaterialFromMtlxStr(scene, "textured", texMtlx.c_str());
    if (!texMat) { fprintf(stderr, "textured material not created\n"); return 8; }
    c = quad(scene, 0.5f, 0.6f, texMat, 3);
  }
  GiRenderBuffer* rb = giCreateRenderBuffer(32, 18, GiRenderBufferFormat::Float32Vec4);
  GiRenderParams rp{};
  GiAovBinding bind{GiAovId::Color, {0}, rb};
  const float clear[4] = {0.1f, 0.1f, 0.1f, 1.0f};
  memcpy(bind.clearValue, clear, 16);
  rp.aovBindings.push_back(bind);
  rp.camera = GiCameraDesc{{0, -3, 3}, {0, 0.7071068f, -0.7071068f}, {0, 0.7071068f, 0.7071068f}, 0.9f, 0, 0, 5.0f, 0.1f, 100.0f, 0};
  rp.domeLight = nullptr;
  rp.renderSettings = GiRenderSettings{false, false, true, true, 0, true, 1.0f, 6, 10.0f, 7, 0, 1.0f, false, true, 3, 0.95f, 8, 0};
  rp.scene = scene;
  if (giRender(rp) != GiStatus::Ok) { fprintf(stderr, "giRender failed\n"); return 4; }
  const float* px = (const float*)giGetRenderBufferMem(rb);
  double sum = 0, red = 0, grn = 0, blu = 0; int lit = 0, bluish = 0;
  for (int i = 0; i < 32 * 18; i++) { sum += px[4 * i] + px[4 * i + 1] + px[4 * i + 2]; red += px[4 * i]; grn += px[4 * i + 1]; blu += px[4 * i + 2]; bluish += px[4 * i + 2] > 2.0f * px[4 * i] && px[4 * i + 2] > 2.0f * px[4 * i + 1]; lit += px[4 * i + 1] > 0.11f; if (px[4 * i + 3] != 1.0f) return 5; }
  // the lamp is reddish (1, .9, .8): without the green primvar the image is red-heavy
  if (!(grn > 1.05 * red)) { fprintf(stderr, "displayColor primvar not applied (r=%f g=%f)\n", red, grn); return 7; }
  if (c && bluish < 5) { fprintf(stderr, "image-driven diffuseColor not applied (bluish pixels: %d)\n", bluish); return 9; }
  (void)blu;
  unsigned long long hash = 1469598103934665603ull; // FNV-1a over the image bytes: equal hashes <=> bit-identical images (multi-device test)
  for (size_t i = 0; i < (size_t)32 * 18 * 16; i++) { hash ^= ((const unsigned char*)px)[i]; hash *= 1099511628211ull; }
  printf("gtl_smoke ok sum=%.6f lit=%d bluish=%d hash=%016llx\n", sum, lit, bluish, hash);
  giDestroyMesh(a); giDestroyMesh(b); if (c) giDestroyMesh(c);
  giDestroyMaterial(floorMat); giDestroyMaterial(lampMat); if (texMat) giDestroyMaterial(texMat);
  giDestroyRenderBuffer(rb); giDestroyScene(scene); giTerminate();
  return (lit >= 10 && sum > 10.0) ? 0 : 6;
}
