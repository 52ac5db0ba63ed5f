// Compiles -- with nothing but -Iinclude -- the statements hdGatling makes against the gi boundary, in hdGatling's own spelling, so a
// header that drifts from the reference interface fails HERE instead of in a USD build we cannot run:
//   materialNetworkCompiler.cpp:548-601  building GiMaterialParameters (GbVec2f/3f/4f, GbColor, GbTextureAsset{ path, isSrgb })
//   materialNetworkCompiler.cpp:664      giCreateMaterialFromMdlFile(scene, id, fileUri, subIdentifier, params)
//   materialNetworkCompiler.cpp:685      giCreateMaterialFromMtlxDoc(scene, id, mx::DocumentPtr)
//   mesh.cpp:1092-1104                   GiMeshDesc by designated initialisers WITHOUT faceCount / vertexCount, giCreateMesh
//   rendererPlugin.cpp:64-72             GiInitParams by designated initialisers, giInitialize
//   renderPass.cpp:295, renderDelegate.cpp:128-137, light.cpp, renderBuffer.cpp: the remaining calls
// With a GPU (argv[1] == "run") it also runs them: an OmniPBR-parameterised material and a MaterialX-document material must come back
// non-null and colour the image (red / green halves), and the mesh built without counts must be hit.
#include <gtl/gi/Gi.h>
#include <MaterialXFormat/XmlIo.h>   // tests/cpp/mock_mtlx (or a real MaterialX)

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace mx = MaterialX;
using namespace gtl;

static GiMaterialParameters translateParameters(const std::string& texPath)
{
  GiMaterialParameters giParams;
  std::string name = "enable_emission";
  giParams[name] = false;
  giParams["uv_space_index"] = 0;
  giParams["reflection_roughness_constant"] = 0.6f;
  giParams["texture_scale"] = GbVec2f{ 1.0f, 1.0f };
  giParams["some_vector"] = GbVec3f{ 0.0f, 1.0f, 0.0f };
  giParams["some_vec4"] = GbVec4f{ 0.0f, 1.0f, 0.0f, 1.0f };
  giParams["diffuse_color_constant"] = GbColor{ 0.9f, 0.05f, 0.05f };
  bool isSrgb = true;
  if (!texPath.empty()) giParams["diffuse_texture"] = GbTextureAsset{ texPath, isSrgb };
  return giParams;
}

int main(int argc, char** argv)
{
  const bool run = argc > 1 && !strcmp(argv[1], "run");
  std::vector<std::string> mdlSearchPaths;
  std::string shaderPath = "shaders", resourcePath = ".", mtlxCustomNodesPath = "./mtlx";
  std::shared_ptr<mx::Document> mtlxStdLib = mx::createDocument();
  GiInitParams params = {
    .shaderPath = shaderPath.c_str(),
    .mdlRuntimePath = resourcePath.c_str(),
    .mdlSearchPaths = mdlSearchPaths,
    .mtlxStdLib = mtlxStdLib,
    .mtlxCustomNodesPath = mtlxCustomNodesPath
  };
  if (!run) { printf("hdgatling_mimic compiled (pass 'run' on a GPU box)\n"); return 0; }
  if (giInitialize(params) != GiStatus::Ok) return 2;
  GiScene* scene = giCreateScene();

  GiMaterialParameters mdlParams = translateParameters(argc > 2 ? argv[2] : "");
  std::string fileUri = "/opt/gatling/mdl/OmniPBR.mdl", subIdentifier = "OmniPBR";
  GiMaterial* red = giCreateMaterialFromMdlFile(scene, "/World/Looks/red", fileUri.c_str(), subIdentifier.c_str(), mdlParams);
  GiMaterial* unknownMdl = giCreateMaterialFromMdlFile(scene, "/World/Looks/x", "/tmp/custom.mdl", "custom", GiMaterialParameters{{"frobnicate", 1.0f}});

  mx::DocumentPtr doc = mx::createDocument();   // what HdMtlxCreateMtlxDocumentFromHdNetwork emits: upstream nodes in a nodegraph
  doc->xml = "<?xml version=\"1.0\"?><materialx version=\"1.38\"><nodegraph name=\"NG_green\">"
             "<constant name=\"c\" type=\"color3\"><input name=\"value\" type=\"color3\" value=\"0.05, 0.9, 0.05\" /></constant>"
             "<output name=\"out\" type=\"color3\" nodename=\"c\" /></nodegraph>"
             "<UsdPreviewSurface name=\"SR_green\" type=\"surfaceshader\"><input name=\"diffuseColor\" type=\"color3\" nodegraph=\"NG_green\" output=\"out\" />"
             "<input name=\"roughness\" type=\"float\" value=\"0.7\" /></UsdPreviewSurface>"
             "<surfacematerial name=\"green\" type=\"material\"><input name=\"surfaceshader\" type=\"surfaceshader\" nodename=\"SR_green\" /></surfacematerial></materialx>";
  GiMaterial* green = giCreateMaterialFromMtlxDoc(scene, "/World/Looks/green", doc);
  if (!red || !green || unknownMdl) { fprintf(stderr, "material creation: red %p green %p unknown %p\n", (void*)red, (void*)green, (void*)unknownMdl); return 3; }

  // two quads side by side facing +z, created the way mesh.cpp does (no counts in the description)
  std::vector<GiMesh*> meshes;
  for (int side = 0; side < 2; side++) {
    const float x0 = side ? 0.0f : -2.0f, x1 = side ? 2.0f : 0.0f;
    std::vector<GiVertex> giVertices = {
      GiVertex{{x0, -2.0f, 0.0f}, 0.0f, {0, 0, 1}, 0.0f, {1, 0, 0}, 1.0f}, GiVertex{{x1, -2.0f, 0.0f}, 1.0f, {0, 0, 1}, 0.0f, {1, 0, 0}, 1.0f},
      GiVertex{{x1, 2.0f, 0.0f}, 1.0f, {0, 0, 1}, 1.0f, {1, 0, 0}, 1.0f}, GiVertex{{x0, 2.0f, 0.0f}, 0.0f, {0, 0, 1}, 1.0f, {1, 0, 0}, 1.0f}};
    struct SubMeshData { std::vector<GiFace> faces; std::vector<int> faceIds; int maxFaceId = -1; } subMesh;
    subMesh.faces = {GiFace{{0, 1, 2}}, GiFace{{0, 2, 3}}}; subMesh.faceIds = {0, 0}; subMesh.maxFaceId = 0;
    std::vector<GiPrimvarData> secondaryPrimvars;
    bool isLeftHanded = false;
    GiMeshDesc desc = {
      .faces = subMesh.faces,
      .faceIds = subMesh.faceIds,
      .id = side + 1,
      .isDoubleSided = true,
      .isLeftHanded = isLeftHanded,
      .name = "/World/quad",
      .maxFaceId = (uint32_t) subMesh.maxFaceId,
      .primvars = secondaryPrimvars,
      .vertices = giVertices,
    };
    GiMesh* m = giCreateMesh(scene, desc);
    const float I[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    giSetMeshTransform(m, &I[0][0]);
    giSetMeshInstanceTransforms(m, 1, &I);
    std::vector<int> instanceIds = {0};
    giSetMeshInstanceIds(m, (uint32_t)instanceIds.size(), instanceIds.data());
    giSetMeshMaterial(m, side ? green : red);
    meshes.push_back(m);
  }
  const uint32_t W = 32, H = 16;
  GiRenderBuffer* rb = giCreateRenderBuffer(W, H, GiRenderBufferFormat::Float32Vec4);
  GiAovBinding color{GiAovId::Color, {0}, rb};
  const float white[4] = {1, 1, 1, 1}; memcpy(color.clearValue, white, 16);
  GiRenderParams rp{{color}, GiCameraDesc{{0, 0, 5}, {0, 0, -1}, {0, 1, 0}, 0.6f, 0.0f, 5.0f, 0.05f, 0.1f, 100.0f, 0.0f}, nullptr,
                    GiRenderSettings{false, false, true, true, 0.0f, true, 1.0f, 4, 10.0f, 8, 0, 1.0f, false, false, 3, 0.95f, 16, 0.0f}, scene};
  if (giRender(rp) != GiStatus::Ok) return 4;
  const float* px = static_cast<const float*>(giGetRenderBufferMem(rb));
  const float* l = px + (8 * W + 8) * 4; const float* r = px + (8 * W + 24) * 4;
  printf("left %.3f %.3f %.3f  right %.3f %.3f %.3f\n", l[0], l[1], l[2], r[0], r[1], r[2]);
  const bool ok = l[0] > 3.0f * l[1] && l[0] > 0.2f && r[1] > 3.0f * r[0] && r[1] > 0.2f;
  for (GiMesh* m : meshes) giDestroyMesh(m);
  giDestroyMaterial(red); giDestroyMaterial(green);
  giDestroyRenderBuffer(rb); giDestroyScene(scene); giTerminate();
  if (!ok) { fprintf(stderr, "materials did not reach the image\n"); return 5; }
  printf("hdgatling_mimic ok\n");
  return 0;
}
