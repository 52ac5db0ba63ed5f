// gtl/gi/Gi.h -- the C++ face of the MI355X-native gi core.
//
// hdGatling includes <gtl/gi/Gi.h> and calls 52 free functions in namespace gtl over opaque handles and a few POD
// descriptions (reference interface: /root/reference/src/gi/gtl/gi/Gi.h:32-261).  A drop-in must offer exactly those names,
// argument orders and struct layouts, so this header declares them -- organised the way THIS implementation is built: every
// entry is a thin forwarder (gatling_amd/csrc/gtl_shim.cpp) onto the C ABI include/gi_c.h, the four analytic light types come
// out of one table (GTL_GI_LIGHT_TYPES), and each declaration names the reference line it stands in for and what differs.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <string_view>
#include <unordered_map>
#include <variant>
#include <vector>

#include <gtl/gb/ParamTypes.h>   // GbVec2f/3f/4f, GbColor, GbTextureAsset{absPath, isSrgb} -- as the reference does (Gi.h:28)

namespace gtl
{
  // ---- handles (Gi.h:58-68): opaque here as there; each wraps one giC* handle -------------------------------------------
  struct GiAsset; struct GiScene; struct GiMaterial; struct GiMesh; struct GiRenderBuffer; struct GiDomeLight; struct GiShaderCache;

  // ---- enumerations -------------------------------------------------------------------------------------------------------
  enum class GiStatus { Ok, Error };                                                              // Gi.h:32
  enum class GiAovId                                                                              // Gi.h:36-56 == GiCAovId
  { Color = 0, Normal, NEE, Barycentrics, Texcoords, Bounces, ClockCycles, Opacity, Tangents, Bitangents, ThinWalled, ObjectId, Depth,
    FaceId, InstanceId, DoubleSided, Albedo, COUNT };
  enum class GiRenderBufferFormat { Int32, Float32, Float32Vec4 };                                // Gi.h:70-75
  enum class GiPrimvarType { Float, Vec2, Vec3, Vec4, Int, Int2, Int3, Int4 };                    // Gi.h:76-79
  enum class GiPrimvarInterpolation { Constant, Instance, Uniform, Vertex, COUNT };               // Gi.h:81-84
  constexpr static const uint32_t GI_MAX_AOV_COMP_SIZE = 16;                                      // Gi.h:34

  // ---- plain descriptions; layouts are part of the contract (static_asserts against the C ABI live in gtl_shim.cpp) --------
  struct GiPrimvarData { std::string name; GiPrimvarType type; GiPrimvarInterpolation interpolation; std::vector<uint8_t> data; }; // Gi.h:86-92
  struct GiVertex { float pos[3]; float u; float norm[3]; float v; float tangent[3]; float bitangentSign; };   // Gi.h:110-118, 48 B == GiCVertex
  struct GiFace { uint32_t v_i[3]; };                                                                          // Gi.h:120-122
  struct GiCameraDesc                                                                                          // Gi.h:94-108 == GiCCameraDesc
  { float position[3], forward[3], up[3]; float vfov, fStop, focusDistance, focalLength, clipStart, clipEnd, exposure; };
  struct GiMeshDesc                                                                                            // Gi.h:124-137; arrays are copied by giCreateMesh
  { // hdGatling fills this with designated initialisers and leaves faceCount / vertexCount at zero (mesh.cpp:1092-1102): like the reference
    // (Gi.cpp:628), giCreateMesh takes the counts from the vectors and ignores the two count fields
    uint32_t faceCount; const std::vector<GiFace>& faces; const std::vector<int>& faceIds;
    int id; bool isDoubleSided; bool isLeftHanded; const char* name; uint32_t maxFaceId;
    const std::vector<GiPrimvarData>& primvars; // float primvars feed material inputs bound to them by name (scene_data_lookup_*)
    uint32_t vertexCount; const std::vector<GiVertex>& vertices;
  };
  struct GiRenderSettings                                                                                      // Gi.h:139-159 (bools widen to int32 in GiCRenderSettings)
  {
    bool clippingPlanes, depthOfField, domeLightCameraVisible, filterImportanceSampling; float frame; bool jitteredSampling;
    float lightIntensityMultiplier; uint32_t maxBounces; float maxSampleValue; uint32_t maxVolumeWalkLength, mediumStackSize;
    float metersPerSceneUnit; bool nextEventEstimation, progressiveAccumulation; uint32_t rrBounceOffset; float rrInvMinTermProb;
    uint32_t spp; float time;
  };
  struct GiAovBinding { GiAovId aovId; uint8_t clearValue[GI_MAX_AOV_COMP_SIZE]; GiRenderBuffer* renderBuffer; };  // Gi.h:161-166
  struct GiRenderParams                                                                                            // Gi.h:168-175
  { std::vector<GiAovBinding> aovBindings; GiCameraDesc camera; GiDomeLight* domeLight; GiRenderSettings renderSettings; GiScene* scene; };
  struct GiInitParams // Gi.h:177-184.  Every field is accepted and unused: kernels are precompiled HIP, there is no shader generation, MDL or MaterialX library
  { std::string_view shaderPath, mdlRuntimePath; const std::vector<std::string>& mdlSearchPaths; const std::shared_ptr<void> mtlxStdLib; std::string mtlxCustomNodesPath; };

  class GiAssetReader // Gi.h:186-194; recorded by giRegisterAssetReader (image decoding stays outside this core, see giCCreateTexture)
  {
  public:
    virtual ~GiAssetReader() = default;
    virtual GiAsset* open(const char* path) = 0;
    virtual size_t size(const GiAsset* asset) const = 0;
    virtual void* data(const GiAsset* asset) const = 0;
    virtual void close(GiAsset* asset) = 0;
  };

  // Gi.h:196-197: what hdGatling's _TranslateMaterialParameters produces (materialNetworkCompiler.cpp:533-612)
  using GiMaterialParameterValue = std::variant<bool, int, float, GbVec2f, GbVec3f, GbVec4f, GbColor, GbTextureAsset>;
  using GiMaterialParameters = std::unordered_map<std::string, GiMaterialParameterValue>;

  // ---- process (Gi.h:199-201) -----------------------------------------------------------------------------------------------
  GiStatus giInitialize(const GiInitParams&);   // -> giCInitialize(HIP device 0 or $GATLING_DEVICE); Error without a GPU: no CPU fallback
  void giTerminate();                           // -> giCTerminate
  void giRegisterAssetReader(GiAssetReader*);

  // ---- scene and frame (Gi.h:222-225) -----------------------------------------------------------------------------------------
  GiScene* giCreateScene();                     // -> giCCreateScene
  void giDestroyScene(GiScene*);                // -> giCDestroyScene
  GiStatus giRender(const GiRenderParams&);     // -> giCRender: blocks until every bound AOV is complete in host memory (Gi.cpp:2492-2502)

  // ---- materials (Gi.h:203-207).  The MaterialX string is scanned for a UsdPreviewSurface / open_pbr_surface node: constant inputs
  // become the GiCMaterialDesc parameter block, inputs fed by a primvar reader become scene-data bindings; everything else is nullptr,
  // which hdGatling answers with its default material (mesh.cpp:598-603).
  GiMaterial* giCreateMaterialFromMtlxStr(GiScene*, const char* name, const char* mtlxSrc);
  // doc is a MaterialX::DocumentPtr.  gtl_shim_mtlx.cpp (built when the MaterialX headers are found) serialises it with
  // MaterialX::writeToXmlString and feeds the same scanner; without that translation unit the call fails with nullptr and a message.
  GiMaterial* giCreateMaterialFromMtlxDoc(GiScene*, const char* name, const std::shared_ptr<void> doc);
  using GtlMtlxDocToXml = std::string (*)(const std::shared_ptr<void>& doc);   // [ext] MaterialX::DocumentPtr -> XML text
  void gtlRegisterMtlxDocSerializer(GtlMtlxDocToXml);                           // [ext] called by gtl_shim_mtlx.cpp's static initialiser
  // [ext] C-linkage doors to the MaterialX reader for harnesses without C++ (tests/test_mtlx_parity.py): `gtlCreateMaterialFromMtlxStrC(GiCScene*, name, xml)`
  // = giCreateMaterialFromMtlxStr returning the C handle; `gtlMaterialDescFromMtlxStrC(xml, GiCMaterialDesc*)` = the parameter block alone (no device needed).
  // There is no MDL compiler here.  The parameters hdGatling hands over by name are mapped onto the closed-form blocks: the OmniPBR
  // family shipped with the reference (src/gi/mdl/OmniPBR*.mdl) -> the UsdPreviewSurface block, UsdPreviewSurface / open_pbr_surface
  // spelled parameters directly; texture assets are decoded and bound.  Modules none of whose parameters are recognised: nullptr.
  GiMaterial* giCreateMaterialFromMdlFile(GiScene*, const char* name, const char* filePath, const char* subIdentifier,
                                          const GiMaterialParameters& params = {});
  void giDestroyMaterial(GiMaterial*);

  // ---- meshes (Gi.h:209-217): setters only mark the scene dirty, the BVH is rebuilt by the next giRender -------------------------
  GiMesh* giCreateMesh(GiScene*, const GiMeshDesc&);                                      // -> giCCreateMesh + giCSetMeshPrimvars
  void giSetMeshTransform(GiMesh*, const float* mat4x4);                                  // row-major, USD row-vector convention (Gi.cpp:641-650)
  void giSetMeshInstanceTransforms(GiMesh*, uint32_t count, const float (*transforms)[4][4]);
  void giSetMeshInstancerPrimvars(GiMesh*, const std::vector<GiPrimvarData>&);            // -> giCSetMeshInstancerPrimvars
  void giSetMeshInstanceIds(GiMesh*, uint32_t count, int* ids);
  void giSetMeshMaterial(GiMesh*, GiMaterial*);
  void giSetMeshVisibility(GiMesh*, bool visible);
  void giDestroyMesh(GiMesh*);

  // ---- analytic lights (Gi.h:227-251): four types with one life cycle; X(Type, own setters...) -------------------------------------
#define GTL_GI_LIGHT_TYPES(X) X(Sphere) X(Distant) X(Rect) X(Disk)
#define GTL_GI_DECLARE_LIGHT(Type) \
  struct Gi##Type##Light; \
  Gi##Type##Light* giCreate##Type##Light(GiScene*);                 /* -> giCCreate<Type>Light: dense store, swap-remove on destroy */ \
  void giDestroy##Type##Light(GiScene*, Gi##Type##Light*); \
  void giSet##Type##LightBaseEmission(Gi##Type##Light*, float* rgb); \
  void giSet##Type##LightDiffuseSpecular(Gi##Type##Light*, float diffuse, float specular);
  GTL_GI_LIGHT_TYPES(GTL_GI_DECLARE_LIGHT)
#undef GTL_GI_DECLARE_LIGHT
  void giSetSphereLightPosition(GiSphereLight*, float* position);
  void giSetSphereLightRadius(GiSphereLight*, float radiusX, float radiusY, float radiusZ);    // ellipsoid area: Knud-Thomsen (Gi.cpp:2635-2651)
  void giSetDistantLightDirection(GiDistantLight*, float* direction);
  void giSetDistantLightAngle(GiDistantLight*, float angle);                                   // invPdf = 2 pi (1 - cos(angle / 2)) (Gi.cpp:2723-2735)
  void giSetRectLightOrigin(GiRectLight*, float* origin);
  void giSetRectLightTangents(GiRectLight*, float* t0, float* t1);                             // stored octahedral-encoded; normal = t1 x t0
  void giSetRectLightDimensions(GiRectLight*, float width, float height);
  void giSetDiskLightOrigin(GiDiskLight*, float* origin);
  void giSetDiskLightTangents(GiDiskLight*, float* t0, float* t1);
  void giSetDiskLightRadius(GiDiskLight*, float radiusX, float radiusY);

  // ---- dome light (Gi.h:253-257): .hdr / .pfm / .png / baseline .jpg files are decoded in-library, other formats arrive through giCSetDomeLightTexture ----
  GiDomeLight* giCreateDomeLight(GiScene*, const char* filePath);
  void giDestroyDomeLight(GiDomeLight*);
  void giSetDomeLightRotation(GiDomeLight*, float* quat);
  void giSetDomeLightBaseEmission(GiDomeLight*, float* rgb);
  void giSetDomeLightDiffuseSpecular(GiDomeLight*, float diffuse, float specular);

  // ---- render buffers (Gi.h:259-261): pinned host memory, valid from creation to destruction -----------------------------------------
  GiRenderBuffer* giCreateRenderBuffer(uint32_t width, uint32_t height, GiRenderBufferFormat format);
  void giDestroyRenderBuffer(GiRenderBuffer*);
  void* giGetRenderBufferMem(GiRenderBuffer*);
}
