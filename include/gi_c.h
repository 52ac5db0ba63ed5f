/*
 * gi_c.h -- C ABI of the MI355X-native render core ("libgatling_gi.so").
 *
 * This is the drop-in boundary for the path BASELINE.json names: the `gi` render loop of
 * pablode/gatling.  In the reference that boundary is the C++ API of
 * /root/reference/src/gi/gtl/gi/Gi.h:199-261 (free functions over opaque structs, linked as a static
 * library into hdGatling.so).  Every entry point below cites the Gi.h declaration it replaces; argument
 * meaning, ownership and error behaviour follow the reference:
 *   - the library owns every handle; callers destroy explicitly;
 *   - giCCreateMesh COPIES all vertex/face data before returning (Gi.cpp:620-638);
 *   - transforms are row-major 4x4 floats in USD's row-vector convention (Gi.cpp:641-658);
 *   - setters only mark the scene dirty; BVH build + upload happen in the next giCRender
 *     (hdGatling/renderDelegate.cpp:208-213);
 *   - giCRender blocks until the AOVs are complete in host memory (Gi.cpp:2492-2502);
 *   - status returns: GI_C_OK / GI_C_ERROR; creators return NULL on failure; no exceptions cross.
 * Differences, all forced by plain-C types: std::vector / std::string_view arguments become
 * pointer + count; MaterialX/MDL material creation (Gi.h:204-206) becomes giCCreateMaterial with a
 * closed-form parameter block (DESIGN.md "Materials"); the gtl:: C++ shim on top of this header lives in
 * include/gtl/gi/Gi.h.  Extensions that the reference does not have are marked [ext].
 *
 * The library is HIP-only: if no gfx950 device / HIP runtime is available giCInitialize fails
 * (there is no CPU fallback).
 */
#ifndef GATLING_GI_C_H
#define GATLING_GI_C_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GI_C_OK 0
#define GI_C_ERROR 1

#define GI_C_MAX_AOV_COMP_SIZE 16 /* Gi.h:32 */

/* Gi.h:36-56 */
typedef enum GiCAovId {
  GI_C_AOV_COLOR = 0, GI_C_AOV_NORMAL, GI_C_AOV_NEE, GI_C_AOV_BARYCENTRICS, GI_C_AOV_TEXCOORDS, GI_C_AOV_BOUNCES,
  GI_C_AOV_CLOCK_CYCLES, GI_C_AOV_OPACITY, GI_C_AOV_TANGENTS, GI_C_AOV_BITANGENTS, GI_C_AOV_THIN_WALLED,
  GI_C_AOV_OBJECT_ID, GI_C_AOV_DEPTH, GI_C_AOV_FACE_ID, GI_C_AOV_INSTANCE_ID, GI_C_AOV_DOUBLE_SIDED, GI_C_AOV_ALBEDO,
  GI_C_AOV_COUNT
} GiCAovId;

/* Gi.h:70-75 */
typedef enum GiCRenderBufferFormat { GI_C_FORMAT_INT32 = 0, GI_C_FORMAT_FLOAT32 = 1, GI_C_FORMAT_FLOAT32_VEC4 = 2 } GiCRenderBufferFormat;

typedef struct GiCScene GiCScene;
typedef struct GiCMaterial GiCMaterial;
typedef struct GiCMesh GiCMesh;
typedef struct GiCSphereLight GiCSphereLight;
typedef struct GiCDistantLight GiCDistantLight;
typedef struct GiCRectLight GiCRectLight;
typedef struct GiCDiskLight GiCDiskLight;
typedef struct GiCDomeLight GiCDomeLight;
typedef struct GiCRenderBuffer GiCRenderBuffer;

/* Gi.h:96-108 (same layout) */
typedef struct GiCCameraDesc {
  float position[3];
  float forward[3];
  float up[3];
  float vfov;
  float fStop;
  float focusDistance;
  float focalLength;
  float clipStart;
  float clipEnd;
  float exposure;
} GiCCameraDesc;

/* Gi.h:110-118 (same layout, 48 bytes) */
typedef struct GiCVertex {
  float pos[3];
  float u;
  float norm[3];
  float v;
  float tangent[3];
  float bitangentSign;
} GiCVertex;

/* Gi.h:120-122 */
typedef struct GiCFace { uint32_t v_i[3]; } GiCFace;

/* Gi.h:124-137 with vectors flattened to pointer+count; GiMeshDesc.primvars arrive through giCSetMeshPrimvars */
typedef struct GiCMeshDesc {
  uint32_t faceCount;
  const GiCFace* faces;
  const int32_t* faceIds; /* faceCount entries or NULL */
  int32_t id;
  int32_t isDoubleSided;
  int32_t isLeftHanded;
  const char* name;
  uint32_t maxFaceId;
  uint32_t vertexCount;
  const GiCVertex* vertices;
} GiCMeshDesc;

/* Gi.h:139-159 (bools widened to int32 for a stable ABI) */
typedef struct GiCRenderSettings {
  int32_t clippingPlanes;
  int32_t depthOfField;
  int32_t domeLightCameraVisible;
  int32_t filterImportanceSampling;
  float frame;
  int32_t jitteredSampling;
  float lightIntensityMultiplier;
  uint32_t maxBounces;
  float maxSampleValue;
  uint32_t maxVolumeWalkLength;
  uint32_t mediumStackSize;
  float metersPerSceneUnit;
  int32_t nextEventEstimation;
  int32_t progressiveAccumulation;
  uint32_t rrBounceOffset;
  float rrInvMinTermProb;
  uint32_t spp;
  float time;
} GiCRenderSettings;

/* Gi.h:161-166 */
typedef struct GiCAovBinding {
  int32_t aovId; /* GiCAovId */
  uint8_t clearValue[GI_C_MAX_AOV_COMP_SIZE];
  GiCRenderBuffer* renderBuffer;
} GiCAovBinding;

/* Gi.h:168-175 */
typedef struct GiCRenderParams {
  const GiCAovBinding* aovBindings;
  uint32_t aovBindingCount;
  GiCCameraDesc camera;
  GiCDomeLight* domeLight;
  GiCRenderSettings renderSettings;
  GiCScene* scene;
  /* Non-colour AOVs: Normal, Barycentrics, Texcoords, Opacity, Tangents, Bitangents, ThinWalled, ObjectId, Depth, FaceId, InstanceId,
   * DoubleSided and Albedo are produced by a primary-hit pass (vec3 AOVs need Float32Vec4 buffers, Depth Float32, ids Int32 --
   * Gi.cpp:302-316); NEE, Bounces and ClockCycles follow whole paths and are filled by the colour pass (which then runs even without a
   * Color binding).  ClockCycles is a Turbo heat map, normalised to the frame maximum, alpha 255 (_EncodeRenderBufferAsHeatmap,
   * Gi.cpp:327-343) of a deterministic cost proxy -- the ray segments traced for the pixel -- instead of the shader clock. */
  /* [ext] multi-GPU sharding: render only image rows [rowBegin,rowEnd) of the full image whose size is the
   * render buffers' size; rowEnd == 0 means "all rows".  RNG streams use the global pixel index so an N-way
   * split is bit-identical to the single-GPU image. */
  uint32_t rowBegin;
  uint32_t rowEnd;
  /* [ext] rows rowBegin, rowBegin + rowStride, ... below rowEnd (0 and 1 mean every row).  Rank r of N rendering
   * (rowBegin = r, rowEnd = height, rowStride = N) gets an even share of cheap and expensive image regions: on the cornell frame
   * contiguous eighths differ by 0.61x .. 1.19x of the mean cost. */
  uint32_t rowStride;
} GiCRenderParams;

/* Closed-form material classes: replaces MaterialX->MDL->GLSL codegen (src/mc, GlslShaderGen) */
#define GI_C_MAT_DIFFUSE 0u             /* config C1 "diffuse only" model */
#define GI_C_MAT_USD_PREVIEW_SURFACE 1u /* diffuse + GGX specular + clearcoat */
#define GI_C_MAT_OPEN_PBR 2u             /* coat + metal (F82-tint) + dielectric reflection + rough refraction + diffuse */
#define GI_C_MAT_PARAM_COUNT 64u
/* indices into GiCMaterialDesc.p */
#define GI_C_P_BASE_COLOR 0
#define GI_C_P_EMISSION 3
#define GI_C_P_USE_SPECULAR_WORKFLOW 6
#define GI_C_P_SPECULAR_COLOR 7
#define GI_C_P_METALLIC 10
#define GI_C_P_ROUGHNESS 11
#define GI_C_P_CLEARCOAT 12
#define GI_C_P_CLEARCOAT_ROUGHNESS 13
#define GI_C_P_OPACITY 14
#define GI_C_P_OPACITY_THRESHOLD 15
#define GI_C_P_IOR 16
#define GI_C_P_BASE_WEIGHT 17
#define GI_C_P_SPECULAR_WEIGHT 18
#define GI_C_P_COAT_COLOR 19
#define GI_C_P_COAT_IOR 22
#define GI_C_P_TRANSMISSION_WEIGHT 23
#define GI_C_P_TRANSMISSION_COLOR 24
#define GI_C_P_DIFFUSE_ROUGHNESS 27
#define GI_C_P_TRANSMISSION_DEPTH 28
#define GI_C_P_TRANSMISSION_SCATTER 29 /* 3: OpenPBR transmission_scatter (open_pbr_surface.mtlx:35) */
#define GI_C_P_TRANSMISSION_SCATTER_ANISOTROPY 47 /* transmission_scatter_anisotropy (:37); slots 38..46 are reserved (32..37: subsurface radius, coat / specular rotation) */
#define GI_C_P_COAT_DARKENING 48   /* OpenPBR coat_darkening (open_pbr_surface.mtlx:64; default 1): strength of the base darkening under the coat (:470-541) */
#define GI_C_P_FUZZ_WEIGHT 49      /* OpenPBR fuzz_weight / fuzz_color (3) / fuzz_roughness (:57-59): the fuzz (sheen) layer over the coat (:569-581) */
#define GI_C_P_FUZZ_COLOR 50
#define GI_C_P_FUZZ_ROUGHNESS 53
#define GI_C_P_SUBSURFACE_WEIGHT 55  /* OpenPBR subsurface_weight (open_pbr_surface.mtlx:43, 213-218); modelled for thin-walled materials (:140-196), else treated as 0 */
#define GI_C_P_SUBSURFACE_COLOR 56   /* 3: subsurface_color (:45), default 0.8 */
#define GI_C_P_SUBSURFACE_ANISOTROPY 59 /* subsurface_scatter_anisotropy (:51) */
#define GI_C_P_THIN_FILM_WEIGHT 62    /* OpenPBR thin_film_weight (open_pbr_surface.mtlx:71, 426-431, 461-464): mixes a thin film's interference into the Fresnel factor of the dielectric and metal lobes */
#define GI_C_P_THIN_FILM_THICKNESS 63 /* thin_film_thickness in micrometres (:73, 300-304), default 0.5 */
#define GI_C_P_THIN_FILM_IOR 6        /* thin_film_ior (:75), default 1.4 -- OpenPBR class only: the slot is useSpecularWorkflow for UsdPreviewSurface */
#define GI_C_P_SPECULAR_ANISOTROPY 60 /* OpenPBR specular_roughness_anisotropy (open_pbr_surface.mtlx:27, 133-136): GGX stretched along the tangent, alpha_t = r^2 sqrt(2 / (1 + (1 - a)^2)), alpha_b = (1 - a) alpha_t */
#define GI_C_P_COAT_ANISOTROPY 61     /* coat_roughness_anisotropy (:65, 552-555), along the coat's tangent: see GI_C_P_COAT_ROTATION */
#define GI_C_P_COAT_ROTATION 36       /* OpenPBR geometry_coat_tangent (open_pbr_surface.mtlx:91, 561) in the form documents bind it: the geometry tangent turned by this many
                                         TURNS (1 = 360 degrees) towards the bitangent, about the coat's normal (rotate3d of Tworld; Standard Surface's coat_rotation).
                                         0 = the geometry tangent.  Read for an anisotropic coat only (coat weight and coat_roughness_anisotropy > 0).  Like 32..35 an input
                                         of the USER block only (API version 7) */
#define GI_C_P_SPECULAR_ROTATION 37   /* OpenPBR geometry_tangent (open_pbr_surface.mtlx:89; the tangent of the dielectric and conductor lobes, :385 ... 457) in the same form:
                                         the geometry tangent turned by this many turns towards the bitangent (Standard Surface's specular_rotation; glTF's anisotropy_rotation
                                         / 2 pi).  Read for anisotropic base lobes only (specular_roughness_anisotropy > 0); the coat keeps the geometry tangent and its own
                                         turn.  USER block only (API version 7; slots 38..46 stay reserved) */
#define GI_C_P_SUBSURFACE_RADIUS 32       /* OpenPBR subsurface_radius (open_pbr_surface.mtlx:47, default 1): with _RADIUS_SCALE the per-channel mean free path of the volumetric
                                            subsurface_bsdf (:182-192) of materials that are not thin-walled; live in renders with a medium stack (mediumStackSize > 0) */
#define GI_C_P_SUBSURFACE_RADIUS_SCALE 33 /* 3 floats, subsurface_radius_scale (:49, default 1, 0.5, 0.25).  Slots 32..35 are inputs of the USER block only: the device copy of a
                                            material keeps derived constants at these indices (written after the inputs were read).  Taken as given: zeros are zeros (the mean free path is clamped to
                                            1e-6 per channel); the OpenPBR defaults are the front ends' to set (gtl_shim.cpp does; API version 6) */
#define GI_C_P_THIN_WALLED 54      /* OpenPBR geometry_thin_walled (:88) != 0: MDL thin_walled semantics (rp_main.chit:153-157, 188-189, 447) */

/* Note: p[GI_C_P_OPACITY] is the cutout opacity (1 = opaque); a zero-filled block is a fully transparent material. */
typedef struct GiCMaterialDesc {
  uint32_t klass;
  uint32_t flags;
  float p[GI_C_MAT_PARAM_COUNT];
} GiCMaterialDesc;

/* [ext] Textures.  The reference loads images by file path inside gi (TextureManager.cpp:100-275 through imgio's PNG / JPEG /
 * EXR / HDR / TIFF decoders, which are out of scope here) and samples them through the MDL renderer runtime
 * (mdl_interface.glsl:8-38, 127-145).  This boundary takes DECODED pixels: linear float RGBA, row 0 first (the v = 0 side);
 * the pixels are copied.  Lookups are bilinear, LOD 0, after the runtime's wrap handling. */
typedef struct GiCTexture GiCTexture;
typedef struct GiCTextureDesc { uint32_t width, height; const float* rgba; } GiCTextureDesc;
GiCTexture* giCCreateTexture(GiCScene* scene, const GiCTextureDesc* desc);
/* [ext] decodes .png (8/16-bit, non-interlaced), baseline / extended-sequential .jpg (Huffman, 8-bit, grey or YCbCr, any subsampling,
 * restart intervals; progressive files are refused), .hdr or .pfm in-library; srgbToLinear applies the sRGB EOTF to 8-bit colour;
 * a (path, srgbToLinear) pair that is already loaded and alive returns the SAME handle with one more reference (the file cache of
 * TextureManager.cpp:100-150); giCDestroyTexture releases one reference */
GiCTexture* giCCreateTextureFromFile(GiCScene* scene, const char* filePath, int32_t srgbToLinear);
/* [ext] the decoder alone (no device needed; tests): returns 1 and fills width/height (+ rgba if it holds width*height*4 floats).  Goes through the registered
 * asset reader and image loader like giCCreateTextureFromFile / giCCreateDomeLight do. */
int giCDebugDecodeImage(const char* filePath, int32_t srgbToLinear, uint32_t* width, uint32_t* height, float* rgba, uint64_t rgbaFloats);

/* Asset reader -- replaces gtl::GiAssetReader (/root/reference/src/gi/gtl/gi/Gi.h:186-194, giRegisterAssetReader :201).  The reference reads EVERY image through it
 * (TextureManager.cpp:39-52 `_ReadImage`: open -> size -> data -> decode -> close; hdGatling registers an ArResolver-backed one, rendererPlugin.cpp:95-143, 189), so
 * paths a plain fopen cannot serve -- usdz package members, URI resolvers, search-path relative names -- load.  With a reader registered the library opens image
 * `filePath`s (giCCreateTextureFromFile, giCCreateDomeLight, giCDebugDecodeImage) through it and never touches the file system for them; with none (the default, or
 * after giCRegisterAssetReader(NULL)) it reads the file itself.  The struct is copied.  `open` returns an opaque asset or NULL; `data` stays valid until `close`. */
typedef struct GiCAssetReader {
  void* user;
  void* (*open)(void* user, const char* path);
  uint64_t (*size)(void* user, void* asset);
  const void* (*data)(void* user, void* asset);
  void (*close)(void* user, void* asset);
} GiCAssetReader;
void giCRegisterAssetReader(const GiCAssetReader* reader);

/* Image loader hook -- the reference decodes image bytes with imgio (`ImgioLoadImage(data, size, &img, flags)`, TextureManager.cpp:48; PNG, JPEG incl. progressive,
 * EXR, HDR, TIFF), which is outside this library (imgio stays as-is on the reference side).  The library's own decoders cover .png, baseline .jpg,
 * .hdr and .pfm; a registered loader is asked FIRST for every image (bytes from the asset reader or the file), so a host that has imgio hands EXR / TIFF / progressive
 * JPEG over through it.  `load` returns 1 and fills `out` (pixels in imgio's orientation: row 0 = the picture's BOTTOM scanline) or 0 = "not mine", in which case the
 * in-library decoders are tried; `release` is called once for every successful `load` after the pixels have been copied.  keepHdr mirrors ImgioLoadFlags::KeepHdr
 * (set for dome lights, Gi.cpp:2215-2230).  8-bit data is unorm; the sRGB EOTF is applied by the library when the texture was created with srgbToLinear. */
#define GI_C_IMAGE_RGBA8_UNORM 1   /* imgio ImgioFormat::RGBA8_UNORM  */
#define GI_C_IMAGE_RGB16_FLOAT 2   /*       ImgioFormat::RGB16_FLOAT  */
#define GI_C_IMAGE_RGBA16_FLOAT 3  /*       ImgioFormat::RGBA16_FLOAT */
#define GI_C_IMAGE_R32_FLOAT 4     /*       ImgioFormat::R32_FLOAT    */
#define GI_C_IMAGE_RGBA32_FLOAT 5  /* [ext] */
typedef struct GiCDecodedImage { uint32_t format, width, height, reserved; const void* pixels; void* handle; } GiCDecodedImage;
typedef struct GiCImageLoader {
  void* user;
  int32_t (*load)(void* user, const char* path, const void* bytes, uint64_t size, int32_t keepHdr, GiCDecodedImage* out);
  void (*release)(void* user, GiCDecodedImage* image);
} GiCImageLoader;
void giCSetImageLoader(const GiCImageLoader* loader); /* NULL: in-library decoders only (the default) */
void giCDestroyTexture(GiCTexture* texture);

/* texturable inputs of the closed-form materials (UsdUVTexture semantics: value = texel * scale + bias at the hit's st) */
#define GI_C_TEX_BASE_COLOR 0 /* diffuseColor / base_color (rgb) */
#define GI_C_TEX_EMISSION 1   /* emissiveColor / emission (rgb)  */
#define GI_C_TEX_ROUGHNESS 2  /* scalar: `channel` of the texel  */
#define GI_C_TEX_METALLIC 3   /* scalar                          */
#define GI_C_TEX_NORMAL 4     /* tangent-space normal (rgb, usually scale 2 bias -1); bent by mdl_adapt_normal (mdl_interface.glsl:238-256) */
#define GI_C_TEX_OPACITY 5    /* scalar: cutout opacity (UsdPreviewSurface opacity / OpenPBR geometry_opacity), evaluated per candidate hit by the any-hit test (rp_main.ahit:51-60); texture only */
#define GI_C_TEX_COAT_NORMAL 6 /* vector: OpenPBR geometry_coat_normal (open_pbr_surface.mtlx:87, 560), a tangent-space normal map for the coat lobe's own shading frame (OpenPBR class only) */
#define GI_C_TEX_TRANSMISSION_WEIGHT 7 /* scalar: OpenPBR transmission_weight (open_pbr_surface.mtlx:29) */
#define GI_C_TEX_TRANSMISSION_COLOR 8  /* rgb: OpenPBR transmission_color (:31) -- the surface tint when transmission_depth is 0; with a depth the colour defines the MEDIUM's absorption,
                                         * which stays the material's constant (a path's medium is pushed once, where it enters) */
#define GI_C_TEX_SLOT_COUNT 9
#define GI_C_TEX_WRAP_CLAMP 0 /* mdl_types.glsl:117-120 */
#define GI_C_TEX_WRAP_REPEAT 1
#define GI_C_TEX_WRAP_MIRRORED_REPEAT 2
#define GI_C_TEX_WRAP_CLIP 3
typedef struct GiCTextureBinding {
  GiCTexture* texture; /* NULL removes the binding */
  int32_t wrapS, wrapT;
  int32_t channel;
  float scale[4], bias[4];
} GiCTextureBinding;
int giCSetMaterialTexture(GiCMaterial* material, int32_t input, const GiCTextureBinding* binding);
/* [ext] Texture-coordinate transform of an input's lookup: s' = (xf[0] s + xf[1] t) + xf[2], t' = (xf[3] s + xf[4] t) + xf[5]; NULL = identity.  What a UsdTransform2d
 * node between the primvar reader and a UsdUVTexture's `st` means (UsdPreviewSurface specification: result = in * scale, rotated counter-clockwise by `rotation`
 * degrees, + translation: xf = {c sx, -s sy, tx, s sx, c sy, ty}); the reference compiles the node through MaterialX -> MDL (src/mc/impl/MtlxMdlCodeGen.cpp:186-215),
 * the gtl shim's MaterialX reader folds it into these six floats. */
int giCSetMaterialTextureTransform(GiCMaterial* material, int32_t input, const float* xf);

/* Scene data (primvars).  Gi.h:76-92: GiPrimvarData with the byte vector flattened to pointer + size; data = 4-byte elements (float, or int32 for
 * the Int types).  A material input bound to a primvar NAME reads the primvar of the hit mesh (instancer primvars first, mesh primvars
 * override, Gi.cpp:913-929) through scene_data_lookup_float3 / _float (mdl_interface.glsl:281-301, 337-424): barycentric blend of the
 * three vertex values, or the uniform / instance / constant value; integer primvars go through scene_data_lookup_int (:426-476): the
 * value of the NEAREST vertex (largest barycentric weight), converted to float.  The names "CAMERA_POSITION" (colour inputs) and
 * "FRAME" (scalar inputs) are answered from the camera / GiCRenderSettings.frame like the reference's UBO short cuts (:329-334,
 * 390-395).  A mesh without that primvar keeps the input's constant.  A texture on the same input takes precedence. */
#define GI_C_PRIMVAR_FLOAT 0
#define GI_C_PRIMVAR_VEC2 1
#define GI_C_PRIMVAR_VEC3 2
#define GI_C_PRIMVAR_VEC4 3
#define GI_C_PRIMVAR_INT 4
#define GI_C_PRIMVAR_INT2 5
#define GI_C_PRIMVAR_INT3 6
#define GI_C_PRIMVAR_INT4 7
#define GI_C_INTERP_CONSTANT 0
#define GI_C_INTERP_INSTANCE 1
#define GI_C_INTERP_UNIFORM 2
#define GI_C_INTERP_VERTEX 3
typedef struct GiCPrimvarData { const char* name; int32_t type; int32_t interpolation; const void* data; uint64_t dataSize; } GiCPrimvarData;
int giCSetMeshPrimvars(GiCMesh* mesh, uint32_t count, const GiCPrimvarData* primvars);          /* GiMeshDesc.primvars (Gi.h:134) */
int giCSetMeshInstancerPrimvars(GiCMesh* mesh, uint32_t count, const GiCPrimvarData* primvars); /* Gi.h:213 */
int giCSetMaterialPrimvarInput(GiCMaterial* material, int32_t input, const char* primvarName);  /* [ext] NULL or "" removes it; inputs: base colour, emission, roughness, metalness, transmission weight / colour (not the normal, opacity and coat-normal slots: texture only) */

/* [ext] per-frame statistics of the last giCRender on a scene (measurement) */
typedef struct GiCRenderStats {
  double renderMs;       /* wall time of the bounce loop incl. final D2H of the colour AOV */
  double bvhBuildMs;     /* host BVH8 build (0 if not rebuilt)                              */
  double uploadMs;       /* scene upload (0 if not rebuilt)                                 */
  double traceMs;        /* sum of closest-hit traversal kernel time (HIP events)           */
  double shadeMs;        /* sum of shade kernel time                                        */
  double raygenMs;       /* sum of raygen/accumulate kernel time                            */
  double shadowMs;       /* sum of shadow traversal kernel time                             */
  uint64_t samples;      /* pixels * spp rendered by this call                              */
  uint64_t segments;     /* closest-hit rays traced                                         */
  uint64_t shadowRays;   /* shadow rays traced                                              */
  uint64_t nodesVisited; /* BVH8 nodes fetched by closest-hit traversal (if counting on)    */
  uint64_t trisTested;   /* triangles tested by closest-hit traversal (if counting on)      */
  uint64_t shadowNodesVisited;
  uint64_t shadowTrisTested;
  uint32_t iterations;   /* wavefront iterations                                            */
  uint32_t traceLaunches;
  uint32_t nodeCount;    /* BVH8 nodes                                                      */
  uint32_t triangleCount;
  uint32_t fusedPath;    /* 1: the colour pass ran as the fused persistent kernel k_path (LDS-resident scene) */
  uint32_t batches;      /* sample batches the frame was cut into (per-sample colour buffer budget; memory plan of giCRender) */
  uint32_t poolSlots;    /* slots of the persistent path pool this render used (0: fused kernel)                              */
  uint32_t inactiveTriangleCount; /* instanced triangles left out of the BVH: a non-finite or out-of-range (> 1e18) vertex, a non-invertible transform (was reserved0) */
} GiCRenderStats;

/* Gi.h:199-200.  deviceOrdinal selects the HIP device (the reference picks one Vulkan device by score,
 * CgpuVk.cpp:892-909).  One giCInitialize per process, like the reference's global state (Gi.cpp:244-259). */
int giCInitialize(int deviceOrdinal);
void giCTerminate(void);
/* [ext] Multi-device form (north_star: "pixels/samples shard across the 8 GPUs of one node, scene replicated, tiles gathered"; the reference picks ONE
 * device, CgpuVk.cpp:892-909).  The first ordinal is the primary device: render buffers, textures and the single-device entry points live there.  Every
 * giCRender of a whole frame then deals the image rows round-robin to the devices (row r to device r mod N, scene replicated at build time, one host
 * thread per device), and the shares are copied into the primary device's render buffer over xGMI (strided peer copies straight into place) before the
 * one D2H -- the image is bit-identical to a one-device render.  giCInitialize does the same when $GATLING_DEVICES ("0,1,2,3" or "all") is set, which is
 * how an unmodified caller (hdGatling) gets every GPU of the node.  Renders that shard rows themselves (rowStride > 1 / a row range) stay on the primary. */
int giCInitializeDevices(const int32_t* deviceOrdinals, uint32_t count);
/* [ext] Version of this header's ABI: bumped whenever a struct grows or an entry point changes meaning (5: GiCRenderStats gained batches / poolSlots, GI_C_TEX_SLOT_COUNT 9,
 * the subsurface radius slots of GiCMaterialDesc, the asset-reader / image-loader hooks; 6: GiCRenderStats.reserved0 became inactiveTriangleCount,
 * giCDebugShadeClass, an all-zero subsurface radius is no longer read as "unset", the hostile-input rules above giCRender;
 * 7: GI_C_P_COAT_ROTATION / GI_C_P_SPECULAR_ROTATION, slots that were reserved).  A caller compares giCGetApiVersion() with the GI_C_API_VERSION it was built with. */
#define GI_C_API_VERSION 7u
uint32_t giCGetApiVersion(void);
uint32_t giCGetDeviceCount(void);
/* [ext] can the primary device and device `index` of the list address each other's memory?  1 = yes (peer access enabled both ways, or the same physical device):
 * row shares are gathered with strided peer copies over xGMI; 0 = no, -1 = the query / the enabling failed: that device's shares are staged through pinned host
 * memory by the library (correct, slower).  tools/run_scale.sh prints it next to every multi-device measurement. */
int32_t giCGetDevicePeerAccess(uint32_t index);
/* [ext] last error message of the calling thread's most recent failing call ("" if none) */
const char* giCGetLastError(void);

/* Gi.h:204-207 */
GiCMaterial* giCCreateMaterial(GiCScene* scene, const char* name, const GiCMaterialDesc* desc);
void giCDestroyMaterial(GiCMaterial* mat);

/* Gi.h:209-216 */
GiCMesh* giCCreateMesh(GiCScene* scene, const GiCMeshDesc* desc);
void giCSetMeshTransform(GiCMesh* mesh, const float* mat4x4);
void giCSetMeshInstanceTransforms(GiCMesh* mesh, uint32_t count, const float* transforms /* count x 16 */);
void giCSetMeshInstanceIds(GiCMesh* mesh, uint32_t count, const int32_t* ids);
void giCSetMeshMaterial(GiCMesh* mesh, GiCMaterial* mat);
void giCSetMeshVisibility(GiCMesh* mesh, int32_t visible);
void giCDestroyMesh(GiCMesh* mesh);

/* Gi.h:218 */
/* Hostile input (round 6; tests/test_hostile_inputs.py).  The reference copies what Hydra hands it (giCreateMesh, Gi.cpp:628) and leaves the rest to the Vulkan driver,
 * which ignores inactive primitives (CgpuVk.cpp:2561-2670); this library builds its own BVH, so it states its rules:
 *   - a triangle with a vertex that is not finite or lies beyond 1e18 in magnitude AFTER the instance transform is inactive: left out of the BVH, never hit, its
 *     scene-order id kept (FaceId / ObjectId / InstanceId AOVs and giCTraceRays answers do not shift); one warning per mesh on stderr, the count in
 *     GiCRenderStats.inactiveTriangleCount;
 *   - every triangle of an instance whose composed transform has a non-finite entry or no finite inverse (NaN, singular, zero matrix) is inactive;
 *   - giCCreateMaterial refuses a parameter block with a non-finite entry (NULL, like a material the reference fails to compile: the caller falls back to its default
 *     material); a light with a non-finite field is left out of the device arrays and the light counts until a setter repairs it (one warning on stderr);
 *   - a normal / tangent with a non-finite component is taken as +Z, a non-finite texture coordinate as 0, a non-finite bitangent sign as +1;
 *   - giCRender refuses (GI_C_ERROR, giCGetLastError names the field): a camera or dome light with a non-finite field, a forward / up vector that cannot be normalised, a vertical
 *     field of view outside (0, pi); non-finite float render settings (maxSampleValue may be +inf: no clamp); spp 0; mediumStackSize > 15; images beyond 65 535
 *     pixels a side (imageDims is packed 16 + 16 bits, rp_main.h:25-56).  A 0 x N image is a no-op (GI_C_OK).  spp x pixels may exceed 2^32: the frame is cut
 *     into batches of fewer than 2^32 work items.
 * Never a crash, a hang or a NaN pixel from geometry; every accepted case renders the image the oracle renders from the same scene with the inactive triangles
 * replaced by zero-area ones. */
int giCRender(const GiCRenderParams* params);

/* Gi.h:220-221 */
GiCScene* giCCreateScene(void);
void giCDestroyScene(GiCScene* scene);

/* Gi.h:223-228 */
GiCSphereLight* giCCreateSphereLight(GiCScene* scene);
void giCDestroySphereLight(GiCScene* scene, GiCSphereLight* light);
void giCSetSphereLightPosition(GiCSphereLight* light, const float* position);
void giCSetSphereLightBaseEmission(GiCSphereLight* light, const float* rgb);
void giCSetSphereLightRadius(GiCSphereLight* light, float radiusX, float radiusY, float radiusZ);
void giCSetSphereLightDiffuseSpecular(GiCSphereLight* light, float diffuse, float specular);

/* Gi.h:230-235 */
GiCDistantLight* giCCreateDistantLight(GiCScene* scene);
void giCDestroyDistantLight(GiCScene* scene, GiCDistantLight* light);
void giCSetDistantLightDirection(GiCDistantLight* light, const float* direction);
void giCSetDistantLightBaseEmission(GiCDistantLight* light, const float* rgb);
void giCSetDistantLightAngle(GiCDistantLight* light, float angle);
void giCSetDistantLightDiffuseSpecular(GiCDistantLight* light, float diffuse, float specular);

/* Gi.h:237-243 */
GiCRectLight* giCCreateRectLight(GiCScene* scene);
void giCDestroyRectLight(GiCScene* scene, GiCRectLight* light);
void giCSetRectLightOrigin(GiCRectLight* light, const float* origin);
void giCSetRectLightTangents(GiCRectLight* light, const float* t0, const float* t1);
void giCSetRectLightBaseEmission(GiCRectLight* light, const float* rgb);
void giCSetRectLightDimensions(GiCRectLight* light, float width, float height);
void giCSetRectLightDiffuseSpecular(GiCRectLight* light, float diffuse, float specular);

/* Gi.h:245-251 */
GiCDiskLight* giCCreateDiskLight(GiCScene* scene);
void giCDestroyDiskLight(GiCScene* scene, GiCDiskLight* light);
void giCSetDiskLightOrigin(GiCDiskLight* light, const float* origin);
void giCSetDiskLightTangents(GiCDiskLight* light, const float* t0, const float* t1);
void giCSetDiskLightBaseEmission(GiCDiskLight* light, const float* rgb);
void giCSetDiskLightRadius(GiCDiskLight* light, float radiusX, float radiusY);
void giCSetDiskLightDiffuseSpecular(GiCDiskLight* light, float diffuse, float specular);

/* Gi.h:253-257.  The dome light is an equirectangular image looked up by miss rays (rp_main.miss:46-86).  filePath is
 * decoded when it is a Radiance .hdr (RGBE), a .pfm, a .png or a baseline .jpg; other formats need imgio (out of scope) -- hand the decoded pixels
 * over with giCSetDomeLightTexture instead.  A dome light without an image is ignored, exactly like a dome light whose file
 * fails to load in the reference (Gi.cpp:2221-2230): miss rays then see the fallback dome (the colour clear value). */
GiCDomeLight* giCCreateDomeLight(GiCScene* scene, const char* filePath);
void giCSetDomeLightTexture(GiCDomeLight* light, GiCTexture* texture); /* [ext] */
void giCDestroyDomeLight(GiCDomeLight* light);
void giCSetDomeLightRotation(GiCDomeLight* light, const float* quat);
void giCSetDomeLightBaseEmission(GiCDomeLight* light, const float* rgb);
void giCSetDomeLightDiffuseSpecular(GiCDomeLight* light, float diffuse, float specular);

/* Gi.h:259-261 */
GiCRenderBuffer* giCCreateRenderBuffer(uint32_t width, uint32_t height, int32_t format /* GiCRenderBufferFormat */);
void giCDestroyRenderBuffer(GiCRenderBuffer* renderBuffer);
void* giCGetRenderBufferMem(GiCRenderBuffer* renderBuffer);

/* [ext] device-resident copy of the render buffer (valid after the first giCRender that bound it), so a
 * multi-GPU caller can gather tiles with RCCL without a host round trip. */
void* giCGetRenderBufferDeviceMem(GiCRenderBuffer* renderBuffer);
/* [ext] when nonzero, giCRender leaves results on the device only (no D2H copy); default 0 = reference behaviour */
void giCSetRenderBufferDeviceOnly(GiCRenderBuffer* renderBuffer, int32_t deviceOnly);
/* [ext] statistics of the last giCRender on this scene; countTraversal != 0 in giCSetSceneOption enables
 * node/triangle counters (slower; measurement runs only). */
int giCGetRenderStats(const GiCScene* scene, GiCRenderStats* out);
#define GI_C_SCENE_OPTION_COUNT_TRAVERSAL 1
/* value N > 0: record HIP events around the stage launches of every N-th wavefront iteration (1 = every launch; events
 * on every launch cost ~16 % of a frame) and scale the per-stage totals accordingly; 0 = off */
#define GI_C_SCENE_OPTION_KERNEL_TIMERS 2
/* size of the persistent path pool (slots; default 4 Mi) and of the per-sample colour buffer (MiB; default 49152, at most a sixth of the device's memory) -- results
 * do not depend on either (tests/test_gpu_parity.py::test_pool_and_batch_invariance); 0 restores the default */
#define GI_C_SCENE_OPTION_POOL_SLOTS 3
#define GI_C_SCENE_OPTION_SAMPLE_BUFFER_MB 4
/* Traversal kernel for scenes whose BVH does not fit LDS: 0 = block-synchronous k_trace; N in 1..64 = persistent waves that
 * claim new rays once N lanes are idle (k_trace_dyn, default 8); -1 = default.  Results are identical either way. */
#define GI_C_SCENE_OPTION_TRACE_DYNAMIC 5
#define GI_C_SCENE_OPTION_TWO_LEVEL 6     /* [ext] 1: two-level BVH (TLAS over instances + one object-space BLAS per mesh) for scenes beyond LDS; default 0: one
                                             flat BVH over the instanced triangles (faster today); the image does not depend on it */
/* [ext] Scenes whose whole BVH fits LDS (<= 384 nodes, <= 128 triangles; no medium stack, no dome image) are rendered by a fused persistent kernel:
 * -1 (default) / 2 = k_path, one path per lane in registers; 1 = k_path_bw, the wave-local wavefront, when next-event estimation is off (k_path otherwise);
 * 0 = run the wavefront stage kernels on them too.  The image does not depend on it. */
#define GI_C_SCENE_OPTION_FUSED_PATH 7
/* [ext] upper bound on the devices a giCRender of this scene uses (0 = all the library was initialised on; 1 = primary only). */
#define GI_C_SCENE_OPTION_DEVICES 8
int giCSetSceneOption(GiCScene* scene, int32_t option, int32_t value);
/* [ext] closest hit of one ray through the device traversal kernel (parity tests of the BVH8 path).
 * Returns 1 on hit (t,u,v, instance, prim written), 0 on miss, <0 on error.  Candidates on cut-out materials pass the any-hit test of the render
 * (rp_main.ahit:51-60) with the random state 0: the answer is a function of the ray and the scene alone. */
int giCTraceRays(GiCScene* scene, uint32_t count, const float* origins /*3*count*/, const float* dirs /*3*count*/,
                 float tMin, float tMax, float* outTUV /*3*count*/, int32_t* outInstPrim /*2*count, -1 on miss*/);

/* [ext] host-only self check of the BVH8 builder (no device needed): builds the tree over `triCount` triangles
 * (9 floats each: v0, v1, v2) and verifies that every triangle is reachable and lies inside the dequantised box of
 * every ancestor slot.  Returns the number of violations (0 = conservative), <0 on error. */
/* [ext] device-side known-answer hook: closed-form BSDF sample + evaluate for `count` explicit shading frames.
 * in: 22 floats per item (normal, tangentU, tangentV, geomNormal, k1, k2, xi[4]); out: 15 floats per item
 * (k2, bsdf_over_pdf, pdf, event, eval diffuse, eval glossy, eval pdf). */
int giCDebugEvalBsdf(const GiCMaterialDesc* desc, uint32_t count, const float* in, float* out);
/* [ext] host-only: the shade class an untextured material's hits are binned and shaded by (k_shade variants; the reference compiles one hit shader per material with
 * feature #defines, GlslShaderGen.cpp:204-274): 0 diffuse, 1 UsdPreviewSurface, 2 OpenPBR with every lobe, 3 OpenPBR BASE (no coat / fuzz / thin film / anisotropy /
 * transmission / subsurface, not thin-walled, all parameters finite).  giCDebugEvalBsdf runs the variant this names.  <0 on error. */
int giCDebugShadeClass(const GiCMaterialDesc* desc);
/* [ext] device self check: the kernels' square root (the compiler's correctly rounded expansion without the steps that ordinary arguments do not need) against sqrtf
 * for the `count` float bit patterns from `first` on; count = 2^32 covers every float.  Returns the number of arguments whose results differ (0), -1 on error. */
int64_t giCDebugCheckSqrt(uint32_t first, uint64_t count);
/* [ext] device-side hook for the MDL renderer runtime's remaining texture entry points (mdl_interface.glsl:45-65, 86-105, 167-221), which only MDL-generated code
 * calls: `rgba` = width x height x depth RGBA float texels (slice by slice; depth 1 = a 2-D image); per query 8 floats (kind, valid, c0, c1, c2, wrapU, wrapV, wrapW)
 * with kind 0 tex_texel_float4_2d(c0, c1), 1 tex_resolution_2d, 2 tex_lookup_float4_3d(c0, c1, c2; wraps), 3 tex_texel_float4_3d(c0, c1, c2); valid 0 = the invalid
 * texture; per result 4 floats. */
int giCDebugTexRuntime(const float* rgba, uint32_t width, uint32_t height, uint32_t depth, uint32_t count, const float* queries, float* out);
int giCDebugValidateBvh(const float* triVerts, uint32_t triCount, uint32_t* outNodeCount, uint32_t* outMaxDepth);
/* [ext] the same host-only check for the partitioned layout incremental transform updates use: `partCount` consecutive triangle ranges, one
 * subtree each, joined by a top tree over the subtree roots.  Returns the violations of the assembled tree (0 = every triangle reachable and inside every
 * ancestor slot's box), <0 on error. */
int giCDebugValidatePartitionedBvh(const float* triVerts, uint32_t triCount, uint32_t partCount, uint32_t* outNodeCount, uint32_t* outMaxDepth);

#ifdef __cplusplus
}
#endif
#endif
