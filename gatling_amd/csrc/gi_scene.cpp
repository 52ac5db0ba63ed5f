// gi_scene.cpp -- scene, material and mesh containers with their dirty flags, primvars (Gi.cpp:620-782, Gi.h:76-92)
// (one of the translation units gi_c.cpp was split into in round 6; shared declarations: gi_host.h)
#include "gi_host.h"

extern "C" {
// ---------------------------------------------------------------------------------------------------------------
// scene / material / mesh
// ---------------------------------------------------------------------------------------------------------------
GiCScene* giCCreateScene(void)
{
  if (!g_ctx.initialized) { setError("giCCreateScene before giCInitialize"); return nullptr; }
  return new GiCScene();
}

void giCDestroyScene(GiCScene* s)
{
  if (!s) return;
  std::lock_guard<std::mutex> g(g_ctx.resourceMutex);
  for (auto& r : s->replicas) {
    (void)hipSetDevice(g_ctx.devs[r->slot].device);
    (void)hipStreamSynchronize(g_ctx.devs[r->slot].stream);
    r->releaseAll();
  }
  (void)hipSetDevice(g_ctx.device);
  (void)hipStreamSynchronize(g_ctx.stream);
  s->releaseAll();
  delete s;
}

GiCMaterial* giCCreateMaterial(GiCScene* scene, const char* name, const GiCMaterialDesc* desc)
{
  if (!scene || !desc) { setError("giCCreateMaterial: null argument"); return nullptr; }
  if (desc->klass > GI_C_MAT_OPEN_PBR) { setError("giCCreateMaterial: unsupported material class"); return nullptr; }
  // Hostile input (include/gi_c.h): a non-finite parameter would be NaN radiance on every path that meets the material.  Refused like a material the reference
  // fails to compile (giCreateMaterialFrom* return nullptr there, and hdGatling falls back to its default material, material.cpp)
  for (uint32_t i = 0; i < MAT_PARAM_COUNT; i++)
    if (!std::isfinite(desc->p[i])) { setError("giCCreateMaterial: parameter " + std::to_string(i) + " of material '" + (name ? name : "") + "' is not finite");
        return nullptr; }
  GiCMaterial* m = new GiCMaterial{scene, name ? name : "", *desc};
  // (subsurface_radius / subsurface_radius_scale, slots 32..35, are taken as given -- zeros included: the per-channel mean free path is clamped to 1e-6, as the
  // oracle does.  Round 5 read an all-zero radius AND scale as "unset"; that guess belongs
  // to the front ends, which set OpenPBR's defaults themselves: gtl_shim.cpp -- ADVICE r05)
  std::lock_guard<std::mutex> g(scene->mutex);
  scene->materials.push_back(m);
  scene->dirty |= DIRTY_MATERIALS | DIRTY_BVH | DIRTY_FRAMEBUFFER;
  return m;
}

void giCDestroyMaterial(GiCMaterial* mat)
{
  if (!mat) return;
  GiCScene* s = mat->scene;
  {
    std::lock_guard<std::mutex> g(s->mutex);
    s->materials.erase(std::remove(s->materials.begin(), s->materials.end(), mat), s->materials.end());
    for (GiCMesh* m : s->meshes) if (m->material == mat) m->material = nullptr;
    s->dirty |= DIRTY_MATERIALS | DIRTY_BVH | DIRTY_FRAMEBUFFER;
  }
  delete mat;
}

static GiCMesh* createMeshImpl(GiCScene* scene, const GiCMeshDesc* d);
GiCMesh* giCCreateMesh(GiCScene* scene, const GiCMeshDesc* d)
{
  try { return createMeshImpl(scene, d); }
  catch (const std::exception& e) { setError(std::string("giCCreateMesh: ") + e.what()); return nullptr; }
}
static GiCMesh* createMeshImpl(GiCScene* scene, const GiCMeshDesc* d)
{
  if (!scene || !d) { setError("giCCreateMesh: null argument"); return nullptr; }
  if ((d->faceCount && !d->faces) || (d->vertexCount && !d->vertices)) { setError("giCCreateMesh: null arrays"); return nullptr; }
  for (uint32_t i = 0; i < d->faceCount; i++)
    for (int k = 0; k < 3; k++)
      if (d->faces[i].v_i[k] >= d->vertexCount) { setError("giCCreateMesh: face index out of range"); return nullptr; }
  std::unique_ptr<GiCMesh> m(new GiCMesh());
  m->scene = scene; m->name = d->name ? d->name : "";
  m->vertices.assign(d->vertices, d->vertices + d->vertexCount); // copies, like giProcessMeshData (Gi.cpp:628)
  m->faces.assign(d->faces, d->faces + d->faceCount);
  if (d->faceIds) m->faceIds.assign(d->faceIds, d->faceIds + d->faceCount);
  m->id = d->id; m->doubleSided = d->isDoubleSided != 0; m->flipFacing = d->isLeftHanded != 0; m->maxFaceId = d->maxFaceId;
  std::lock_guard<std::mutex> g(scene->mutex);
  scene->meshes.push_back(m.get());
  scene->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER;
  return m.release();
}

void giCSetMeshTransform(GiCMesh* mesh, const float* mat4x4)
{
  if (!mesh || !mat4x4) return;
  std::lock_guard<std::mutex> g(mesh->scene->mutex); // buildScene reads the mesh under this lock (giCRender on another thread)
  memcpy(mesh->transform, mat4x4, sizeof(float) * 16);
  // same triangles elsewhere: incremental update (every instance of the mesh moves)
  if (mesh->builtInstances != 0xffffffffu) { mesh->xformDirty = true; mesh->instDirty.clear(); mesh->scene->dirty |= DIRTY_XFORM | DIRTY_FRAMEBUFFER; }
  else mesh->scene->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER;
}

void giCSetMeshInstanceTransforms(GiCMesh* mesh, uint32_t count, const float* transforms)
{
  try { // (a copy of the caller's array: an allocation failure must not cross the C ABI)
  if (!mesh || (count && !transforms)) return;
  std::vector<float> copy(transforms, transforms + (size_t)count * 16); // copy outside the lock, swap inside
  std::lock_guard<std::mutex> g(mesh->scene->mutex);
  mesh->instanceTransforms.swap(copy); // (`copy` now holds the previous transforms)
  if (mesh->builtInstances == count && copy.size() == (size_t)count * 16) { // same instance count: instances moved -- note which
    const bool all = mesh->xformDirty && mesh->instDirty.empty();
    if (!all) {
      if (mesh->instDirty.size() != count) mesh->instDirty.assign(count, 0);
      for (uint32_t i = 0; i < count; i++) if (memcmp(&copy[16 * (size_t)i], &mesh->instanceTransforms[16 * (size_t)i], 64) != 0) mesh->instDirty[i] = 1;
    }
    mesh->xformDirty = true; mesh->scene->dirty |= DIRTY_XFORM | DIRTY_FRAMEBUFFER;
  }
  else mesh->scene->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER;
  } catch (const std::exception& e) { setError(std::string("giCSetMeshInstanceTransforms: ") + e.what()); }
}

void giCSetMeshInstanceIds(GiCMesh* mesh, uint32_t count, const int32_t* ids)
{
  try { // (a copy of the caller's array: an allocation failure must not cross the C ABI)
  if (!mesh || (count && !ids)) return;
  std::vector<int32_t> copy(ids, ids + count);
  std::lock_guard<std::mutex> g(mesh->scene->mutex);
  mesh->instanceIds.swap(copy);
  mesh->scene->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER;
  } catch (const std::exception& e) { setError(std::string("giCSetMeshInstanceIds: ") + e.what()); }
}

void giCSetMeshMaterial(GiCMesh* mesh, GiCMaterial* mat)
{
  if (!mesh) return;
  std::lock_guard<std::mutex> g(mesh->scene->mutex);
  mesh->material = mat;
  mesh->scene->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER;
}

void giCSetMeshVisibility(GiCMesh* mesh, int32_t visible)
{
  if (!mesh) return;
  std::lock_guard<std::mutex> g(mesh->scene->mutex);
  mesh->visible = visible != 0;
  mesh->scene->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER;
}

void giCDestroyMesh(GiCMesh* mesh)
{
  if (!mesh) return;
  GiCScene* s = mesh->scene;
  {
    std::lock_guard<std::mutex> g(s->mutex);
    s->meshes.erase(std::remove(s->meshes.begin(), s->meshes.end(), mesh), s->meshes.end());
    s->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER;
  }
  delete mesh;
}

// ---------------------------------------------------------------------------------------------------------------
// scene data (primvars): Gi.h:76-92, 134, 213
// ---------------------------------------------------------------------------------------------------------------
static int setPrimvarsImpl(GiCMesh* mesh, std::vector<GiCPrimvar>& dst, uint32_t count, const GiCPrimvarData* pv);
static int setPrimvars(GiCMesh* mesh, std::vector<GiCPrimvar>& dst, uint32_t count, const GiCPrimvarData* pv)
{
  try { return setPrimvarsImpl(mesh, dst, count, pv); }
  catch (const std::exception& e) { setError(std::string("giCSetMesh*Primvars: ") + e.what()); return GI_C_ERROR; }
}
static int setPrimvarsImpl(GiCMesh* mesh, std::vector<GiCPrimvar>& dst, uint32_t count, const GiCPrimvarData* pv)
{
  if (!mesh || (count && !pv)) { setError("giCSetMesh*Primvars: bad arguments"); return GI_C_ERROR; }
  std::vector<GiCPrimvar> v;
  for (uint32_t i = 0; i < count; i++) {
    if (!pv[i].name || pv[i].type < 0 || pv[i].type > GI_C_PRIMVAR_INT4 || pv[i].interpolation < 0
        || pv[i].interpolation > GI_C_INTERP_VERTEX) { setError("giCSetMesh*Primvars: bad primvar"); return GI_C_ERROR; }
    GiCPrimvar p{pv[i].name, pv[i].type, pv[i].interpolation, {}};
    // float and int32 elements are both 4 bytes: integer primvars keep their bit patterns in the float array (scene_data_lookup_int reads them back)
    if (pv[i].data) p.data.assign((const float*)pv[i].data, (const float*)pv[i].data + pv[i].dataSize / 4);
    v.push_back(std::move(p));
  }
  std::lock_guard<std::mutex> g(mesh->scene->mutex);
  dst = std::move(v);
  mesh->scene->dirty |= DIRTY_BVH | DIRTY_FRAMEBUFFER; // Gi.cpp:685-700
  return GI_C_OK;
}
int giCSetMeshPrimvars(GiCMesh* mesh, uint32_t count, const GiCPrimvarData* pv) { return mesh
    ? setPrimvars(mesh, mesh->primvars, count, pv) : (setError("giCSetMeshPrimvars: null mesh"), GI_C_ERROR); }
int giCSetMeshInstancerPrimvars(GiCMesh* mesh, uint32_t count, const GiCPrimvarData* pv) { return mesh
    ? setPrimvars(mesh, mesh->instancerPrimvars, count, pv) : (setError("giCSetMeshInstancerPrimvars: null mesh"), GI_C_ERROR); }
int giCSetMaterialPrimvarInput(GiCMaterial* mat, int32_t input, const char* name)
{
  if (!mat || input < 0 || input >= GI_C_TEX_SLOT_COUNT || input == GI_C_TEX_NORMAL || input == GI_C_TEX_OPACITY
      || input == GI_C_TEX_COAT_NORMAL) { setError("giCSetMaterialPrimvarInput: bad arguments"); return GI_C_ERROR; }
  std::lock_guard<std::mutex> g(mat->scene->mutex);
  mat->primvarInput[input] = name ? name : "";
  mat->scene->dirty |= DIRTY_MATERIALS | DIRTY_BVH | DIRTY_FRAMEBUFFER;
  return GI_C_OK;
}

} // extern "C"
