// gi_textures.cpp -- textures, the asset reader / image loader hooks, texture bindings of materials (TextureManager.cpp:39-275)
// (one of the translation units gi_c.cpp was split into in round 6; shared declarations: gi_host.h)
#include "gi_host.h"

extern "C" {
// ---------------------------------------------------------------------------------------------------------------
// textures [ext]: decoded pixels in, device copies made with the next scene build (TextureManager.cpp:100-275 minus imgio)
// ---------------------------------------------------------------------------------------------------------------
static GiCTexture* createTextureImpl(GiCScene* scene, const GiCTextureDesc* desc);
GiCTexture* giCCreateTexture(GiCScene* scene, const GiCTextureDesc* desc)
{
  try { return createTextureImpl(scene, desc); }
  catch (const std::exception& e) { setError(std::string("giCCreateTexture: ") + e.what()); return nullptr; }
}
static GiCTexture* createTextureImpl(GiCScene* scene, const GiCTextureDesc* desc)
{
  if (!scene || !desc || !desc->rgba || desc->width == 0 || desc->height == 0) { setError("giCCreateTexture: bad arguments"); return nullptr; }
  std::unique_ptr<GiCTexture> t(new GiCTexture{scene, desc->width, desc->height,
      std::vector<float>(desc->rgba, desc->rgba + (size_t)desc->width * desc->height * 4)});
  std::lock_guard<std::mutex> g(scene->mutex);
  scene->textures.push_back(t.get());
  scene->dirty |= DIRTY_MATERIALS | DIRTY_FRAMEBUFFER;
  return t.release();
}

// ---- image files: asset reader + loader hook in front of the in-library decoders
// --------------------------------------------------------------------------------
// (TextureManager.cpp:39-52: every image goes open -> size -> data -> ImgioLoadImage -> close through the registered GiAssetReader)
static std::mutex g_imageHookMutex;
static GiCAssetReader g_assetReader{};   // .open == nullptr: none registered
static GiCImageLoader g_imageLoader{};   // .load == nullptr: none registered
void giCRegisterAssetReader(const GiCAssetReader* r)
{
  std::lock_guard<std::mutex> g(g_imageHookMutex);
  if (r && r->open && r->size && r->data && r->close) g_assetReader = *r; else g_assetReader = GiCAssetReader{};
}
void giCSetImageLoader(const GiCImageLoader* l)
{
  std::lock_guard<std::mutex> g(g_imageHookMutex);
  if (l && l->load) g_imageLoader = *l; else g_imageLoader = GiCImageLoader{};
}
static float halfBitsToFloat(uint16_t h) // IEEE binary16 -> binary32, exact
{
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
  uint32_t bits;
  if (exp == 0u) {
    if (man == 0u) bits = sign;
    else { int e = -1; uint32_t m = man; do { e++; m <<= 1; } while (!(m & 0x400u)); bits = sign | (uint32_t)(127 - 15 - e) << 23 | (m & 0x3ffu) << 13; }
  } else if (exp == 31u) bits = sign | 0x7f800000u | man << 13;
  else bits = sign | (exp + 112u) << 23 | man << 13;
  float f; memcpy(&f, &bits, 4); return f;
}
// the image behind `path` as linear float RGBA in imgio's orientation
extern "C++" bool loadImage(const char* path, bool srgbToLinear, bool keepHdr, uint32_t& w, uint32_t& h, std::vector<float>& px)
{
  GiCAssetReader reader; GiCImageLoader loader;
  { std::lock_guard<std::mutex> g(g_imageHookMutex); reader = g_assetReader; loader = g_imageLoader; }
  std::vector<uint8_t> fileBytes; const uint8_t* bytes = nullptr; size_t size = 0; void* asset = nullptr;
  if (reader.open) {
    asset = reader.open(reader.user, path);
    if (!asset) return false;
    size = (size_t)reader.size(reader.user, asset);
    bytes = static_cast<const uint8_t*>(reader.data(reader.user, asset));
  } else {
    if (!readFileBytes(path, fileBytes)) return false;
    bytes = fileBytes.data(); size = fileBytes.size();
  }
  bool ok = false;
  GiCDecodedImage img{};
  if (bytes && loader.load && loader.load(loader.user, path, bytes, (uint64_t)size, keepHdr ? 1 : 0, &img) == 1) {
    const size_t n = (size_t)img.width * img.height;
    // (a hook's answer is untrusted: a bogus width x height must not become a bad_alloc that leaves through the extern "C" callers, and release / close run
    // whatever happens -- ADVICE r05. 2^28 texels = 4 GiB of fp32 RGBA is the cap; the reference's largest texture is bounded by maxImageDimension2D, 16 384^2
    // = 2^28)
    if (img.pixels && n > 0 && n <= ((size_t)1 << 28) && img.format >= GI_C_IMAGE_RGBA8_UNORM && img.format <= GI_C_IMAGE_RGBA32_FLOAT) try {
      w = img.width; h = img.height; px.assign(n * 4, 1.0f);
      for (size_t i = 0; i < n; i++) {
        float* o = &px[i * 4];
        switch (img.format) {
          case GI_C_IMAGE_RGBA8_UNORM: { const uint8_t* p = static_cast<const uint8_t*>(img.pixels) + i * 4;
              for (int c = 0; c < 3; c++) o[c] = srgbToLinear ? srgb8ToLinear(p[c]) : (float)p[c] / 255.0f; o[3] = (float)p[3] / 255.0f; break; }
          case GI_C_IMAGE_RGB16_FLOAT: { const uint16_t* p = static_cast<const uint16_t*>(img.pixels) + i * 3;
              for (int c = 0; c < 3; c++) o[c] = halfBitsToFloat(p[c]); break; }
          case GI_C_IMAGE_RGBA16_FLOAT: { const uint16_t* p = static_cast<const uint16_t*>(img.pixels) + i * 4;
              for (int c = 0; c < 4; c++) o[c] = halfBitsToFloat(p[c]); break; }
          case GI_C_IMAGE_R32_FLOAT: { const float v = static_cast<const float*>(img.pixels)[i]; o[0] = o[1] = o[2] = v; break; }
          default: memcpy(o, static_cast<const float*>(img.pixels) + i * 4, 16); break;
        }
      }
      ok = true;
    } catch (const std::exception&) { ok = false; px.clear(); }
    if (loader.release) loader.release(loader.user, &img);
  }
  if (!ok && bytes) ok = decodeImageBytes(bytes, size, srgbToLinear, w, h, px);
  if (asset) reader.close(reader.user, asset);
  return ok;
}

// File textures are shared: a path that is already loaded (and still alive) yields the same texture with one more reference,
// as GiTextureManager's weak-pointer cache does (TextureManager.cpp:100-150); giCDestroyTexture drops one reference.
GiCTexture* giCCreateTextureFromFile(GiCScene* scene, const char* filePath, int32_t srgbToLinear)
{
  if (!scene || !filePath) { setError("giCCreateTextureFromFile: bad arguments"); return nullptr; }
  const std::string key = std::string(srgbToLinear ? "s:" : "l:") + filePath;
  {
    std::lock_guard<std::mutex> g(scene->mutex);
    for (GiCTexture* t : scene->textures) if (t->cacheKey == key) { t->refs++; return t; }
  }
  uint32_t w = 0, h = 0; std::vector<float> px;
  if (!loadImage(filePath, srgbToLinear != 0, /*keepHdr=*/false, w, h, px)) {
    setError("giCCreateTextureFromFile: cannot open or decode the image (in-library: .png, baseline .jpg, .hdr, .pfm; other formats through "
             "giCSetImageLoader)");
    return nullptr;
  }
  GiCTextureDesc td{w, h, px.data()};
  GiCTexture* t = giCCreateTexture(scene, &td);
  if (t) { std::lock_guard<std::mutex> g(scene->mutex); t->cacheKey = key; }
  return t;
}

int giCDebugDecodeImage(const char* filePath, int32_t srgbToLinear, uint32_t* width, uint32_t* height, float* rgba, uint64_t rgbaFloats)
{
  uint32_t w = 0, h = 0; std::vector<float> px;
  if (!filePath || !loadImage(filePath, srgbToLinear != 0, /*keepHdr=*/false, w, h, px)) return 0;
  if (width) *width = w;
  if (height) *height = h;
  if (rgba && rgbaFloats >= px.size()) memcpy(rgba, px.data(), px.size() * sizeof(float));
  return 1;
}

void giCDestroyTexture(GiCTexture* tex)
{
  if (!tex) return;
  GiCScene* s = tex->scene;
  {
    std::lock_guard<std::mutex> g(s->mutex);
    if (--tex->refs != 0u) return; // shared file texture still in use
    s->textures.erase(std::remove(s->textures.begin(), s->textures.end(), tex), s->textures.end());
    for (GiCMaterial* m : s->materials) for (auto& b : m->tex) if (b.texture == tex) b.texture = nullptr;
    s->dirty |= DIRTY_MATERIALS | DIRTY_BVH | DIRTY_FRAMEBUFFER;
  }
  delete tex;
}

int giCSetMaterialTexture(GiCMaterial* mat, int32_t input, const GiCTextureBinding* binding)
{
  if (!mat || input < 0 || input >= GI_C_TEX_SLOT_COUNT) { setError("giCSetMaterialTexture: bad arguments"); return GI_C_ERROR; }
  if (binding && binding->texture && binding->texture->scene != mat->scene) { setError("giCSetMaterialTexture: texture belongs to another scene");
      return GI_C_ERROR; }
  if (binding && (binding->wrapS < 0 || binding->wrapS > 3 || binding->wrapT < 0 || binding->wrapT > 3)) { setError("giCSetMaterialTexture: bad wrap mode");
      return GI_C_ERROR; }
  std::lock_guard<std::mutex> g(mat->scene->mutex);
  if (binding) mat->tex[input] = *binding; else mat->tex[input] = GiCTextureBinding{};
  mat->scene->dirty |= DIRTY_MATERIALS | DIRTY_BVH | DIRTY_FRAMEBUFFER;
  return GI_C_OK;
}

int giCSetMaterialTextureTransform(GiCMaterial* mat, int32_t input, const float* xf)
{
  if (!mat || input < 0 || input >= GI_C_TEX_SLOT_COUNT) { setError("giCSetMaterialTextureTransform: bad arguments"); return GI_C_ERROR; }
  std::lock_guard<std::mutex> g(mat->scene->mutex);
  mat->hasTexXf[input] = xf != nullptr;
  if (xf) memcpy(mat->texXf[input], xf, sizeof(float) * 6);
  mat->scene->dirty |= DIRTY_MATERIALS | DIRTY_BVH | DIRTY_FRAMEBUFFER;
  return GI_C_OK;
}

} // extern "C"
