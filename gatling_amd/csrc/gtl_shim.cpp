// gtl_shim.cpp -- the reference's C++ API (include/gtl/gi/Gi.h, mirroring /root/reference/src/gi/gtl/gi/Gi.h:199-261) on top of
// the C ABI (include/gi_c.h).  Every function forwards 1:1; std::vector arguments become pointer + count.  The one
// non-mechanical piece is material creation: a tiny scanner reads UsdPreviewSurface / open_pbr_surface (and, translated onto the same closed forms,
// standard_surface / gltf_pbr) nodes with constant, primvar or image inputs out of a MaterialX document string (what hdGatling's material network compiler
// produces, src/hdGatling/materialNetworkCompiler.cpp:667-720) and fills a closed-form parameter block.
#include "../../include/gtl/gi/Gi.h"
#include "../../include/gi_c.h"

#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <map>
#include <string>

namespace gtl
{
  struct GiScene { GiCScene* h; };
  struct GiMesh { GiCMesh* h; };
  struct GiMaterial { GiCMaterial* h; std::vector<GiCTexture*> textures; /* created from the document's image nodes, owned by the material */ };
  struct GiRenderBuffer { GiCRenderBuffer* h; };
  struct GiSphereLight { GiCSphereLight* h; };
  struct GiDistantLight { GiCDistantLight* h; };
  struct GiRectLight { GiCRectLight* h; };
  struct GiDiskLight { GiCDiskLight* h; };
  struct GiDomeLight { GiCDomeLight* h; };

  namespace
  {
    GiAssetReader* s_assetReader = nullptr;

    // ---- minimal MaterialX reader: first <UsdPreviewSurface|open_pbr_surface ...> element and its <input name value> children
    struct MtlxNode { std::string category; std::map<std::string, std::string> inputs; std::map<std::string, std::string> connections;
        /* input -> upstream node name */
                      std::map<std::string, std::string> outputs; /* input -> which output of that node (connections through a nodegraph) */ };

    std::string attr(const std::string& tag, const char* name)
    {
      std::string key = std::string(name) + "=\"";
      size_t p = tag.find(key);
      if (p == std::string::npos) return {};
      p += key.size();
      size_t e = tag.find('"', p);
      return e == std::string::npos ? std::string() : tag.substr(p, e - p);
    }

    bool findSurfaceNode(const std::string& doc, MtlxNode& out)
    {
      static const char* kCats[] = {"UsdPreviewSurface", "open_pbr_surface", "standard_surface", "gltf_pbr"};
      size_t best = std::string::npos; const char* bestCat = nullptr;
      for (const char* c : kCats) {
        std::string open = std::string("<") + c;
        size_t p = 0;
        while ((p = doc.find(open, p)) != std::string::npos) {
          char nx = doc[p + open.size()];
          if (nx == ' ' || nx == '\t' || nx == '\n' || nx == '\r' || nx == '>') break;
          p += open.size();
        }
        if (p != std::string::npos && p < best) { best = p; bestCat = c; }
      }
      if (!bestCat) return false;
      out.category = bestCat;
      size_t close = doc.find(std::string("</") + bestCat, best);
      size_t headEnd = doc.find('>', best);
      if (headEnd == std::string::npos) return false;
      if (doc[headEnd - 1] == '/' || close == std::string::npos) return true; // no inputs: all defaults
      size_t p = headEnd;
      while ((p = doc.find("<input", p)) != std::string::npos && p < close) {
        size_t e = doc.find('>', p);
        if (e == std::string::npos) break;
        std::string tag = doc.substr(p, e - p + 1);
        std::string name = attr(tag, "name"), value = attr(tag, "value");
        if (!name.empty() && !value.empty() && attr(tag, "nodename").empty() && attr(tag, "nodegraph").empty()) out.inputs[name] = value; // constants
        else if (!name.empty() && !attr(tag, "nodename").empty()) out.connections[name] = attr(tag, "nodename");
        else if (!name.empty() && !attr(tag, "nodegraph").empty()) {
          // HdMtlxCreateMtlxDocumentFromHdNetwork puts upstream nodes into a <nodegraph> and connects the surface input to one of its
          // <output name=... nodename=...>: follow it to the node that feeds the output
          const std::string ng = attr(tag, "nodegraph"), on = attr(tag, "output");
          size_t g = doc.find("<nodegraph name=\"" + ng + "\"");
          size_t gEnd = g == std::string::npos ? g : doc.find("</nodegraph>", g);
          size_t q = g;
          while (g != std::string::npos && (q = doc.find("<output", q)) != std::string::npos && q < gEnd) {
            size_t oe = doc.find('>', q);
            if (oe == std::string::npos) break;
            const std::string otag = doc.substr(q, oe - q + 1);
            if (on.empty()
                || attr(otag, "name") == on) { if (!attr(otag, "nodename").empty()) { out.connections[name] = attr(otag, "nodename");
                out.outputs[name] = attr(otag, "output"); } break; }
            q = oe;
          }
        }
        p = e;
      }
      return true;
    }

    // If `nodeName` is a primvar reader (MaterialX <geompropvalue geomprop=...>, or a UsdPrimvarReader_* with varname), its primvar name
    std::string primvarOfNode(const std::string& doc, const std::string& nodeName)
    {
      const std::string key = "name=\"" + nodeName + "\"";
      size_t p = 0;
      while ((p = doc.find(key, p)) != std::string::npos) {
        size_t lt = doc.rfind('<', p);
        if (lt == std::string::npos) break;
        size_t sp = doc.find_first_of(" \t\r\n>", lt);
        const std::string cat = doc.substr(lt + 1, sp - lt - 1);
        if (cat == "geompropvalue" || cat.rfind("UsdPrimvarReader", 0) == 0) {
          size_t close = doc.find("</" + cat, p), q = p;
          while ((q = doc.find("<input", q)) != std::string::npos && (close == std::string::npos || q < close)) {
            size_t e = doc.find('>', q);
            if (e == std::string::npos) break;
            const std::string tag = doc.substr(q, e - q + 1), n = attr(tag, "name");
            if (n == "geomprop" || n == "varname") return attr(tag, "value");
            q = e;
          }
          return {};
        }
        p += key.size();
      }
      return {};
    }

    // The upstream node `nodeName`: its category and its constant inputs (name -> value)
    bool readNode(const std::string& doc, const std::string& nodeName, MtlxNode& out)
    {
      const std::string key = "name=\"" + nodeName + "\"";
      size_t p = 0;
      while ((p = doc.find(key, p)) != std::string::npos) {
        size_t lt = doc.rfind('<', p);
        if (lt == std::string::npos) break;
        size_t sp = doc.find_first_of(" \t\r\n>", lt);
        const std::string cat = doc.substr(lt + 1, sp - lt - 1);
        if (cat != "input" && cat != "output") {
          out.category = cat;
          size_t headEnd = doc.find('>', p);
          if (headEnd == std::string::npos || doc[headEnd - 1] == '/') return true;
          size_t close = doc.find("</" + cat, p), q = headEnd;
          while ((q = doc.find("<input", q)) != std::string::npos && (close == std::string::npos || q < close)) {
            size_t e = doc.find('>', q);
            if (e == std::string::npos) break;
            const std::string tag = doc.substr(q, e - q + 1), n = attr(tag, "name");
            if (!n.empty()) { out.inputs[n] = attr(tag, "value"); if (!attr(tag, "nodename").empty()) out.connections[n] = attr(tag, "nodename");
                              if (!attr(tag, "colorspace").empty()) out.inputs[n + ":colorspace"] = attr(tag, "colorspace"); }
            q = e;
          }
          return true;
        }
        p += key.size();
      }
      return false;
    }

    int floats(const std::string& s, float* out, int maxN)
    {
      int n = 0; const char* p = s.c_str();
      while (*p && n < maxN) {
        char* end = nullptr;
        float v = strtof(p, &end);
        if (end == p) { if (!strncmp(p, "true", 4)) { v = 1.0f; end = (char*)p + 4; } else if (!strncmp(p, "false", 5)) { v = 0.0f; end = (char*)p + 5;
            } else { p++; continue; } }
        out[n++] = v; p = end;
        while (*p == ',' || *p == ' ') p++;
      }
      return n;
    }

    void setN(const MtlxNode& n, const char* name, float* dst, int count)
    {
      auto it = n.inputs.find(name);
      if (it != n.inputs.end()) floats(it->second, dst, count);
    }

    // a material input fed by an image node (UsdUVTexture / image / tiledimage): file + UsdUVTexture's wrap, scale, bias, colour space
    struct ImageInput { std::string file; int wrapS = GI_C_TEX_WRAP_REPEAT, wrapT = GI_C_TEX_WRAP_REPEAT, channel = 0;
        float scale[4] = {1, 1, 1, 1}, bias[4] = {0, 0, 0, 0}; bool srgb = false;
                        bool hasXf = false; float xf[6] = {1, 0, 0, 0, 1, 0}; /* UsdTransform2d upstream of `st` */ };
    int wrapMode(const std::string& v)
    {
      if (v == "clamp") return GI_C_TEX_WRAP_CLAMP;
      if (v == "mirror") return GI_C_TEX_WRAP_MIRRORED_REPEAT;
      if (v == "black") return GI_C_TEX_WRAP_CLIP;
      return GI_C_TEX_WRAP_REPEAT; // "repeat", "useMetadata", unset
    }

    bool descFromMtlx(const char* src, GiCMaterialDesc& d, std::string (&primvars)[GI_C_TEX_SLOT_COUNT], ImageInput (&images)[GI_C_TEX_SLOT_COUNT])
    {
      MtlxNode n;
      if (!src || !findSurfaceNode(src, n)) return false;
      const std::string doc = src;
      // an input fed by a <constant> node (hdGatling's network patchers emit those for colour / float mismatches) is a constant
      for (auto it = n.connections.begin(); it != n.connections.end();) {
        MtlxNode up;
        if (readNode(doc, it->second, up) && up.category == "constant" && up.inputs.count("value")) { n.inputs[it->first] = up.inputs["value"];
            it = n.connections.erase(it); }
        else ++it;
      }
      // `constant` / `count`: where the input's constant lives in the parameter block -- a primvar reader's own fallback value (MaterialX <geompropvalue>'s
      // `default`, UsdPrimvarReader's `fallback`) replaces it: that is what a mesh WITHOUT the primvar shows (hdGatling's default material reads displayColor
      // with default 0.18, /root/reference/src/hdGatling/renderDelegate.cpp:64-78)
      auto bind = [&](const char* input, int slot, float* constant = nullptr, int count = 0) {
        auto it = n.connections.find(input);
        if (it == n.connections.end()) return;
        primvars[slot] = primvarOfNode(doc, it->second);
        MtlxNode up;
        if (!primvars[slot].empty()) {
          if (constant && readNode(doc, it->second, up)) {
            auto d = up.inputs.find("default");
            if (d == up.inputs.end()) d = up.inputs.find("fallback");
            if (d != up.inputs.end() && !d->second.empty()) { float v[4] = {0, 0, 0, 0}; const int got = floats(d->second, v, count < 4 ? count : 4);
                for (int i = 0; got > 0 && i < count; i++) constant[i] = v[i < got ? i : got - 1]; }
          }
          return;
        }
        if (!readNode(doc, it->second, up)) return;
        // Between the image and the surface input MaterialX documents put per-channel affine nodes -- `normalmap` (2 x - 1, xy times its `scale`), `multiply` /
        // `add` / `subtract` with a constant (tints, gains), `convert` / `dot` (pass-through) -- which fold into the binding's scale and bias: walking
        // upstream, the value seen by the surface is S * x + B of the node's input x.  Anything
        // else (a second texture, a procedural) ends the walk: the input keeps its constant.
        float S[4] = {1, 1, 1, 1}, B[4] = {0, 0, 0, 0};
        for (int depth = 0; depth < 8 && up.category != "UsdUVTexture" && up.category != "image" && up.category != "tiledimage"; depth++) {
          const std::string cat = up.category;
          float c[4] = {0, 0, 0, 0}; std::string next;
          auto constantOf = [&](const char* in) { auto v = up.inputs.find(in);
              if (v == up.inputs.end() || v->second.empty() || up.connections.count(in)) return false;
                                                  const int got = floats(v->second, c, 4); for (int i = got; got > 0 && i < 4; i++) c[i] = c[got - 1];
                                                      return got > 0; };
          if (cat == "normalmap") {
            if (!up.connections.count("in")) return;
            float k = 1.0f; if (up.inputs.count("scale") && !up.inputs["scale"].empty() && !up.connections.count("scale")) floats(up.inputs["scale"], &k, 1);
            const float kk[4] = {k, k, 1.0f, 1.0f};
            for (int i = 0; i < 4; i++) { B[i] += S[i] * -kk[i]; S[i] *= 2.0f * kk[i]; }
            next = up.connections["in"];
          } else if (cat == "multiply" || cat == "add" || cat == "subtract") {
            const bool c2 = constantOf("in2") && up.connections.count("in1"), c1 = !c2 && cat != "subtract" && constantOf("in1") && up.connections.count("in2");
            if (!c1 && !c2) return;
            next = up.connections[c2 ? "in1" : "in2"];
            for (int i = 0; i < 4; i++) { if (cat == "multiply") S[i] *= c[i]; else if (cat == "add") B[i] += S[i] * c[i]; else B[i] -= S[i] * c[i]; }
          } else if (cat == "convert" || cat == "dot") {
            if (!up.connections.count("in")) return;
            next = up.connections["in"];
          } else return;
          MtlxNode nx; if (!readNode(doc, next, nx)) return;
          up = nx;
        }
        if (up.category != "UsdUVTexture" && up.category != "image" && up.category != "tiledimage") return;
        ImageInput& im = images[slot];
        im.file = up.inputs["file"];
        im.wrapS = wrapMode(up.inputs.count("wrapS") ? up.inputs["wrapS"] : up.inputs["uaddressmode"]);
        im.wrapT = wrapMode(up.inputs.count("wrapT") ? up.inputs["wrapT"] : up.inputs["vaddressmode"]);
        if (up.inputs.count("scale")) floats(up.inputs["scale"], im.scale, 4);
        if (up.inputs.count("bias")) floats(up.inputs["bias"], im.bias, 4);
        // (identity walk: x * 1 + 0 -- the values as written)
        for (int i = 0; i < 4; i++) { im.bias[i] = B[i] + S[i] * im.bias[i]; im.scale[i] = S[i] * im.scale[i]; }
        const bool colour = slot == GI_C_TEX_BASE_COLOR || slot == GI_C_TEX_EMISSION || slot == GI_C_TEX_TRANSMISSION_COLOR;
        const std::string cs = up.inputs.count("sourceColorSpace") ? up.inputs["sourceColorSpace"] : "auto";
        im.srgb = cs == "sRGB" || (cs == "auto" && colour); // UsdUVTexture: auto = sRGB for 8-bit colour data
        if (up.category != "UsdUVTexture") // MaterialX image nodes name the file's colour space on the `file` input; without one the file is taken as linear
          im.srgb = up.inputs.count("file:colorspace") && up.inputs["file:colorspace"] == "srgb_texture";
        // texture coordinates through a UsdTransform2d (UsdPreviewSurface specification: result = in * scale, rotated counter-clockwise by `rotation` degrees,
        // + translation) -> the six floats of giCSetMaterialTextureTransform; cos / sin in double, rounded once (== gatling_amd/scene.py usd_transform_2d)
        auto stc = up.connections.find("st"); if (stc == up.connections.end()) stc = up.connections.find("texcoord");
        MtlxNode xfn;
        if (stc != up.connections.end() && readNode(doc, stc->second, xfn) && xfn.category == "UsdTransform2d") {
          float rot = 0.0f, sc2[2] = {1.0f, 1.0f}, tr[2] = {0.0f, 0.0f};
          if (xfn.inputs.count("rotation")) floats(xfn.inputs["rotation"], &rot, 1);
          if (xfn.inputs.count("scale")) floats(xfn.inputs["scale"], sc2, 2);
          if (xfn.inputs.count("translation")) floats(xfn.inputs["translation"], tr, 2);
          const double rad = (double)rot * 3.14159265358979323846 / 180.0, c = cos(rad), sn = sin(rad);
          im.hasXf = true;
          im.xf[0] = (float)(c * (double)sc2[0]); im.xf[1] = (float)(-sn * (double)sc2[1]); im.xf[2] = tr[0];
          im.xf[3] = (float)(sn * (double)sc2[0]); im.xf[4] = (float)(c * (double)sc2[1]); im.xf[5] = tr[1];
        }
        // which output feeds a scalar input: <input ... nodename="tex" output="g"/>
        const std::string needle = std::string("name=\"") + input + "\"";
        size_t q = doc.find(needle);
        if (q != std::string::npos) { size_t e = doc.find('>', q); const std::string tag = doc.substr(q, e - q); std::string o = attr(tag + " ", "output");
          if (n.outputs.count(input)) o = n.outputs[input];
          im.channel = o == "g" ? 1 : o == "b" ? 2 : o == "a" ? 3 : 0; }
      };
      memset(&d, 0, sizeof(d));
      float* p = d.p;
      if (n.category == "UsdPreviewSurface") {
        d.klass = GI_C_MAT_USD_PREVIEW_SURFACE; // fallback values of the UsdPreviewSurface specification
        p[GI_C_P_BASE_COLOR] = p[GI_C_P_BASE_COLOR + 1] = p[GI_C_P_BASE_COLOR + 2] = 0.18f;
        p[GI_C_P_ROUGHNESS] = 0.5f; p[GI_C_P_CLEARCOAT_ROUGHNESS] = 0.01f; p[GI_C_P_OPACITY] = 1.0f; p[GI_C_P_IOR] = 1.5f;
        setN(n, "diffuseColor", p + GI_C_P_BASE_COLOR, 3); setN(n, "emissiveColor", p + GI_C_P_EMISSION, 3);
        setN(n, "useSpecularWorkflow", p + GI_C_P_USE_SPECULAR_WORKFLOW, 1); setN(n, "specularColor", p + GI_C_P_SPECULAR_COLOR, 3);
        setN(n, "metallic", p + GI_C_P_METALLIC, 1); setN(n, "roughness", p + GI_C_P_ROUGHNESS, 1);
        setN(n, "clearcoat", p + GI_C_P_CLEARCOAT, 1); setN(n, "clearcoatRoughness", p + GI_C_P_CLEARCOAT_ROUGHNESS, 1);
        setN(n, "opacity", p + GI_C_P_OPACITY, 1); setN(n, "opacityThreshold", p + GI_C_P_OPACITY_THRESHOLD, 1); setN(n, "ior", p + GI_C_P_IOR, 1);
        bind("diffuseColor", GI_C_TEX_BASE_COLOR, p + GI_C_P_BASE_COLOR, 3); bind("emissiveColor", GI_C_TEX_EMISSION, p + GI_C_P_EMISSION, 3);
        bind("roughness", GI_C_TEX_ROUGHNESS, p + GI_C_P_ROUGHNESS, 1); bind("metallic", GI_C_TEX_METALLIC, p + GI_C_P_METALLIC, 1);
        bind("normal", GI_C_TEX_NORMAL); bind("opacity", GI_C_TEX_OPACITY); // typically the texture's alpha: <input name="opacity" nodename="tex" output="a"/>
        return true;
      }
      d.klass = GI_C_MAT_OPEN_PBR; // defaults: src/gi/mtlx/open_pbr_surface.mtlx:11-92
      p[GI_C_P_BASE_WEIGHT] = 1.0f; p[GI_C_P_BASE_COLOR] = p[GI_C_P_BASE_COLOR + 1] = p[GI_C_P_BASE_COLOR + 2] = 0.8f;
      p[GI_C_P_SPECULAR_WEIGHT] = 1.0f; p[GI_C_P_SPECULAR_COLOR] = p[GI_C_P_SPECULAR_COLOR + 1] = p[GI_C_P_SPECULAR_COLOR + 2] = 1.0f;
      p[GI_C_P_ROUGHNESS] = 0.3f; p[GI_C_P_IOR] = 1.5f; p[GI_C_P_OPACITY] = 1.0f;
      p[GI_C_P_TRANSMISSION_COLOR] = p[GI_C_P_TRANSMISSION_COLOR + 1] = p[GI_C_P_TRANSMISSION_COLOR + 2] = 1.0f;
      p[GI_C_P_COAT_COLOR] = p[GI_C_P_COAT_COLOR + 1] = p[GI_C_P_COAT_COLOR + 2] = 1.0f; p[GI_C_P_COAT_IOR] = 1.6f; p[GI_C_P_COAT_DARKENING] = 1.0f;
      p[GI_C_P_FUZZ_COLOR] = p[GI_C_P_FUZZ_COLOR + 1] = p[GI_C_P_FUZZ_COLOR + 2] = 1.0f; p[GI_C_P_FUZZ_ROUGHNESS] = 0.5f; p[GI_C_P_THIN_FILM_THICKNESS] = 0.5f;
          p[GI_C_P_THIN_FILM_IOR] = 1.4f;
      p[GI_C_P_SUBSURFACE_COLOR] = p[GI_C_P_SUBSURFACE_COLOR + 1] = p[GI_C_P_SUBSURFACE_COLOR + 2] = 0.8f;
      p[GI_C_P_SUBSURFACE_RADIUS] = 1.0f; p[GI_C_P_SUBSURFACE_RADIUS_SCALE] = 1.0f; p[GI_C_P_SUBSURFACE_RADIUS_SCALE + 1] = 0.5f;
          p[GI_C_P_SUBSURFACE_RADIUS_SCALE + 2] = 0.25f;
      float lum = 0.0f, ecol[3] = {1.0f, 1.0f, 1.0f};
      if (n.category == "standard_surface") {
        // Autodesk Standard Surface 1.0.1 (the reference compiles MaterialX's own standard_surface graph through MDL): read onto the OpenPBR closed forms,
        // input by input -- OpenPBR is that model's successor and keeps its layering (fuzz over coat over {metal | glass | subsurface | diffuse+specular}).
        // Defaults are the Standard Surface specification's, not OpenPBR's.  specular_rotation / coat_rotation (turns; the model's graph rotates the tangent by
        // 360 x it about the normal) are the tangents' turns.  What has no counterpart is dropped: transmission_dispersion, transmission_extra_roughness,
        // coat_affect_color / coat_affect_roughness, and coat_darkening stays 0 (the model has no such term).
        p[GI_C_P_BASE_WEIGHT] = 0.8f; p[GI_C_P_BASE_COLOR] = p[GI_C_P_BASE_COLOR + 1] = p[GI_C_P_BASE_COLOR + 2] = 1.0f; p[GI_C_P_ROUGHNESS] = 0.2f;
        p[GI_C_P_CLEARCOAT_ROUGHNESS] = 0.1f; p[GI_C_P_COAT_IOR] = 1.5f; p[GI_C_P_COAT_DARKENING] = 0.0f; p[GI_C_P_FUZZ_ROUGHNESS] = 0.3f;
        p[GI_C_P_SUBSURFACE_COLOR] = p[GI_C_P_SUBSURFACE_COLOR + 1] = p[GI_C_P_SUBSURFACE_COLOR + 2] = 1.0f;
        p[GI_C_P_SUBSURFACE_RADIUS] = 1.0f;
            p[GI_C_P_SUBSURFACE_RADIUS_SCALE] = p[GI_C_P_SUBSURFACE_RADIUS_SCALE + 1] = p[GI_C_P_SUBSURFACE_RADIUS_SCALE + 2] = 1.0f;
        p[GI_C_P_THIN_FILM_THICKNESS] = 0.0f; p[GI_C_P_THIN_FILM_IOR] = 1.5f;
        setN(n, "base", p + GI_C_P_BASE_WEIGHT, 1); setN(n, "base_color", p + GI_C_P_BASE_COLOR, 3);
            setN(n, "diffuse_roughness", p + GI_C_P_DIFFUSE_ROUGHNESS, 1);
        setN(n, "metalness", p + GI_C_P_METALLIC, 1); setN(n, "specular", p + GI_C_P_SPECULAR_WEIGHT, 1);
            setN(n, "specular_color", p + GI_C_P_SPECULAR_COLOR, 3);
        setN(n, "specular_roughness", p + GI_C_P_ROUGHNESS, 1); setN(n, "specular_IOR", p + GI_C_P_IOR, 1);
            setN(n, "specular_anisotropy", p + GI_C_P_SPECULAR_ANISOTROPY, 1);
        setN(n, "transmission", p + GI_C_P_TRANSMISSION_WEIGHT, 1); setN(n, "transmission_color", p + GI_C_P_TRANSMISSION_COLOR, 3);
        setN(n, "transmission_depth", p + GI_C_P_TRANSMISSION_DEPTH, 1); setN(n, "transmission_scatter", p + GI_C_P_TRANSMISSION_SCATTER, 3);
        setN(n, "transmission_scatter_anisotropy", p + GI_C_P_TRANSMISSION_SCATTER_ANISOTROPY, 1);
        setN(n, "subsurface", p + GI_C_P_SUBSURFACE_WEIGHT, 1); setN(n, "subsurface_color", p + GI_C_P_SUBSURFACE_COLOR, 3);
            setN(n, "subsurface_anisotropy", p + GI_C_P_SUBSURFACE_ANISOTROPY, 1);
        // mean free path = scale x radius (colour) = OpenPBR's radius x radius_scale
        setN(n, "subsurface_scale", p + GI_C_P_SUBSURFACE_RADIUS, 1); setN(n, "subsurface_radius", p + GI_C_P_SUBSURFACE_RADIUS_SCALE, 3);
        setN(n, "sheen", p + GI_C_P_FUZZ_WEIGHT, 1); setN(n, "sheen_color", p + GI_C_P_FUZZ_COLOR, 3); setN(n, "sheen_roughness", p + GI_C_P_FUZZ_ROUGHNESS, 1);
        setN(n, "coat", p + GI_C_P_CLEARCOAT, 1); setN(n, "coat_color", p + GI_C_P_COAT_COLOR, 3); setN(n, "coat_roughness", p + GI_C_P_CLEARCOAT_ROUGHNESS, 1);
        setN(n, "coat_IOR", p + GI_C_P_COAT_IOR, 1); setN(n, "coat_anisotropy", p + GI_C_P_COAT_ANISOTROPY, 1);
        setN(n, "coat_rotation", p + GI_C_P_COAT_ROTATION, 1); setN(n, "specular_rotation", p + GI_C_P_SPECULAR_ROTATION, 1);
        // nanometres, 0 = no film -> weight + micrometres
        float nm = 0.0f; setN(n, "thin_film_thickness", &nm, 1); setN(n, "thin_film_IOR", p + GI_C_P_THIN_FILM_IOR, 1);
        p[GI_C_P_THIN_FILM_WEIGHT] = nm > 0.0f ? 1.0f : 0.0f; p[GI_C_P_THIN_FILM_THICKNESS] = nm > 0.0f ? nm * 0.001f : 0.5f;
        // colour opacity -> its mean
        float op3[3] = {1.0f, 1.0f, 1.0f}; setN(n, "opacity", op3, 3); p[GI_C_P_OPACITY] = (op3[0] + op3[1] + op3[2]) * (1.0f / 3.0f);
        setN(n, "thin_walled", p + GI_C_P_THIN_WALLED, 1);
        setN(n, "emission", &lum, 1); setN(n, "emission_color", ecol, 3);
        for (int i = 0; i < 3; i++) p[GI_C_P_EMISSION + i] = lum * ecol[i];
        bind("base_color", GI_C_TEX_BASE_COLOR, p + GI_C_P_BASE_COLOR, 3); bind("specular_roughness", GI_C_TEX_ROUGHNESS, p + GI_C_P_ROUGHNESS, 1);
            bind("metalness", GI_C_TEX_METALLIC, p + GI_C_P_METALLIC, 1);
        bind("normal", GI_C_TEX_NORMAL); bind("opacity", GI_C_TEX_OPACITY); bind("coat_normal", GI_C_TEX_COAT_NORMAL);
        bind("transmission", GI_C_TEX_TRANSMISSION_WEIGHT, p + GI_C_P_TRANSMISSION_WEIGHT, 1);
            bind("transmission_color", GI_C_TEX_TRANSMISSION_COLOR, p + GI_C_P_TRANSMISSION_COLOR, 3);
        return true;
      }
      if (n.category == "gltf_pbr") {
        // MaterialX's glTF PBR node (KHR_materials_* folded in) onto the same closed forms.  Defaults are the node's (metallic 1, roughness 1, sheen off ...).
        // thickness 0 is glTF's thin-walled transmission; a thick one attenuates with attenuation_color over attenuation_distance (OpenPBR transmission_color
        // at transmission_depth).  alpha_mode: 0 OPAQUE (alpha ignored), 1 MASK (constant
        // alpha against alpha_cutoff), 2 BLEND (alpha as opacity: stochastic cutout).
        // Dropped: occlusion (baked ambient occlusion has no place in a path tracer), dispersion.
        p[GI_C_P_BASE_COLOR] = p[GI_C_P_BASE_COLOR + 1] = p[GI_C_P_BASE_COLOR + 2] = 1.0f; p[GI_C_P_METALLIC] = 1.0f; p[GI_C_P_ROUGHNESS] = 1.0f;
        p[GI_C_P_CLEARCOAT_ROUGHNESS] = 0.0f; p[GI_C_P_COAT_IOR] = 1.5f; p[GI_C_P_COAT_DARKENING] = 0.0f; p[GI_C_P_FUZZ_ROUGHNESS] = 0.0f;
            p[GI_C_P_THIN_FILM_IOR] = 1.3f;
        setN(n, "base_color", p + GI_C_P_BASE_COLOR, 3); setN(n, "metallic", p + GI_C_P_METALLIC, 1); setN(n, "roughness", p + GI_C_P_ROUGHNESS, 1);
        setN(n, "specular", p + GI_C_P_SPECULAR_WEIGHT, 1); setN(n, "specular_color", p + GI_C_P_SPECULAR_COLOR, 3); setN(n, "ior", p + GI_C_P_IOR, 1);
        setN(n, "transmission", p + GI_C_P_TRANSMISSION_WEIGHT, 1);
        float thick = 0.0f, attDist = 0.0f, attCol[3] = {1.0f, 1.0f, 1.0f};
        setN(n, "thickness", &thick, 1); setN(n, "attenuation_distance", &attDist, 1); setN(n, "attenuation_color", attCol, 3);
        p[GI_C_P_THIN_WALLED] = thick > 0.0f ? 0.0f : 1.0f;
        if (thick > 0.0f && attDist > 0.0f && attDist < 3.0e38f) { for (int i = 0; i < 3; i++) p[GI_C_P_TRANSMISSION_COLOR + i] = attCol[i];
            p[GI_C_P_TRANSMISSION_DEPTH] = attDist; }
        setN(n, "clearcoat", p + GI_C_P_CLEARCOAT, 1); setN(n, "clearcoat_roughness", p + GI_C_P_CLEARCOAT_ROUGHNESS, 1);
        // KHR_materials_sheen: black = off
        float sheen[3] = {0.0f, 0.0f, 0.0f}; setN(n, "sheen_color", sheen, 3); setN(n, "sheen_roughness", p + GI_C_P_FUZZ_ROUGHNESS, 1);
        const bool hasSheen = sheen[0] > 0.0f || sheen[1] > 0.0f || sheen[2] > 0.0f;
        p[GI_C_P_FUZZ_WEIGHT] = hasSheen ? 1.0f : 0.0f; for (int i = 0; i < 3; i++) p[GI_C_P_FUZZ_COLOR + i] = hasSheen ? sheen[i] : 1.0f;
        float nm = 100.0f; setN(n, "iridescence", p + GI_C_P_THIN_FILM_WEIGHT, 1); setN(n, "iridescence_ior", p + GI_C_P_THIN_FILM_IOR, 1);
            setN(n, "iridescence_thickness", &nm, 1);
        p[GI_C_P_THIN_FILM_THICKNESS] = nm * 0.001f; // nanometres -> micrometres
        setN(n, "anisotropy_strength", p + GI_C_P_SPECULAR_ANISOTROPY, 1);
        { float rad = 0.0f; setN(n, "anisotropy_rotation", &rad, 1); p[GI_C_P_SPECULAR_ROTATION] = rad / 6.2831855f; } // radians from the tangent -> turns
        float alpha = 1.0f, mode = 0.0f, cutoff = 0.5f; setN(n, "alpha", &alpha, 1); setN(n, "alpha_mode", &mode, 1); setN(n, "alpha_cutoff", &cutoff, 1);
        p[GI_C_P_OPACITY] = mode < 0.5f ? 1.0f : (mode < 1.5f ? (alpha >= cutoff ? 1.0f : 0.0f) : alpha);
        float strength = 1.0f; ecol[0] = ecol[1] = ecol[2] = 0.0f; setN(n, "emissive", ecol, 3); setN(n, "emissive_strength", &strength, 1);
        for (int i = 0; i < 3; i++) p[GI_C_P_EMISSION + i] = strength * ecol[i];
        bind("base_color", GI_C_TEX_BASE_COLOR, p + GI_C_P_BASE_COLOR, 3); bind("roughness", GI_C_TEX_ROUGHNESS, p + GI_C_P_ROUGHNESS, 1);
            bind("metallic", GI_C_TEX_METALLIC, p + GI_C_P_METALLIC, 1);
        bind("normal", GI_C_TEX_NORMAL); bind("clearcoat_normal", GI_C_TEX_COAT_NORMAL);
            bind("transmission", GI_C_TEX_TRANSMISSION_WEIGHT, p + GI_C_P_TRANSMISSION_WEIGHT, 1);
        if (mode >= 1.5f) bind("alpha", GI_C_TEX_OPACITY);
        return true;
      }
      setN(n, "base_weight", p + GI_C_P_BASE_WEIGHT, 1); setN(n, "base_color", p + GI_C_P_BASE_COLOR, 3);
      setN(n, "base_diffuse_roughness", p + GI_C_P_DIFFUSE_ROUGHNESS, 1); setN(n, "base_metalness", p + GI_C_P_METALLIC, 1);
      setN(n, "specular_weight", p + GI_C_P_SPECULAR_WEIGHT, 1); setN(n, "specular_color", p + GI_C_P_SPECULAR_COLOR, 3);
      setN(n, "specular_roughness", p + GI_C_P_ROUGHNESS, 1); setN(n, "specular_ior", p + GI_C_P_IOR, 1);
      setN(n, "transmission_weight", p + GI_C_P_TRANSMISSION_WEIGHT, 1); setN(n, "transmission_color", p + GI_C_P_TRANSMISSION_COLOR, 3);
      setN(n, "transmission_depth", p + GI_C_P_TRANSMISSION_DEPTH, 1); setN(n, "transmission_scatter", p + GI_C_P_TRANSMISSION_SCATTER, 3);
      setN(n, "transmission_scatter_anisotropy", p + GI_C_P_TRANSMISSION_SCATTER_ANISOTROPY, 1);
      setN(n, "coat_weight", p + GI_C_P_CLEARCOAT, 1); setN(n, "coat_color", p + GI_C_P_COAT_COLOR, 3);
      setN(n, "coat_roughness", p + GI_C_P_CLEARCOAT_ROUGHNESS, 1); setN(n, "coat_ior", p + GI_C_P_COAT_IOR, 1);
          setN(n, "coat_darkening", p + GI_C_P_COAT_DARKENING, 1);
      setN(n, "fuzz_weight", p + GI_C_P_FUZZ_WEIGHT, 1); setN(n, "fuzz_color", p + GI_C_P_FUZZ_COLOR, 3);
          setN(n, "fuzz_roughness", p + GI_C_P_FUZZ_ROUGHNESS, 1);
      setN(n, "geometry_thin_walled", p + GI_C_P_THIN_WALLED, 1);
      setN(n, "subsurface_weight", p + GI_C_P_SUBSURFACE_WEIGHT, 1); setN(n, "subsurface_color", p + GI_C_P_SUBSURFACE_COLOR, 3);
      setN(n, "subsurface_scatter_anisotropy", p + GI_C_P_SUBSURFACE_ANISOTROPY, 1);
      // the volumetric form's mean free path
      setN(n, "subsurface_radius", p + GI_C_P_SUBSURFACE_RADIUS, 1); setN(n, "subsurface_radius_scale", p + GI_C_P_SUBSURFACE_RADIUS_SCALE, 3);
      setN(n, "specular_roughness_anisotropy", p + GI_C_P_SPECULAR_ANISOTROPY, 1); setN(n, "coat_roughness_anisotropy", p + GI_C_P_COAT_ANISOTROPY, 1);
      setN(n, "thin_film_weight", p + GI_C_P_THIN_FILM_WEIGHT, 1); setN(n, "thin_film_thickness", p + GI_C_P_THIN_FILM_THICKNESS, 1);
          setN(n, "thin_film_ior", p + GI_C_P_THIN_FILM_IOR, 1);
      // [ext] the tangents' turns as plain floats (what the flat parameter vocabularies below carry)
      setN(n, "coat_rotation", p + GI_C_P_COAT_ROTATION, 1); setN(n, "specular_rotation", p + GI_C_P_SPECULAR_ROTATION, 1);
      // geometry_tangent / geometry_coat_tangent (open_pbr_surface.mtlx:89, 91; 385 ... 457, 561) as documents feed them: <rotate3d in=tangent amount=DEGREES
      // axis=normal>, optionally behind a <normalize> -- the turn of that tangent, degrees / 360.  Anything else upstream keeps the geometry tangent.
      auto tangentTurn = [&](const char* input, int slot) {
        auto it = n.connections.find(input);
        MtlxNode up;
        if (it == n.connections.end() || !readNode(doc, it->second, up)) return;
        if (up.category == "normalize" && up.connections.count("in")) {
          const std::string src2 = up.connections["in"];
          up = MtlxNode(); (void)readNode(doc, src2, up);
        }
        if (up.category == "rotate3d" && up.inputs.count("amount")) {
          float deg = 0.0f; floats(up.inputs["amount"], &deg, 1);
          p[slot] = deg / 360.0f;
        }
      };
      tangentTurn("geometry_coat_tangent", GI_C_P_COAT_ROTATION); tangentTurn("geometry_tangent", GI_C_P_SPECULAR_ROTATION);
      setN(n, "emission_luminance", &lum, 1); setN(n, "emission_color", ecol, 3); setN(n, "geometry_opacity", p + GI_C_P_OPACITY, 1);
      for (int i = 0; i < 3; i++) p[GI_C_P_EMISSION + i] = lum * ecol[i];
      bind("base_color", GI_C_TEX_BASE_COLOR, p + GI_C_P_BASE_COLOR, 3); bind("specular_roughness", GI_C_TEX_ROUGHNESS, p + GI_C_P_ROUGHNESS, 1);
          bind("base_metalness", GI_C_TEX_METALLIC, p + GI_C_P_METALLIC, 1);
      bind("geometry_normal", GI_C_TEX_NORMAL); bind("geometry_opacity", GI_C_TEX_OPACITY); bind("geometry_coat_normal", GI_C_TEX_COAT_NORMAL);
      bind("transmission_weight", GI_C_TEX_TRANSMISSION_WEIGHT, p + GI_C_P_TRANSMISSION_WEIGHT, 1);
          bind("transmission_color", GI_C_TEX_TRANSMISSION_COLOR, p + GI_C_P_TRANSMISSION_COLOR, 3);
      return true;
    }
  }

  GiStatus giInitialize(const GiInitParams&)
  {
    const char* dev = getenv("GATLING_DEVICE");
    return giCInitialize(dev ? atoi(dev) : 0) == GI_C_OK ? GiStatus::Ok : GiStatus::Error;
  }
  void giTerminate() { giCTerminate(); }
  // Gi.h:201.  The reference hands the reader to its texture manager, which opens EVERY image through it (TextureManager.cpp:39-52; hdGatling's is backed by
  // ArResolver, rendererPlugin.cpp:95-143, 189): the C++ object becomes the C ABI's four callbacks.
  void giRegisterAssetReader(GiAssetReader* reader)
  {
    s_assetReader = reader;
    if (!reader) { giCRegisterAssetReader(nullptr); return; }
    GiCAssetReader r{};
    r.user = reader;
    r.open = [](void* u, const char* path) -> void* { return static_cast<GiAssetReader*>(u)->open(path); };
    r.size = [](void* u, void* a) -> uint64_t { return (uint64_t)static_cast<GiAssetReader*>(u)->size(static_cast<GiAsset*>(a)); };
    r.data = [](void* u, void* a) -> const void* { return static_cast<GiAssetReader*>(u)->data(static_cast<GiAsset*>(a)); };
    r.close = [](void* u, void* a) { static_cast<GiAssetReader*>(u)->close(static_cast<GiAsset*>(a)); };
    giCRegisterAssetReader(&r);
  }

  static GiMaterial* makeMaterial(GiScene* scene, const char* name, const GiCMaterialDesc& d, const std::string (&primvars)[GI_C_TEX_SLOT_COUNT],
      const ImageInput (&images)[GI_C_TEX_SLOT_COUNT])
  {
    GiCMaterial* h = giCCreateMaterial(scene->h, name, &d);
    if (!h) return nullptr;
    auto* mat = new GiMaterial{h, {}};
    for (int slot = 0; slot < GI_C_TEX_SLOT_COUNT; slot++) {
      if (!primvars[slot].empty()) giCSetMaterialPrimvarInput(h, slot, primvars[slot].c_str());
      const ImageInput& im = images[slot];
      if (im.file.empty()) continue;
      GiCTexture* t = giCCreateTextureFromFile(scene->h, im.file.c_str(), im.srgb ? 1 : 0); // .png / .jpg / .hdr / .pfm; others: the input keeps its constant
      if (!t) continue;
      mat->textures.push_back(t);
      GiCTextureBinding b{t, im.wrapS, im.wrapT, im.channel, {im.scale[0], im.scale[1], im.scale[2], im.scale[3]},
          {im.bias[0], im.bias[1], im.bias[2], im.bias[3]}};
      giCSetMaterialTexture(h, slot, &b);
      if (im.hasXf) giCSetMaterialTextureTransform(h, slot, im.xf);
    }
    return mat;
  }

  GiMaterial* giCreateMaterialFromMtlxStr(GiScene* scene, const char* name, const char* mtlxSrc)
  {
    GiCMaterialDesc d; std::string primvars[GI_C_TEX_SLOT_COUNT]; ImageInput images[GI_C_TEX_SLOT_COUNT];
    if (!scene || !descFromMtlx(mtlxSrc, d, primvars, images)) return nullptr;
    return makeMaterial(scene, name, d, primvars, images);
  }

  // [ext] C-linkage doors to the two material routes above, for harnesses that cannot call C++ (ctypes: tests/test_mtlx_parity.py drives the SAME reader
  // hdGatling's documents go through and compares the image with the parameter-block route).  The wrapper object is dropped; the GiCMaterial lives on.
  extern "C" GiCMaterial* gtlCreateMaterialFromMtlxStrC(GiCScene* scene, const char* name, const char* mtlxSrc)
  {
    if (!scene) return nullptr;
    GiScene wrap{scene};
    GiMaterial* m = giCreateMaterialFromMtlxStr(&wrap, name, mtlxSrc);
    if (!m) return nullptr;
    GiCMaterial* h = m->h;
    delete m; // (file textures created for the document stay owned by the scene)
    return h;
  }
  // [ext] what the reader makes of the image node feeding `slot`: 0 = no image, 1 = image, 2 = image with a texture-coordinate transform (xf6 filled)
  extern "C" int gtlMtlxImageInputC(const char* mtlxSrc, int slot, float* xf6, char* fileOut, int fileCap)
  {
    std::string primvars[GI_C_TEX_SLOT_COUNT]; ImageInput images[GI_C_TEX_SLOT_COUNT];
    GiCMaterialDesc d;
    if (slot < 0 || slot >= GI_C_TEX_SLOT_COUNT || !descFromMtlx(mtlxSrc, d, primvars, images) || images[slot].file.empty()) return 0;
    if (fileOut && fileCap > 0) { strncpy(fileOut, images[slot].file.c_str(), (size_t)fileCap - 1); fileOut[fileCap - 1] = 0; }
    if (xf6) memcpy(xf6, images[slot].xf, sizeof(float) * 6);
    return images[slot].hasXf ? 2 : 1;
  }
  // [ext] the rest of that binding: scale[4], bias[4] (upstream normalmap / multiply / add nodes folded in), {sRGB decode, channel}; same return value
  extern "C" int gtlMtlxImageBindingC(const char* mtlxSrc, int slot, float* scale4, float* bias4, int* srgbChannel2)
  {
    std::string primvars[GI_C_TEX_SLOT_COUNT]; ImageInput images[GI_C_TEX_SLOT_COUNT];
    GiCMaterialDesc d;
    if (slot < 0 || slot >= GI_C_TEX_SLOT_COUNT || !descFromMtlx(mtlxSrc, d, primvars, images) || images[slot].file.empty()) return 0;
    if (scale4) memcpy(scale4, images[slot].scale, sizeof(float) * 4);
    if (bias4) memcpy(bias4, images[slot].bias, sizeof(float) * 4);
    if (srgbChannel2) { srgbChannel2[0] = images[slot].srgb ? 1 : 0; srgbChannel2[1] = images[slot].channel; }
    return images[slot].hasXf ? 2 : 1;
  }
  extern "C" int gtlMaterialDescFromMtlxStrC(const char* mtlxSrc, GiCMaterialDesc* out)
  {
    std::string primvars[GI_C_TEX_SLOT_COUNT]; ImageInput images[GI_C_TEX_SLOT_COUNT];
    GiCMaterialDesc d;
    if (!out || !descFromMtlx(mtlxSrc, d, primvars, images)) return GI_C_ERROR;
    *out = d;
    return GI_C_OK;
  }

  // MaterialX documents (hdGatling's path for every UsdPreviewSurface / MaterialX network, materialNetworkCompiler.cpp:667-686): the
  // document is serialised by gtl_shim_mtlx.cpp -- the one translation unit that needs the MaterialX headers -- which registers itself here.
  static GtlMtlxDocToXml s_docToXml = nullptr;
  void gtlRegisterMtlxDocSerializer(GtlMtlxDocToXml fn) { s_docToXml = fn; }
  GiMaterial* giCreateMaterialFromMtlxDoc(GiScene* scene, const char* name, const std::shared_ptr<void> doc)
  {
    if (!s_docToXml) { fprintf(stderr,
        "[gatling_gi] giCreateMaterialFromMtlxDoc: built without gtl_shim_mtlx.cpp (MaterialX headers not found at build time)\n"); return nullptr; }
    if (!doc) return nullptr;
    const std::string xml = s_docToXml(doc);
    return giCreateMaterialFromMtlxStr(scene, name, xml.c_str());
  }

  // MDL modules (materialNetworkCompiler.cpp:619-665): no MDL compiler here; parameters are matched by NAME against three vocabularies --
  // the OmniPBR family the reference ships (src/gi/mdl/OmniPBR.mdl:53-224), UsdPreviewSurface inputs, open_pbr_surface inputs.
  GiMaterial* giCreateMaterialFromMdlFile(GiScene* scene, const char* name, const char* filePath, const char* subIdentifier, const GiMaterialParameters& params)
  {
    if (!scene) return nullptr;
    auto num = [&](const char* key, float* dst, int n) -> bool {
      auto it = params.find(key);
      if (it == params.end()) return false;
      const GiMaterialParameterValue& v = it->second;
      float tmp[4] = {0, 0, 0, 0}; int have = 0;
      if (auto* b = std::get_if<bool>(&v)) { tmp[0] = *b ? 1.0f : 0.0f; have = 1; }
      else if (auto* i = std::get_if<int>(&v)) { tmp[0] = (float)*i; have = 1; }
      else if (auto* f = std::get_if<float>(&v)) { tmp[0] = *f; have = 1; }
      else if (auto* a = std::get_if<GbVec2f>(&v)) { tmp[0] = a->x; tmp[1] = a->y; have = 2; }
      else if (auto* a3 = std::get_if<GbVec3f>(&v)) { tmp[0] = a3->x; tmp[1] = a3->y; tmp[2] = a3->z; have = 3; }
      else if (auto* a4 = std::get_if<GbVec4f>(&v)) { tmp[0] = a4->x; tmp[1] = a4->y; tmp[2] = a4->z; tmp[3] = a4->w; have = 4; }
      else if (auto* c = std::get_if<GbColor>(&v)) { tmp[0] = c->r; tmp[1] = c->g; tmp[2] = c->b; have = 3; }
      else return false;
      for (int k = 0; k < n; k++) dst[k] = have == 1 ? tmp[0] : tmp[k < have ? k : have - 1];
      return true;
    };
    GiCMaterialDesc d; memset(&d, 0, sizeof(d));
    std::string primvars[GI_C_TEX_SLOT_COUNT]; ImageInput images[GI_C_TEX_SLOT_COUNT];
    auto tex = [&](const char* key, int slot, int channel) -> bool {
      auto it = params.find(key);
      if (it == params.end()) return false;
      const GbTextureAsset* a = std::get_if<GbTextureAsset>(&it->second);
      if (!a || a->absPath.empty()) return false;
      images[slot].file = a->absPath; images[slot].srgb = a->isSrgb; images[slot].channel = channel;
      return true;
    };
    float* p = d.p;
    const std::string module = std::string(filePath ? filePath : "") + "::" + (subIdentifier ? subIdentifier : "");
    bool known = false;
    if (params.count("base_color") || params.count("specular_roughness") || params.count("base_metalness") || module.find("open_pbr") != std::string::npos) {
      // open_pbr_surface vocabulary: the same defaults and inputs as the MaterialX route (src/gi/mtlx/open_pbr_surface.mtlx:11-92)
      std::string xml = "<materialx version=\"1.39\"><open_pbr_surface name=\"m\" type=\"surfaceshader\">";
      // (name, components): scalars are written as one value, colours as three -- the same spelling a MaterialX document uses
      static const struct { const char* name; int n; } kOpbr[] = {
        {"base_weight", 1}, {"base_color", 3}, {"base_diffuse_roughness", 1}, {"base_metalness", 1}, {"specular_weight", 1}, {"specular_color", 3},
            {"specular_roughness", 1},
        {"specular_ior", 1}, {"transmission_weight", 1}, {"transmission_color", 3}, {"transmission_depth", 1}, {"transmission_scatter", 3},
            {"transmission_scatter_anisotropy", 1},
        {"coat_weight", 1}, {"coat_color", 3}, {"coat_roughness", 1}, {"coat_ior", 1}, {"coat_darkening", 1}, {"emission_luminance", 1}, {"emission_color", 3},
        {"geometry_opacity", 1}, {"fuzz_weight", 1}, {"fuzz_color", 3}, {"fuzz_roughness", 1}, {"geometry_thin_walled", 1},
        {"subsurface_weight", 1}, {"subsurface_color", 3}, {"subsurface_scatter_anisotropy", 1}, {"specular_roughness_anisotropy", 1},
            {"coat_roughness_anisotropy", 1}, {"coat_rotation", 1}, {"specular_rotation", 1},
        {"thin_film_weight", 1}, {"thin_film_thickness", 1}, {"thin_film_ior", 1}};
      for (const auto& k : kOpbr) {
        float v[3];
        if (!num(k.name, v, k.n)) continue;
        char buf[200];
        if (k.n == 1) snprintf(buf, sizeof(buf), "<input name=\"%s\" value=\"%.9g\" />", k.name, v[0]);
        else snprintf(buf, sizeof(buf), "<input name=\"%s\" value=\"%.9g, %.9g, %.9g\" />", k.name, v[0], v[1], v[2]);
        xml += buf;
      }
      xml += "</open_pbr_surface></materialx>";
      if (!descFromMtlx(xml.c_str(), d, primvars, images)) return nullptr;
      tex("base_color", GI_C_TEX_BASE_COLOR, 0); tex("specular_roughness", GI_C_TEX_ROUGHNESS, 0); tex("base_metalness", GI_C_TEX_METALLIC, 0);
      known = true;
    } else {
      d.klass = GI_C_MAT_USD_PREVIEW_SURFACE;
      p[GI_C_P_BASE_COLOR] = p[GI_C_P_BASE_COLOR + 1] = p[GI_C_P_BASE_COLOR + 2] = 0.18f;
      p[GI_C_P_ROUGHNESS] = 0.5f; p[GI_C_P_CLEARCOAT_ROUGHNESS] = 0.01f; p[GI_C_P_OPACITY] = 1.0f; p[GI_C_P_IOR] = 1.5f;
      // UsdPreviewSurface vocabulary
      known |= num("diffuseColor", p + GI_C_P_BASE_COLOR, 3); known |= num("emissiveColor", p + GI_C_P_EMISSION, 3);
      known |= num("useSpecularWorkflow", p + GI_C_P_USE_SPECULAR_WORKFLOW, 1); known |= num("specularColor", p + GI_C_P_SPECULAR_COLOR, 3);
      known |= num("metallic", p + GI_C_P_METALLIC, 1); known |= num("roughness", p + GI_C_P_ROUGHNESS, 1);
      known |= num("clearcoat", p + GI_C_P_CLEARCOAT, 1); known |= num("clearcoatRoughness", p + GI_C_P_CLEARCOAT_ROUGHNESS, 1);
      known |= num("opacity", p + GI_C_P_OPACITY, 1); known |= num("opacityThreshold", p + GI_C_P_OPACITY_THRESHOLD, 1); known |= num("ior", p + GI_C_P_IOR, 1);
      // OmniPBR vocabulary (OmniPBR.mdl:53-224; module defaults: diffuse 0.2, roughness 0.5, metallic 0, emission off)
      const bool omni = module.find("OmniPBR") != std::string::npos;
      if (omni) { p[GI_C_P_BASE_COLOR] = p[GI_C_P_BASE_COLOR + 1] = p[GI_C_P_BASE_COLOR + 2] = 0.2f; known = true; }
      float tint[3] = {1.0f, 1.0f, 1.0f};
      known |= num("diffuse_color_constant", p + GI_C_P_BASE_COLOR, 3);
      if (num("diffuse_tint", tint, 3)) for (int k = 0; k < 3; k++) p[GI_C_P_BASE_COLOR + k] *= tint[k];
      known |= num("reflection_roughness_constant", p + GI_C_P_ROUGHNESS, 1); known |= num("metallic_constant", p + GI_C_P_METALLIC, 1);
      float enableEm = 0.0f, emCol[3] = {1.0f, 0.1f, 0.1f}, emInt = 40.0f;
      num("enable_emission", &enableEm, 1); num("emissive_color", emCol, 3); num("emissive_intensity", &emInt, 1);
      // (cd/m2: the reference's unit handling is the MDL SDK's; not pinned)
      if (enableEm != 0.0f) for (int k = 0; k < 3; k++) p[GI_C_P_EMISSION + k] = emCol[k] * emInt;
      float enableOp = 0.0f, opC = 1.0f;
      num("enable_opacity", &enableOp, 1);
      if (enableOp != 0.0f && num("opacity_constant", &opC, 1)) { p[GI_C_P_OPACITY] = opC; num("opacity_threshold", p + GI_C_P_OPACITY_THRESHOLD, 1); }
      if (omni && module.find("ClearCoat") != std::string::npos) { float en = 0.0f, w = 1.0f; num("enable_clearcoat", &en, 1); num("clearcoat_weight", &w, 1);
          if (en != 0.0f) p[GI_C_P_CLEARCOAT] = w; num("clearcoat_roughness", p + GI_C_P_CLEARCOAT_ROUGHNESS, 1); }
      if (tex("diffuse_texture", GI_C_TEX_BASE_COLOR, 0) || tex("diffuseColor", GI_C_TEX_BASE_COLOR, 0)) { known = true;
          for (int k = 0; k < 3; k++) images[GI_C_TEX_BASE_COLOR].scale[k] = tint[k]; }
      float infl = 0.0f;
      if (num("reflection_roughness_texture_influence", &infl, 1) && infl > 0.0f && tex("reflectionroughness_texture", GI_C_TEX_ROUGHNESS, 0)) {
        // lerp(constant, texel, influence)
        images[GI_C_TEX_ROUGHNESS].scale[0] = infl; images[GI_C_TEX_ROUGHNESS].bias[0] = p[GI_C_P_ROUGHNESS] * (1.0f - infl);
      }
      if (num("metallic_texture_influence", &infl, 1) && infl > 0.0f && tex("metallic_texture", GI_C_TEX_METALLIC, 0)) {
        images[GI_C_TEX_METALLIC].scale[0] = infl; images[GI_C_TEX_METALLIC].bias[0] = p[GI_C_P_METALLIC] * (1.0f - infl);
      }
      if (enableEm != 0.0f && tex("emissive_color_texture", GI_C_TEX_EMISSION, 0)) for (int k = 0; k < 3; k++) images[GI_C_TEX_EMISSION].scale[k] = emInt;
      tex("normalmap_texture", GI_C_TEX_NORMAL, 0); tex("normal", GI_C_TEX_NORMAL, 0);
      if (enableOp != 0.0f) { float ot = 0.0f; num("enable_opacity_texture", &ot, 1); if (ot != 0.0f) tex("opacity_texture", GI_C_TEX_OPACITY, 0); }
    }
    if (!known) { fprintf(stderr,
        "[gatling_gi] giCreateMaterialFromMdlFile(%s): no MDL compiler and no recognised parameter -- the delegate's default material is used\n",
        module.c_str()); return nullptr; }
    return makeMaterial(scene, name, d, primvars, images);
  }
  void giDestroyMaterial(GiMaterial* mat)
  {
    if (!mat) return;
    giCDestroyMaterial(mat->h);
    for (GiCTexture* t : mat->textures) giCDestroyTexture(t);
    delete mat;
  }

  static void setPrimvars(GiCMesh* h, const std::vector<GiPrimvarData>& pv, bool instancer)
  {
    std::vector<GiCPrimvarData> c(pv.size());
    for (size_t i = 0; i < pv.size(); i++) c[i] = GiCPrimvarData{pv[i].name.c_str(), (int32_t)pv[i].type, (int32_t)pv[i].interpolation, pv[i].data.data(),
        (uint64_t)pv[i].data.size()};
    if (instancer) giCSetMeshInstancerPrimvars(h, (uint32_t)c.size(), c.data()); else giCSetMeshPrimvars(h, (uint32_t)c.size(), c.data());
  }

  GiMesh* giCreateMesh(GiScene* scene, const GiMeshDesc& d)
  {
    static_assert(sizeof(GiVertex) == sizeof(GiCVertex) && sizeof(GiFace) == sizeof(GiCFace), "layouts must match Gi.h:110-122");
    GiCMeshDesc c{};
    // counts come from the vectors: the reference ignores desc.faceCount / vertexCount (Gi.cpp:628 passes the vectors on) and hdGatling
    // leaves both at zero (designated initialisers, mesh.cpp:1092-1102)
    c.faceCount = (uint32_t)d.faces.size(); c.faces = reinterpret_cast<const GiCFace*>(d.faces.data());
    c.faceIds = d.faceIds.size() >= d.faces.size() && !d.faces.empty() ? d.faceIds.data() : nullptr;
    c.id = d.id; c.isDoubleSided = d.isDoubleSided; c.isLeftHanded = d.isLeftHanded; c.name = d.name; c.maxFaceId = d.maxFaceId;
    c.vertexCount = (uint32_t)d.vertices.size(); c.vertices = reinterpret_cast<const GiCVertex*>(d.vertices.data());
    GiCMesh* h = scene ? giCCreateMesh(scene->h, &c) : nullptr;
    if (!h) return nullptr;
    setPrimvars(h, d.primvars, false);
    return new GiMesh{h};
  }
  void giSetMeshTransform(GiMesh* m, const float* mat4x4) { giCSetMeshTransform(m->h, mat4x4); }
  void giSetMeshInstanceTransforms(GiMesh* m, uint32_t count, const float (*t)[4][4])
  { giCSetMeshInstanceTransforms(m->h, count, reinterpret_cast<const float*>(t)); }
  void giSetMeshInstancerPrimvars(GiMesh* m, const std::vector<GiPrimvarData>& pv) { setPrimvars(m->h, pv, true); }
  void giSetMeshInstanceIds(GiMesh* m, uint32_t count, int* ids) { giCSetMeshInstanceIds(m->h, count, ids); }
  void giSetMeshMaterial(GiMesh* m, GiMaterial* mat) { giCSetMeshMaterial(m->h, mat ? mat->h : nullptr); }
  void giSetMeshVisibility(GiMesh* m, bool visible) { giCSetMeshVisibility(m->h, visible); }
  void giDestroyMesh(GiMesh* m) { if (!m) return; giCDestroyMesh(m->h); delete m; }

  GiStatus giRender(const GiRenderParams& p)
  {
    std::vector<GiCAovBinding> aovs(p.aovBindings.size());
    for (size_t i = 0; i < aovs.size(); i++) {
      aovs[i].aovId = int(p.aovBindings[i].aovId);
      memcpy(aovs[i].clearValue, p.aovBindings[i].clearValue, GI_C_MAX_AOV_COMP_SIZE);
      aovs[i].renderBuffer = p.aovBindings[i].renderBuffer ? p.aovBindings[i].renderBuffer->h : nullptr;
    }
    GiCRenderParams c{};
    c.aovBindings = aovs.data(); c.aovBindingCount = uint32_t(aovs.size());
    static_assert(sizeof(GiCameraDesc) == sizeof(GiCCameraDesc), "layouts must match Gi.h:96-108");
    memcpy(&c.camera, &p.camera, sizeof(GiCCameraDesc));
    c.domeLight = p.domeLight ? p.domeLight->h : nullptr;
    const GiRenderSettings& s = p.renderSettings;
    c.renderSettings = {s.clippingPlanes, s.depthOfField, s.domeLightCameraVisible, s.filterImportanceSampling, s.frame, s.jitteredSampling,
                        s.lightIntensityMultiplier, s.maxBounces, s.maxSampleValue, s.maxVolumeWalkLength, s.mediumStackSize, s.metersPerSceneUnit,
                        s.nextEventEstimation, s.progressiveAccumulation, s.rrBounceOffset, s.rrInvMinTermProb, s.spp, s.time};
    c.scene = p.scene ? p.scene->h : nullptr;
    return giCRender(&c) == GI_C_OK ? GiStatus::Ok : GiStatus::Error;
  }

  GiScene* giCreateScene() { GiCScene* h = giCCreateScene(); return h ? new GiScene{h} : nullptr; }
  void giDestroyScene(GiScene* s) { if (!s) return; giCDestroyScene(s->h); delete s; }

#define GTL_LIGHT(Type, CType)                                                                                          \
  Gi##Type##Light* giCreate##Type##Light(GiScene* scene) \
  { auto* h = scene ? giCCreate##Type##Light(scene->h) : nullptr; return h ? new Gi##Type##Light{h} : nullptr; } \
  void giDestroy##Type##Light(GiScene* scene, Gi##Type##Light* l) { if (!l) return; giCDestroy##Type##Light(scene->h, l->h); delete l; }
  GTL_LIGHT(Sphere, GiCSphereLight) GTL_LIGHT(Distant, GiCDistantLight) GTL_LIGHT(Rect, GiCRectLight) GTL_LIGHT(Disk, GiCDiskLight)
#undef GTL_LIGHT

  void giSetSphereLightPosition(GiSphereLight* l, float* v) { giCSetSphereLightPosition(l->h, v); }
  void giSetSphereLightBaseEmission(GiSphereLight* l, float* v) { giCSetSphereLightBaseEmission(l->h, v); }
  void giSetSphereLightRadius(GiSphereLight* l, float x, float y, float z) { giCSetSphereLightRadius(l->h, x, y, z); }
  void giSetSphereLightDiffuseSpecular(GiSphereLight* l, float d, float s) { giCSetSphereLightDiffuseSpecular(l->h, d, s); }
  void giSetDistantLightDirection(GiDistantLight* l, float* v) { giCSetDistantLightDirection(l->h, v); }
  void giSetDistantLightBaseEmission(GiDistantLight* l, float* v) { giCSetDistantLightBaseEmission(l->h, v); }
  void giSetDistantLightAngle(GiDistantLight* l, float a) { giCSetDistantLightAngle(l->h, a); }
  void giSetDistantLightDiffuseSpecular(GiDistantLight* l, float d, float s) { giCSetDistantLightDiffuseSpecular(l->h, d, s); }
  void giSetRectLightOrigin(GiRectLight* l, float* v) { giCSetRectLightOrigin(l->h, v); }
  void giSetRectLightTangents(GiRectLight* l, float* t0, float* t1) { giCSetRectLightTangents(l->h, t0, t1); }
  void giSetRectLightBaseEmission(GiRectLight* l, float* v) { giCSetRectLightBaseEmission(l->h, v); }
  void giSetRectLightDimensions(GiRectLight* l, float w, float h) { giCSetRectLightDimensions(l->h, w, h); }
  void giSetRectLightDiffuseSpecular(GiRectLight* l, float d, float s) { giCSetRectLightDiffuseSpecular(l->h, d, s); }
  void giSetDiskLightOrigin(GiDiskLight* l, float* v) { giCSetDiskLightOrigin(l->h, v); }
  void giSetDiskLightTangents(GiDiskLight* l, float* t0, float* t1) { giCSetDiskLightTangents(l->h, t0, t1); }
  void giSetDiskLightBaseEmission(GiDiskLight* l, float* v) { giCSetDiskLightBaseEmission(l->h, v); }
  void giSetDiskLightRadius(GiDiskLight* l, float x, float y) { giCSetDiskLightRadius(l->h, x, y); }
  void giSetDiskLightDiffuseSpecular(GiDiskLight* l, float d, float s) { giCSetDiskLightDiffuseSpecular(l->h, d, s); }

  GiDomeLight* giCreateDomeLight(GiScene* scene, const char* filePath)
  { auto* h = scene ? giCCreateDomeLight(scene->h, filePath) : nullptr; return h ? new GiDomeLight{h} : nullptr; }
  void giDestroyDomeLight(GiDomeLight* l) { if (!l) return; giCDestroyDomeLight(l->h); delete l; }
  void giSetDomeLightRotation(GiDomeLight* l, float* q) { giCSetDomeLightRotation(l->h, q); }
  void giSetDomeLightBaseEmission(GiDomeLight* l, float* v) { giCSetDomeLightBaseEmission(l->h, v); }
  void giSetDomeLightDiffuseSpecular(GiDomeLight* l, float d, float s) { giCSetDomeLightDiffuseSpecular(l->h, d, s); }

  GiRenderBuffer* giCreateRenderBuffer(uint32_t w, uint32_t h, GiRenderBufferFormat f)
  { auto* b = giCCreateRenderBuffer(w, h, int(f)); return b ? new GiRenderBuffer{b} : nullptr; }
  void giDestroyRenderBuffer(GiRenderBuffer* b) { if (!b) return; giCDestroyRenderBuffer(b->h); delete b; }
  void* giGetRenderBufferMem(GiRenderBuffer* b) { return b ? giCGetRenderBufferMem(b->h) : nullptr; }
}
