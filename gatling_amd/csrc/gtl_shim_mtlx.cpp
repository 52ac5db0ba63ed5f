// gtl_shim_mtlx.cpp -- the one translation unit of the boundary that needs the MaterialX headers.
//
// hdGatling hands every UsdPreviewSurface / MaterialX material over as a MaterialX::DocumentPtr behind a shared_ptr<void>
// (giCreateMaterialFromMtlxDoc, /root/reference/src/gi/gtl/gi/Gi.h:205; call site src/hdGatling/materialNetworkCompiler.cpp:685).  The
// reference feeds that document to its MaterialX -> MDL -> GLSL code generator (src/gi/impl/Gi.cpp:2545-2556, src/mc/impl/Frontend.cpp);
// here it is serialised to XML text and read by the same scanner as giCreateMaterialFromMtlxStr (gtl_shim.cpp): constant inputs of the
// UsdPreviewSurface / open_pbr_surface node, primvar readers and image nodes upstream of them.
//
// Built into libgatling_gi.so when <MaterialXFormat/XmlIo.h> is found (gatling_amd/build.py: $MATERIALX_ROOT/include or the system
// include path); hdGatling already links MaterialXCore / MaterialXFormat, so no new dependency reaches the delegate.
#include <MaterialXCore/Document.h>
#include <MaterialXFormat/XmlIo.h>

#include <gtl/gi/Gi.h>

namespace
{
  std::string docToXml(const std::shared_ptr<void>& doc)
  {
    MaterialX::DocumentPtr d = std::static_pointer_cast<MaterialX::Document>(doc);
    return d ? MaterialX::writeToXmlString(d) : std::string();
  }
  struct Registrar { Registrar() { gtl::gtlRegisterMtlxDocSerializer(&docToXml); } } s_registrar;
}
