// Test double of <MaterialXCore/Document.h>: just enough of the MaterialX API surface for gtl_shim_mtlx.cpp to compile and run
// without a MaterialX install (MaterialX is an OpenUSD dependency, absent from this image).  The "document" carries its XML text.
#pragma once
#include <memory>
#include <string>
namespace MaterialX
{
  class Document { public: std::string xml; };
  using DocumentPtr = std::shared_ptr<Document>;
  inline DocumentPtr createDocument() { return std::make_shared<Document>(); }
}
