// Test double of <MaterialXFormat/XmlIo.h> (signature of MaterialX 1.38 / 1.39: writeToXmlString(DocumentPtr, const XmlWriteOptions* = nullptr)).
#pragma once
#include <MaterialXCore/Document.h>
namespace MaterialX
{
  class XmlWriteOptions;
  inline std::string writeToXmlString(DocumentPtr doc, const XmlWriteOptions* = nullptr) { return doc ? doc->xml : std::string(); }
}
