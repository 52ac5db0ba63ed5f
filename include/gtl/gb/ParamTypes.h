// gtl/gb/ParamTypes.h -- the value types of material parameters that cross the gi boundary.
//
// hdGatling includes this header through <gtl/gi/Gi.h> and builds a GiMaterialParameters map out of these five PODs
// (reference: /root/reference/src/gb/gtl/gb/ParamTypes.h:22-28; call sites src/hdGatling/materialNetworkCompiler.cpp:548-601, where
// a texture asset is brace-initialised as { resolved path, isSrgb }).  Field names, order and types are part of the interface: a
// drop-in must spell them exactly so, otherwise the delegate does not compile.
#pragma once

#include <string>

namespace gtl
{
  struct GbVec2f { float x; float y; };
  struct GbVec3f { float x; float y; float z; };
  struct GbVec4f { float x; float y; float z; float w; };

  struct GbColor { float r; float g; float b; };
  struct GbTextureAsset { std::string absPath; bool isSrgb; };   // absolute (resolved) file path + "decode as sRGB"
}
