// flatten.h — host-side scene flattener (see flatten.cpp)
#ifndef MCRT_FLATTEN_H
#define MCRT_FLATTEN_H

#include "flat_scene.h"
#include "mcrt.h"

#include <cstdint>
#include <string>
#include <vector>

namespace mcrt {
// Returns false and fills `err` for malformed descriptions.
bool flatten_scene(const mcrt_scene_desc* desc, std::vector<uint8_t>& blob, std::string& err);
}  // namespace mcrt

#endif
