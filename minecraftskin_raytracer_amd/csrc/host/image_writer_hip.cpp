// image_writer_hip.cpp — drop-in replacement for the reference's src/output/image_writer.cpp: the
// same `ImageWriter::writePNG(const Image&, const std::string&)` (output/image_writer.h:6-12).
//
// Behaviour kept (image_writer.cpp:6-28): false for an empty image (width/height <= 0 or no
// pixels), float RGBA quantised as (uint8_t)(clamp(c,0,1)*255.0f+0.5f) per channel incl. alpha,
// 8-bit RGBA PNG, false when the file cannot be written.  The encoder is the library's store-only
// writer (mcrt_write_png_rgba8): the file is larger than stbi_write_png's, it decodes to the same
// pixels, and writing it is bounded by memory bandwidth rather than by zlib.
#ifdef MCRT_USE_REFERENCE_HEADERS
#include "output/image_writer.h"
#else
#include "mcskin_types.hpp"
#endif

#include "mcrt.h"

#include <vector>

bool ImageWriter::writePNG(const Image& image, const std::string& path) {
    if (image.width <= 0 || image.height <= 0 || image.pixels.empty()) return false;
    if (image.pixels.size() < static_cast<size_t>(image.width) * static_cast<size_t>(image.height)) return false;
    static_assert(sizeof(Color) == 4 * sizeof(float), "Image::pixels must be a dense float4 array");
    return mcrt_write_png_f32(path.c_str(), reinterpret_cast<const float*>(image.pixels.data()), image.width,
                              image.height) == MCRT_OK;
}
