// png_writer.cpp — the PNG hand-off after the path (ImageWriter::writePNG, image_writer.cpp:6-28):
// the reference's quantiser and a store-only PNG encoder.  Host-only, no HIP: also built under
// ASan/UBSan by tests/test_host_sanitizers.py.
#include "mcrt.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

__attribute__((visibility("hidden"))) int mcrt_detail_fail(int code, const char* msg);  // api.cpp: sets mcrt_last_error()

namespace {
int fail(int code, const std::string& msg) { return mcrt_detail_fail(code, msg.c_str()); }
}  // namespace

extern "C" {

void mcrt_quantize_rgba8(const float* rgba, uint8_t* out, size_t n_pixels) {  // image_writer.cpp:18-22
    for (size_t i = 0; i < n_pixels * 4; ++i) {
        float v = rgba[i];
        v = (v < 0.0f) ? 0.0f : ((1.0f < v) ? 1.0f : v);
        out[i] = static_cast<uint8_t>(v * 255.0f + 0.5f);
    }
}


namespace {
struct Crc32Table {
    uint32_t t[8][256];
    Crc32Table() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int k = 1; k < 8; ++k) t[k][i] = t[0][t[k - 1][i] & 0xffu] ^ (t[k - 1][i] >> 8);
    }
};
// slicing-by-8 CRC-32 (IEEE 802.3, as PNG chunks use it); state in/out without the final xor
uint32_t crc32_update(uint32_t c, const uint8_t* p, size_t n) {
    static const Crc32Table tab;
    while (n >= 8) {
        uint32_t a, b;
        std::memcpy(&a, p, 4);
        std::memcpy(&b, p + 4, 4);
        a ^= c;
        c = tab.t[7][a & 0xffu] ^ tab.t[6][(a >> 8) & 0xffu] ^ tab.t[5][(a >> 16) & 0xffu] ^ tab.t[4][a >> 24] ^
            tab.t[3][b & 0xffu] ^ tab.t[2][(b >> 8) & 0xffu] ^ tab.t[1][(b >> 16) & 0xffu] ^ tab.t[0][b >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = tab.t[0][(c ^ *p++) & 0xffu] ^ (c >> 8);
    return c;
}
// Adler-32 over `n` bytes continuing from (a, b)
void adler32_update(uint32_t& a, uint32_t& b, const uint8_t* p, size_t n) {
    while (n) {
        size_t k = n < 5552 ? n : 5552;  // largest run before the 32-bit sums need a modulo
        n -= k;
        while (k--) {
            a += *p++;
            b += a;
        }
        a %= 65521u;
        b %= 65521u;
    }
}
void put_be32(uint8_t* p, uint32_t v) {
    p[0] = static_cast<uint8_t>(v >> 24), p[1] = static_cast<uint8_t>(v >> 16), p[2] = static_cast<uint8_t>(v >> 8), p[3] = static_cast<uint8_t>(v);
}
size_t png_size(size_t raw) {  // raw = height * (1 + 4 * width) filtered bytes
    const size_t blocks = raw ? (raw + 65534) / 65535 : 1;
    return 8 + 25 + (12 + 2 + raw + 5 * blocks + 4) + 12;
}
}  // namespace

size_t mcrt_encode_png_rgba8(const uint8_t* rgba, int width, int height, uint8_t* out, size_t capacity) {
    if (!rgba || width <= 0 || height <= 0) return 0;
    const size_t stride = static_cast<size_t>(width) * 4, line = stride + 1, raw = line * static_cast<size_t>(height);
    const size_t total = png_size(raw);
    if (raw + 5 * ((raw + 65534) / 65535) + 6 > 0x7fffffffull) return 0;  // one IDAT chunk holds < 2^31 bytes
    if (!out || capacity < total) return total;
    uint8_t* p = out;
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::memcpy(p, sig, 8);
    p += 8;
    // IHDR
    put_be32(p, 13);
    std::memcpy(p + 4, "IHDR", 4);
    put_be32(p + 8, static_cast<uint32_t>(width));
    put_be32(p + 12, static_cast<uint32_t>(height));
    p[16] = 8, p[17] = 6, p[18] = 0, p[19] = 0, p[20] = 0;  // 8-bit RGBA, deflate, adaptive filtering, no interlace
    put_be32(p + 21, crc32_update(0xffffffffu, p + 4, 17) ^ 0xffffffffu);
    p += 25;
    // IDAT: zlib header, stored deflate blocks over [filter byte 0 + scanline] x height, Adler-32
    const size_t blocks = (raw + 65534) / 65535;
    const size_t idat_len = 2 + raw + 5 * blocks + 4;
    put_be32(p, static_cast<uint32_t>(idat_len));
    std::memcpy(p + 4, "IDAT", 4);
    uint8_t* z = p + 8;
    z[0] = 0x78, z[1] = 0x01;
    uint8_t* q = z + 2;
    uint32_t ad_a = 1, ad_b = 0;
    // the filtered stream is produced straight into the stored blocks
    size_t in_block = 0, left = raw;
    auto emit = [&](const uint8_t* src, size_t n) {
        while (n) {
            if (in_block == 0) {
                const size_t len = left < 65535 ? left : 65535;
                q[0] = (left <= 65535) ? 1 : 0;  // BFINAL on the last block, BTYPE = 00 (stored)
                q[1] = static_cast<uint8_t>(len), q[2] = static_cast<uint8_t>(len >> 8);
                q[3] = static_cast<uint8_t>(~len), q[4] = static_cast<uint8_t>((~len) >> 8);
                q += 5;
                in_block = len;
            }
            const size_t k = n < in_block ? n : in_block;
            std::memcpy(q, src, k);
            adler32_update(ad_a, ad_b, src, k);
            q += k, src += k, n -= k, in_block -= k, left -= k;
        }
    };
    const uint8_t zero = 0;
    for (int y = 0; y < height; ++y) {
        emit(&zero, 1);  // filter type 0 (None)
        emit(rgba + static_cast<size_t>(y) * stride, stride);
    }
    put_be32(q, (ad_b << 16) | ad_a);
    q += 4;
    put_be32(q, crc32_update(0xffffffffu, p + 4, 4 + idat_len) ^ 0xffffffffu);
    p = q + 4;
    // IEND
    put_be32(p, 0);
    std::memcpy(p + 4, "IEND", 4);
    put_be32(p + 8, crc32_update(0xffffffffu, p + 4, 4) ^ 0xffffffffu);
    p += 12;
    return static_cast<size_t>(p - out);
}

int mcrt_write_png_rgba8(const char* path, const uint8_t* rgba, int width, int height) {
    if (!path || !rgba || width <= 0 || height <= 0) return fail(MCRT_ERR_INVALID, "bad argument");  // image_writer.cpp:7-9
    const size_t need = mcrt_encode_png_rgba8(rgba, width, height, nullptr, 0);
    if (!need) return fail(MCRT_ERR_INVALID, "image too large for one IDAT chunk");
    std::vector<uint8_t> buf(need);
    if (mcrt_encode_png_rgba8(rgba, width, height, buf.data(), buf.size()) != need) return fail(MCRT_ERR_INVALID, "PNG encoding failed");
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(MCRT_ERR_INVALID, std::string("cannot create ") + path);
    const size_t wr = std::fwrite(buf.data(), 1, buf.size(), f);
    const int cl = std::fclose(f);
    if (wr != buf.size() || cl != 0) return fail(MCRT_ERR_INVALID, std::string("short write to ") + path);
    return MCRT_OK;
}

int mcrt_write_png_f32(const char* path, const float* rgba, int width, int height) {
    if (!path || !rgba || width <= 0 || height <= 0) return fail(MCRT_ERR_INVALID, "bad argument");
    std::vector<uint8_t> q(static_cast<size_t>(width) * height * 4);
    mcrt_quantize_rgba8(rgba, q.data(), static_cast<size_t>(width) * height);
    return mcrt_write_png_rgba8(path, q.data(), width, height);
}

}  // extern "C"
