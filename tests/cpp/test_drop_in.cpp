// test_drop_in.cpp — the reference's TileRenderer tests (tests/test_tile_renderer.cpp:9-159) restated
// against the drop-in TileRenderer (tile_renderer_hip.cpp).  Compiles against the reference's own
// headers (-DMCRT_USE_REFERENCE_HEADERS, dev container) or the mirror types.  `--gpu` also runs
// the cases that render; without it only the host-side behaviour is exercised.
#ifdef MCRT_USE_REFERENCE_HEADERS
#include "raytracer/tile_renderer.h"
#include "output/image_writer.h"
#else
#include "mcskin_types.hpp"
#endif

#include <atomic>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>

static int failures = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond);   \
            ++failures;                                                    \
        }                                                                  \
    } while (0)

static Scene makeSimpleScene() {
    Scene scene;
    scene.backgroundColor = Color(0.1f, 0.1f, 0.1f);
    scene.light.position = Vec3(0, 50, 50);
    scene.light.color = Color(1, 1, 1);
    scene.camera.position = Vec3(0, 18, 40);
    scene.camera.target = Vec3(0, 18, 0);
    scene.camera.up = Vec3(0, 1, 0);
    scene.camera.fov = 60.0f;
    return scene;
}

// a red 8x8x8 box at the camera target, built the way MeshBuilder::buildBox orders its faces
static void addBox(Scene& scene) {
    Mesh m;
    m.ownedTextures[0] = TextureRegion(1, 1, {Color(1, 0, 0, 1)});
    for (int i = 1; i < 6; ++i) m.ownedTextures[i] = m.ownedTextures[0];
    const float lo[3] = {-4, 14, -4}, hi[3] = {4, 22, 4};
    auto P = [&](int c) { return Vec3((c & 1) ? hi[0] : lo[0], (c & 2) ? hi[1] : lo[1], (c & 4) ? hi[2] : lo[2]); };
    const int quad[6][4] = {{2, 3, 1, 0}, {7, 6, 4, 5}, {3, 7, 5, 1}, {6, 2, 0, 4}, {6, 7, 3, 2}, {0, 1, 5, 4}};
    const int texOf[6] = {1, 0, 2, 3, 4, 5};
    for (int f = 0; f < 6; ++f) {
        const int* q = quad[f];
        const int tri[2][3] = {{q[0], q[1], q[2]}, {q[0], q[2], q[3]}};
        for (auto& t : tri) {
            Triangle T;
            T.v0 = P(t[0]), T.v1 = P(t[1]), T.v2 = P(t[2]);
            T.texture = &m.ownedTextures[texOf[f]];
            m.triangles.push_back(T);
        }
    }
    scene.meshes.push_back(m);  // copy: exercises the texture re-pointing
}

int main(int argc, char** argv) {
    const bool gpu = argc > 1 && std::strcmp(argv[1], "--gpu") == 0;

    // generateTiles — test_tile_renderer.cpp:9-57
    {
        auto t = TileRenderer::generateTiles(64, 64, 32);
        CHECK(t.size() == 4u);
        CHECK(t[0].x == 0 && t[0].y == 0 && t[0].width == 32 && t[0].height == 32);
        CHECK(t[1].x == 32 && t[1].y == 0 && t[2].x == 0 && t[2].y == 32 && t[3].x == 32 && t[3].y == 32);
        t = TileRenderer::generateTiles(50, 30, 32);
        CHECK(t.size() == 2u && t[0].width == 32 && t[0].height == 30 && t[1].x == 32 && t[1].width == 18 && t[1].height == 30);
        t = TileRenderer::generateTiles(10, 10, 32);
        CHECK(t.size() == 1u && t[0].width == 10 && t[0].height == 10);
        CHECK(TileRenderer::generateTiles(0, 64, 32).empty() && TileRenderer::generateTiles(64, 0, 32).empty());
        CHECK(TileRenderer::generateTiles(64, 64, 0).empty() && TileRenderer::generateTiles(-1, 64, 32).empty());
    }
    // invalid size → untouched image, no error (tile_renderer.cpp:144-146)
    {
        RayTracer::Config c;
        c.width = 8, c.height = 8, c.tileSize = 0;
        Image img = TileRenderer::render(makeSimpleScene(), c);
        CHECK(img.width == 8 && img.pixels.size() == 64u && TileRenderer::lastErrors().empty());
        CHECK(img.pixels[5] == Color());
    }
    if (!gpu) {
        // without a device the failure is recorded, not thrown, and pixels stay Color()
        RayTracer::Config c;
        c.width = 16, c.height = 16;
        Image img = TileRenderer::render(makeSimpleScene(), c);
        CHECK(img.pixels.size() == 256u);
        if (!TileRenderer::lastErrors().empty()) {
            CHECK(TileRenderer::lastErrors()[0].tileIndex == -1);
            CHECK(img.pixels[0] == Color());
            std::printf("no device: \"%s\"\n", TileRenderer::lastErrors()[0].message.c_str());
        }
    } else {
        Scene scene = makeSimpleScene();
        RayTracer::Config c;
        c.width = 32, c.height = 32, c.maxBounces = 0, c.tileSize = 16, c.threadCount = 2;
        Image img = TileRenderer::render(scene, c);  // RenderProducesCorrectSize
        CHECK(img.width == 32 && img.height == 32 && img.pixels.size() == 32u * 32u && TileRenderer::lastErrors().empty());
        std::atomic<int> calls{0};
        int lastTotal = 0, lastDone = 0;
        c.threadCount = 1;
        TileRenderer::render(scene, c, [&](int done, int total) { calls++; lastTotal = total; lastDone = done; });  // ProgressCallbackInvoked
        CHECK(calls.load() == 4 && lastTotal == 4 && lastDone == 4);
        c.width = 16, c.height = 16, c.tileSize = 8, c.threadCount = 0;  // DefaultThreadCount...
        CHECK(TileRenderer::render(scene, c).width == 16);
        // SingleThreadMatchesMultiThread, on a scene with a box so that rays hit something
        addBox(scene);
        c.maxBounces = 1, c.threadCount = 1;
        Image a = TileRenderer::render(scene, c);
        c.threadCount = 4;
        Image b = TileRenderer::render(scene, c);
        CHECK(a.pixels.size() == b.pixels.size());
        bool same = true, sawBox = false;
        for (size_t i = 0; i < a.pixels.size(); ++i) {
            same = same && a.pixels[i] == b.pixels[i];
            sawBox = sawBox || (a.pixels[i].r > 0.3f && a.pixels[i].g < 0.2f);
        }
        CHECK(same && sawBox);
        c.width = 8, c.height = 8, c.tileSize = 8;
        CHECK(TileRenderer::render(scene, c, nullptr).width == 8);  // NullProgressCallbackIsOk
        // renderTile equals the same tile of render()
        c.width = 32, c.height = 24, c.tileSize = 16, c.samplesPerPixel = 2;
        Image full = TileRenderer::render(scene, c);
        Image part(c.width, c.height);
        auto tiles = TileRenderer::generateTiles(c.width, c.height, c.tileSize);
        TileRenderer::renderTile(tiles[3], scene, c, part);
        bool tileSame = true, restUntouched = true;
        for (int y = 0; y < c.height; ++y)
            for (int x = 0; x < c.width; ++x) {
                bool in = x >= tiles[3].x && x < tiles[3].x + tiles[3].width && y >= tiles[3].y && y < tiles[3].y + tiles[3].height;
                const Color& p = part.pixels[y * c.width + x];
                if (in) tileSame = tileSame && p == full.pixels[y * c.width + x];
                else restUntouched = restUntouched && p == Color();
            }
        CHECK(tileSame && restUntouched && TileRenderer::lastErrors().empty());
        // renderTile takes ANY Tile of the frame (tile_renderer.cpp:71-127), on or off the tile grid: twice the same pixels,
        // nothing outside the rectangle, no error; a rectangle that leaves the frame is refused (the reference would write
        // out of bounds)
        const Tile odd{5, 3, 17, 9};
        Image a1(c.width, c.height), a2(c.width, c.height);
        TileRenderer::renderTile(odd, scene, c, a1);
        TileRenderer::renderTile(odd, scene, c, a2);
        bool same2 = true, outside = true, touched = false;
        for (int y = 0; y < c.height; ++y)
            for (int x = 0; x < c.width; ++x) {
                const bool in = x >= odd.x && x < odd.x + odd.width && y >= odd.y && y < odd.y + odd.height;
                const Color& p1 = a1.pixels[y * c.width + x];
                same2 = same2 && p1 == a2.pixels[y * c.width + x];
                if (in) touched = touched || !(p1 == Color());
                else outside = outside && p1 == Color();
            }
        CHECK(same2 && outside && touched && TileRenderer::lastErrors().empty());
        TileRenderer::renderTile(Tile{30, 20, 8, 8}, scene, c, a1);
        CHECK(!TileRenderer::lastErrors().empty());
    }
    // ImageWriter::writePNG — test_image_writer.cpp: empty image and bad path → false; a valid image
    // → a PNG file that starts with the PNG signature and carries the IHDR dimensions
    {
        CHECK(!ImageWriter::writePNG(Image(), "/tmp/mcrt_drop_in_empty.png"));
        Image img(5, 3);
        for (size_t i = 0; i < img.pixels.size(); ++i) img.pixels[i] = Color(0.1f * static_cast<float>(i % 7), 1.5f, -0.25f, 1.0f);
        CHECK(!ImageWriter::writePNG(img, "/nonexistent_dir_mcrt/x.png"));
        const std::string path = "/tmp/mcrt_drop_in_" + std::to_string(static_cast<long long>(std::hash<std::string>{}(__FILE__) % 100000)) + ".png";
        CHECK(ImageWriter::writePNG(img, path));
        FILE* f = std::fopen(path.c_str(), "rb");
        CHECK(f != nullptr);
        if (f) {
            unsigned char head[24] = {0};
            CHECK(std::fread(head, 1, 24, f) == 24u);
            std::fclose(f);
            std::remove(path.c_str());
            const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
            CHECK(std::memcmp(head, sig, 8) == 0 && std::memcmp(head + 12, "IHDR", 4) == 0);
            CHECK(head[19] == 5 && head[23] == 3);
        }
    }
    std::printf("%s: %d failure(s)\n", gpu ? "gpu" : "host", failures);
    return failures ? 1 : 0;
}
