// render_check — drives racc::render (include/RayAccelerator.h) the way the reference's example app does
// (Renderer/main.cpp:187-209,58-66; TiledRenderer.cpp:55-67): loads a reference-format scene file
// (Renderer/main.cpp:117-191), spawns 128x128 primary-ray tiles, shades with one diffuse-ish bounce per hit up to
// maxDepth, and dumps EVERY traced ray with its Result so the Python side can re-trace them with the oracle.
//   render_check <scene.bin> <out.bin> <width> <height> <maxDepth> [frames]
//   render_check <scene.bin> --null-callbacks <width> <height> <repeat> [frames]
//       the scheduler alone (round-4 verdict, item 4): spawn copies pre-generated 128x128 tiles of primaries into the stream (one memcpy
//       per tile, the tile set `repeat` times over), shade consumes nothing and emits nothing; prints racc::render's own rate — what the
//       ray-stream state machine + the host RayStream path sustain when the callbacks cost nothing (RayAccelerator.cpp:48-156,738-759).
// Output records: {u32 pixel, u32 depth, Ray (32 B), Result (16 B)} = 56 B each, preceded by {u64 count, u64 raysTraced}.
#include "RayAccelerator.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

#pragma pack(push, 1)
struct SceneHeader {   // Renderer/main.cpp:118-133
    uint32_t maxDepth, vertexCount, triangleCount;
    uint16_t viewportWidth, viewportHeight, environmentWidth, environmentHeight;
    float origin[3], target[3], up[3], fov;
};
struct Record { uint32_t pixel, depth; float ray[8]; uint32_t triangle; float t, u, v; };
#pragma pack(pop)
static_assert(sizeof(SceneHeader) == 60 && sizeof(Record) == 56, "layout");

struct Payload { uint32_t pixel, depth; };

struct App {
    std::vector<racc::Vertex> vertices;
    std::vector<uint32_t> indices;
    float camOrigin[3], camRight[3], camUp[3], camView[3];
    unsigned width, height, tilesX, tilesY, maxDepth;
    std::atomic<unsigned> nextTile{0};
    racc::ContextInfo info;
    std::vector<Payload> payload;                       // [stream][slot]
    std::vector<std::vector<Record>> perThread;
    // --null-callbacks
    std::vector<racc::Ray> tileRays;                    // [tile][128 * 128], generated before the clock starts
    unsigned repeat = 1;
    std::atomic<unsigned long long> consumed{0};
};

void cross(const float* a, const float* b, float* r) { r[0] = a[1]*b[2]-a[2]*b[1]; r[1] = a[2]*b[0]-a[0]*b[2]; r[2] = a[0]*b[1]-a[1]*b[0]; }
void normalize(float* v) { const float l = std::sqrt(v[0]*v[0]+v[1]*v[1]+v[2]*v[2]); v[0]/=l; v[1]/=l; v[2]/=l; }
uint32_t hash(uint32_t x) { x = x * 747796405u + 2891336453u; uint32_t w = ((x >> ((x >> 28) + 4)) ^ x) * 277803737u; return (w >> 22) ^ w; }

bool spawn(void* data, unsigned, racc::RayStream* out) {   // TiledRenderer.cpp:55-67 + Camera.cpp:55-85 (pixel centres)
    App* app = static_cast<App*>(data);
    const unsigned tile = app->nextTile++;
    if (tile >= app->tilesX * app->tilesY) return false;
    const unsigned tx = (tile % app->tilesX) * 128, ty = (tile / app->tilesX) * 128;
    Payload* pl = app->payload.data() + size_t(out->index) * app->info.rayStreamSize;
    for (unsigned y = 0; y < 128; ++y)
        for (unsigned x = 0; x < 128; ++x) {
            const float px = float(tx + x) + 0.5f, py = float(ty + y) + 0.5f;
            float d[3];
            for (int k = 0; k < 3; ++k) d[k] = app->camView[k] + app->camRight[k] * px + app->camUp[k] * py;
            normalize(d);
            racc::Ray& r = out->rays[out->count];
            for (int k = 0; k < 3; ++k) { r.origin[k] = app->camOrigin[k]; r.dir[k] = d[k]; }
            r.minT = 0.0f; r.maxT = 1e6f;
            pl[out->count] = Payload{(ty + y) * app->width + tx + x, 0};
            ++out->count;
        }
    return tile != app->tilesX * app->tilesY - 1;
}

bool spawnCopy(void* data, unsigned, racc::RayStream* out) {      // a tile of pre-generated primaries: one memcpy
    App* app = static_cast<App*>(data);
    const unsigned tiles = app->tilesX * app->tilesY, total = tiles * app->repeat;
    const unsigned k = app->nextTile++;
    if (k >= total) return false;
    std::memcpy(out->rays + out->count, app->tileRays.data() + size_t(k % tiles) * 16384, 16384 * sizeof(racc::Ray));
    out->count += 16384;
    return k != total - 1;
}

void shadeNothing(void* data, unsigned, const racc::RayStream*, unsigned start, unsigned end, racc::RayStream*) {
    static_cast<App*>(data)->consumed += end - start;
}

void shade(void* data, unsigned thread, const racc::RayStream* in, unsigned start, unsigned end, racc::RayStream* out) {
    App* app = static_cast<App*>(data);
    const Payload* pin = app->payload.data() + size_t(in->index) * app->info.rayStreamSize;
    Payload* pout = app->payload.data() + size_t(out->index) * app->info.rayStreamSize;
    std::vector<Record>& log = app->perThread[thread];
    for (unsigned i = start; i < end; ++i) {
        const racc::Ray& r = in->rays[i];
        const racc::Result& h = in->results[i];
        Record rec;
        rec.pixel = pin[i].pixel; rec.depth = pin[i].depth;
        std::memcpy(rec.ray, &r, 32);
        rec.triangle = h.triangle; rec.t = h.hit.t; rec.u = h.hit.u; rec.v = h.hit.v;
        log.push_back(rec);
        if (h.triangle == racc::invalidTriangle || pin[i].depth + 1 >= app->maxDepth) continue;
        // PathTracingRenderer.cpp:410-422: o = P + 1e-4*Ng (toward the incoming side), minT = 1e-3, maxT = 1e6
        const uint32_t* tri = &app->indices[size_t(h.triangle) * 3];
        const racc::Vertex &a = app->vertices[tri[0]], &b = app->vertices[tri[1]], &c = app->vertices[tri[2]];
        const float e1[3] = {b.x-a.x, b.y-a.y, b.z-a.z}, e2[3] = {c.x-a.x, c.y-a.y, c.z-a.z};
        float n[3]; cross(e1, e2, n); normalize(n);
        if (n[0]*r.dir[0] + n[1]*r.dir[1] + n[2]*r.dir[2] > 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
        const uint32_t h1 = hash(rec.pixel * 16u + rec.depth), h2 = hash(h1);
        const float r1 = float(h1 >> 8) * (6.2831853f / 16777216.0f), r2 = float(h2 >> 8) * (1.0f / 16777216.0f);
        float helper[3] = {std::fabs(n[0]) > 0.9f ? 0.f : 1.f, std::fabs(n[0]) > 0.9f ? 1.f : 0.f, 0.f}, bu[3], bv[3];
        cross(helper, n, bu); normalize(bu); cross(n, bu, bv);
        const float s = std::sqrt(r2), cz = std::sqrt(1.0f - r2);
        float d[3];
        for (int k = 0; k < 3; ++k) d[k] = n[k] * cz + (bu[k] * std::cos(r1) + bv[k] * std::sin(r1)) * s;
        normalize(d);
        racc::Ray& o = out->rays[out->count];
        for (int k = 0; k < 3; ++k) { o.origin[k] = r.origin[k] + r.dir[k] * h.hit.t + 1e-4f * n[k]; o.dir[k] = d[k]; }
        o.minT = 1e-3f; o.maxT = 1e6f;
        pout[out->count] = Payload{rec.pixel, rec.depth + 1};
        ++out->count;
    }
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 6) { std::fprintf(stderr, "usage: render_check scene.bin out.bin width height maxDepth [frames]\n"); return 2; }
    App app;
    const bool nullCallbacks = std::strcmp(argv[2], "--null-callbacks") == 0;
    app.width = unsigned(std::atoi(argv[3])); app.height = unsigned(std::atoi(argv[4])); app.maxDepth = unsigned(std::atoi(argv[5]));
    if (nullCallbacks) { app.repeat = app.maxDepth ? app.maxDepth : 1; app.maxDepth = 1; }
    const int frames = argc > 6 ? std::atoi(argv[6]) : 1;
    app.tilesX = app.width / 128; app.tilesY = app.height / 128;   // TiledRenderer.cpp:20-22

    FILE* f = std::fopen(argv[1], "rb");
    SceneHeader hdr;
    if (!f || std::fread(&hdr, sizeof(hdr), 1, f) != 1) { std::fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    app.indices.resize(size_t(hdr.triangleCount) * 3);
    app.vertices.resize(hdr.vertexCount);
    std::vector<racc::Color> env(size_t(hdr.environmentWidth) * hdr.environmentHeight);
    bool ok = std::fread(app.indices.data(), 12, hdr.triangleCount, f) == hdr.triangleCount;
    ok = ok && std::fseek(f, long(hdr.triangleCount) * (2 + 16), SEEK_CUR) == 0;                     // materials, triangle normals
    ok = ok && std::fread(app.vertices.data(), 16, hdr.vertexCount, f) == hdr.vertexCount;
    ok = ok && std::fseek(f, long(hdr.vertexCount) * (16 + 8), SEEK_CUR) == 0;                       // normals, texcoords
    ok = ok && std::fread(env.data(), 16, env.size(), f) == env.size();
    std::fclose(f);
    if (!ok) { std::fprintf(stderr, "short scene file\n"); return 2; }

    {   // Camera::lookAt, Renderer/Camera.cpp:13-26 (header.dir is the look-at TARGET, main.cpp:151)
        float fwd[3] = {hdr.target[0]-hdr.origin[0], hdr.target[1]-hdr.origin[1], hdr.target[2]-hdr.origin[2]}, right[3], up[3];
        normalize(fwd); cross(fwd, hdr.up, right); normalize(right); cross(right, fwd, up);
        const float ey = std::tan(0.5f * hdr.fov * 3.14159265f / 180.0f), ex = ey * float(app.width) / float(app.height);
        for (int k = 0; k < 3; ++k) {
            app.camOrigin[k] = hdr.origin[k];
            app.camRight[k] = right[k] * (-2.0f / float(app.width) * ex);
            app.camUp[k] = up[k] * (-2.0f / float(app.height) * ey);
            app.camView[k] = fwd[k] + right[k] * ex + up[k] * ey;
        }
    }

    racc::init();
    // RACC_DEVICES=0,0 (or 0,1,...): one racc::Context over several engine contexts, scene replicated, streams sharded
    racc::GpuContext gpu = nullptr;
    if (const char* d = std::getenv("RACC_DEVICES")) {
        int ordinals[RACC_MAX_DEVICES];
        unsigned n = 0;
        for (const char* p = d; *p && n < RACC_MAX_DEVICES;) {
            ordinals[n++] = std::atoi(p);
            while (*p && *p != ',') ++p;
            if (*p == ',') ++p;
        }
        gpu = racc::gpuContextForDevices(ordinals, n);
    } else {
        gpu = racc::gpuContextForDevice(0);
    }
    racc::Configuration cfg = racc::defaultConfiguration(gpu);
    if (const char* t = std::getenv("RACC_CPU_THREADS")) cfg.cpuThreads = unsigned(std::atoi(t));
    if (const char* t = std::getenv("RACC_BATCH")) cfg.rayStreamBatchSize = unsigned(std::atoi(t));
    if (const char* t = std::getenv("RACC_IN_FLIGHT")) cfg.maxRaysInFlight = unsigned(std::atoi(t));
    if (const char* t = std::getenv("RACC_GPU_THREADS")) cfg.gpuSubmissionThreads = unsigned(std::atoi(t));
    if (const char* t = std::getenv("RACC_SHADE_BATCH")) cfg.cpuShadeBatch = unsigned(std::atoi(t));
    racc::Context* ctx = racc::createContext(cfg);
    if (!ctx) return 3;
    app.info = racc::info(ctx);
    app.payload.resize(size_t(app.info.rayStreamCount) * app.info.rayStreamSize);
    app.perThread.resize(app.info.threadCount);
    racc::Scene* scene = racc::createScene(ctx, app.vertices.data(), hdr.vertexCount, app.indices.data(), hdr.triangleCount * 3);
    racc::Environment* environment = racc::createEnvironment(ctx, env.data(), hdr.environmentWidth, hdr.environmentHeight);
    if (!scene || !environment) return 3;

    if (nullCallbacks) {
        // pre-generate every tile's primaries through the ordinary spawn into a scratch stream, outside the clock
        const unsigned tiles = app.tilesX * app.tilesY;
        app.tileRays.resize(size_t(tiles) * 16384);
        std::vector<Payload> scratchPayload(16384);
        app.payload.assign(size_t(app.info.rayStreamCount) * app.info.rayStreamSize, Payload{0, 0});
        for (unsigned t = 0; t < tiles; ++t) {
            racc::RayStream tmp{0, 0, app.tileRays.data() + size_t(t) * 16384, nullptr};
            spawn(&app, 0, &tmp);
        }
        racc::RenderCallbacks ncb = { &app, spawnCopy, shadeNothing };
        double best = 1e30, sum = 0;
        uint64_t tracedN = 0;
        for (int frame = 0; frame < frames; ++frame) {
            app.nextTile = 0; app.consumed = 0;
            const auto t0 = std::chrono::steady_clock::now();
            tracedN = racc::render(ctx, scene, environment, ncb).raysTraced;
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (const char* err = racc::lastError(ctx)) { std::fprintf(stderr, "render_check: %s\n", err); return 5; }
            if (frame > 0 || frames == 1) { best = dt < best ? dt : best; sum += dt; }
        }
        const int counted = frames > 1 ? frames - 1 : 1;
        std::printf("{\"raysTraced\": %llu, \"consumed\": %llu, \"seconds_best\": %.6f, \"seconds_mean\": %.6f, \"mrays_per_s_best\": %.1f, \"mrays_per_s_mean\": %.1f, "
                    "\"frames_timed\": %d, \"streams\": %u, \"streamSize\": %u, \"cpuThreads\": %u, \"gpuSubmissionThreads\": %u}\n",
                    (unsigned long long)tracedN, (unsigned long long)app.consumed.load(), best, sum / counted, double(tracedN) / best / 1e6, double(tracedN) / (sum / counted) / 1e6,
                    counted, app.info.rayStreamCount, app.info.rayStreamSize, cfg.cpuThreads, cfg.gpuSubmissionThreads);
        const bool okN = tracedN == uint64_t(tiles) * 16384 * app.repeat && app.consumed.load() == tracedN;
        racc::destroy(environment); racc::destroy(scene); racc::destroy(ctx); racc::deinit();
        return okN ? 0 : 4;
    }

    racc::RenderCallbacks cb = { &app, spawn, shade };
    uint64_t traced = 0;
    for (int frame = 0; frame < frames; ++frame) {
        app.nextTile = 0;
        for (auto& v : app.perThread) v.clear();
        traced = racc::render(ctx, scene, environment, cb).raysTraced;
        if (const char* err = racc::lastError(ctx)) { std::fprintf(stderr, "render_check: %s\n", err); return 5; }
    }
    uint64_t count = 0;
    for (auto& v : app.perThread) count += v.size();
    FILE* o = std::fopen(argv[2], "wb");
    std::fwrite(&count, 8, 1, o); std::fwrite(&traced, 8, 1, o);
    for (auto& v : app.perThread) std::fwrite(v.data(), sizeof(Record), v.size(), o);
    std::fclose(o);
    std::printf("{\"raysTraced\": %llu, \"shaded\": %llu, \"streams\": %u, \"streamSize\": %u, \"threads\": %u}\n",
                (unsigned long long)traced, (unsigned long long)count, app.info.rayStreamCount, app.info.rayStreamSize, app.info.threadCount);
    racc::destroy(environment); racc::destroy(scene); racc::destroy(ctx); racc::deinit();
    return count == traced ? 0 : 4;
}
