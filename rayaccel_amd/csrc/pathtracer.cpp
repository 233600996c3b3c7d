// pathtracer.cpp — path-tracing consumer of the intersect path, the counterpart of the reference's example renderer
// for BASELINE.json configs[4] (1920x1080 x 64 spp, multi-bounce ray streams).
//
// What it mirrors (file:line into the reference checkout):
//   scene file                  Renderer/main.cpp:117-191   (header, indices, materials, normals, vertices, env)
//   4 hard-coded materials      Renderer/main.cpp:165-168   ReflectiveDiffuseMaterial(kd, eta)
//   camera                      Renderer/Camera.cpp:13-26,55-85; tiles of 128x128, Renderer/TiledRenderer.cpp:55-67
//   payload                     Renderer/LightPath.h:14-17  weight[3] + pixel (low 24 bits) | depth (high 8)
//   shade()                     Renderer/PathTracingRenderer.cpp:72-566: interpolate the vertex normals at the hit,
//                               sample the material (Renderer/Materials.cpp:39-151: Fresnel-weighted choice between the
//                               mirror direction and a cosine-weighted diffuse direction), multiply the path weight,
//                               drop paths whose weight is <= 0.01 in every channel or whose new direction is on the
//                               wrong side of the geometric normal, offset the origin by 1e-4 * Ng, minT = 1e-3,
//                               maxT = 1e6 (:394-422); a miss adds weight * environment radiance to its pixel (:505-563).
// The per-ray arithmetic lives in pt_shade.h and is shared verbatim with the device-resident consumer (pt_device.hip).
// What differs, deliberately: scalar C++ instead of 8-wide AVX2; a counter-based RNG keyed by (pixel, sample, depth)
// instead of rand()-seeded streams (Camera.cpp:58, PathTracingRenderer.cpp:102), so a frame is reproducible; and a
// 64-bit fixed-point frame buffer, so the image does not depend on the order in which worker threads shade.
// All intersection work goes through racc::render (include/RayAccelerator.h), i.e. the MI355X engine.

#include "RayAccelerator.h"
#include "pt_scene.h"
#include "pt_shade.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

extern "C" {

typedef struct racc_pt_stats {
    uint64_t rays_traced;      // racc::Stats::raysTraced summed over the render calls
    uint64_t primary_rays;
    double seconds;            // wall time inside racc::render
    uint32_t tiles_x, tiles_y; // only floor(W/128) x floor(H/128) tiles are rendered (TiledRenderer.cpp:20-22)
    uint32_t max_depth;
    uint32_t threads;
    uint32_t triangles;
    uint32_t reserved;
} racc_pt_stats;

// Renders samples [spp_first, spp_first + spp_count) of every pixel and writes the SUM of their radiance (not the mean)
// as rgb doubles, row-major width*height*3.  Returns 0 on success.
int racc_pt_render_file(const char* scene_bin, int device, uint32_t width, uint32_t height,
                        uint32_t spp_first, uint32_t spp_count, uint32_t max_depth /* 0 = from the file */,
                        uint32_t cpu_threads /* 0 = default */, double* rgb_sum, racc_pt_stats* stats);
}

namespace {


using ptshade::Vec;
using ptshade::kFixed;
typedef ptshade::PathRec LightPath;   // LightPath.h:14-17
static_assert(sizeof(ptshade::RayRec) == sizeof(racc::Ray) && sizeof(ptshade::HitRec) == sizeof(racc::Result), "record layouts");

struct Renderer {
    const ptscene::Scene* scene = nullptr;
    ptshade::SceneView view{};
    std::vector<ptshade::ShadeTri> tris;   // one aligned 64 B shading record per triangle (the device consumer's layout): a hit
                                           // costs ONE cache miss instead of indices -> 3 vertices + 3 normals + material
    uint32_t width = 0, height = 0, tilesX = 0, tilesY = 0, maxDepth = 0;
    uint32_t sampleFirst = 0, sampleCount = 0;
    std::atomic<uint32_t> nextJob{0};      // job = sample * tiles + tile
    racc::ContextInfo info{};
    std::vector<LightPath> payload;        // [stream][slot], PathTracingRenderer.cpp:64,77
    std::vector<std::atomic<int64_t>> frame;   // rgb fixed point
    std::atomic<uint64_t> primaries{0};

    // Camera.cpp:55-85 with jittered samples from the counter RNG; TiledRenderer.cpp:55-67 for the tile walk.
    static bool spawn(void* data, unsigned, racc::RayStream* out) {
        Renderer* r = static_cast<Renderer*>(data);
        const uint32_t tiles = r->tilesX * r->tilesY, jobs = tiles * r->sampleCount;
        const uint32_t job = r->nextJob++;
        if (job >= jobs) return false;
        const uint32_t sample = r->sampleFirst + job / tiles, tile = job % tiles;
        const uint32_t tx = (tile % r->tilesX) * 128, ty = (tile / r->tilesX) * 128;
        LightPath* lp = r->payload.data() + size_t(out->index) * r->info.rayStreamSize;
#ifdef RACC_PT_SHADE8
        static const bool scalarOnly = [] { const char* e = std::getenv("RACC_PT_SCALAR"); return e && std::atoi(e) != 0; }();
        if (!scalarOnly) {
            for (uint32_t y = 0; y < 128; ++y)
                for (uint32_t x = 0; x < 128; x += 8) {      // eight pixels of a row at a time (lane k = primaryRay of pixel x + k, bit for bit)
                    ptshade::simd::primaryRay8(r->scene->cam, tx + x, ty + y, (ty + y) * r->width + tx + x, sample, reinterpret_cast<ptshade::RayRec*>(&out->rays[out->count]), &lp[out->count]);
                    for (uint32_t k = 0; k < 8; ++k) r->sampleOf(out->index, out->count + k) = sample;
                    out->count += 8;
                }
        } else
#endif
        for (uint32_t y = 0; y < 128; ++y)
            for (uint32_t x = 0; x < 128; ++x) {
                const uint32_t pixel = (ty + y) * r->width + tx + x;
                ptshade::primaryRay(r->scene->cam, tx + x, ty + y, pixel, sample, reinterpret_cast<ptshade::RayRec&>(out->rays[out->count]), lp[out->count]);
                r->sampleOf(out->index, out->count) = sample;
                ++out->count;
            }
        r->primaries += 128 * 128;
        return job != jobs - 1;
    }

    std::vector<uint32_t> sampleIndex;     // [stream][slot]: which sample a path belongs to (RNG key)
    uint32_t& sampleOf(uint32_t stream, uint32_t slot) { return sampleIndex[size_t(stream) * info.rayStreamSize + slot]; }

    static void shade(void* data, unsigned, const racc::RayStream* in, unsigned start, unsigned end, racc::RayStream* out) {
        Renderer* r = static_cast<Renderer*>(data);
        const LightPath* lin = r->payload.data() + size_t(in->index) * r->info.rayStreamSize;
        LightPath* lout = r->payload.data() + size_t(out->index) * r->info.rayStreamSize;
#ifdef RACC_PT_SHADE8
        // surface hits are collected eight at a time and shaded by ptshade::simd::shadeSurface8 (lane k = shadeSurface of hit k, bit for bit);
        // survivors go into the output stream in the order of their rays, as the scalar loop below would put them
        static const bool scalarOnly = [] { const char* e = std::getenv("RACC_PT_SCALAR"); return e && std::atoi(e) != 0; }();
        unsigned pending[8], nPending = 0;
        auto flush8 = [&] {
            const ptshade::RayRec* rays[8]; const ptshade::HitRec* hits[8]; const LightPath* paths[8]; const ptshade::ShadeTri* tris[8]; uint32_t samples[8];
            for (unsigned k = 0; k < 8; ++k) {
                const unsigned i = pending[k];
                rays[k] = reinterpret_cast<const ptshade::RayRec*>(&in->rays[i]); hits[k] = reinterpret_cast<const ptshade::HitRec*>(&in->results[i]);
                paths[k] = &lin[i]; samples[k] = r->sampleOf(in->index, i); tris[k] = &r->tris[hits[k]->triangle];
            }
            ptshade::RayRec nr[8]; LightPath np[8];
            unsigned alive = ptshade::simd::shadeSurface8(r->scene->mat, rays, hits, paths, samples, tris, nr, np);
            for (unsigned k = 0; alive; ++k, alive >>= 1)
                if (alive & 1u) {
                    reinterpret_cast<ptshade::RayRec&>(out->rays[out->count]) = nr[k];
                    lout[out->count] = np[k];
                    r->sampleOf(out->index, out->count) = samples[k];
                    ++out->count;
                }
            nPending = 0;
        };
#endif
        auto shadeOne = [&](unsigned i) {
            const ptshade::RayRec& ray = reinterpret_cast<const ptshade::RayRec&>(in->rays[i]);
            const ptshade::HitRec& hit = reinterpret_cast<const ptshade::HitRec&>(in->results[i]);
            const uint32_t sample = r->sampleOf(in->index, i);
            const ptshade::ShadeTri& st = r->tris[hit.triangle];
            if (!ptshade::shadeSurface(r->scene->mat, ray, hit, lin[i], sample, st.n0, st.n1, st.n2, ptshade::Vec{st.ng[0], st.ng[1], st.ng[2]}, st.material,
                                       reinterpret_cast<ptshade::RayRec&>(out->rays[out->count]), lout[out->count])) return;
            r->sampleOf(out->index, out->count) = sample;
            ++out->count;
        };
        for (unsigned i = start; i < end; ++i) {
            const ptshade::HitRec& hit = reinterpret_cast<const ptshade::HitRec&>(in->results[i]);
            const LightPath& lp = lin[i];
            if (hit.triangle == racc::invalidTriangle) {                    // PathTracingRenderer.cpp:505-563
                long long add[3]; bool valid[3];
                ptshade::missContribution(hit, lp, add, valid);
                const uint32_t pixel = lp.pixelDepth & 0xFFFFFFu;
                for (int ch = 0; ch < 3; ++ch)
                    if (valid[ch]) r->frame[size_t(pixel) * 3 + ch].fetch_add(int64_t(add[ch]), std::memory_order_relaxed);
                continue;
            }
            if ((lp.pixelDepth >> 24) >= r->maxDepth || hit.triangle >= r->view.triangleCount) continue;        // = shadeHit's guard (PathTracingRenderer.cpp:113-114)
            if (i + 16 < end) {               // the record of a hit some rays ahead is on its way while these are shaded
                const uint32_t ahead = reinterpret_cast<const ptshade::HitRec&>(in->results[i + 16]).triangle;
                if (ahead < r->view.triangleCount) __builtin_prefetch(&r->tris[ahead]);
            }
#ifdef RACC_PT_SHADE8
            if (!scalarOnly) {
                pending[nPending++] = i;
                if (nPending == 8) flush8();
                continue;
            }
#endif
            shadeOne(i);
        }
#ifdef RACC_PT_SHADE8
        for (unsigned k = 0; k < nPending; ++k) shadeOne(pending[k]);      // fewer than eight left in this slice: the scalar form (same results, same order)
#endif
    }
};

}  // namespace

// Test hooks into pt_shade.h (host build; the device build of the same header is checked against this one through the
// rendered images): n values of sin/cos(2*pi*r), and n uniforms of the counter RNG.
extern "C" void racc_pt_test_sincos2pi(const float* r, uint32_t n, float* s, float* c) {
    for (uint32_t i = 0; i < n; ++i) ptshade::sincos2pi(r[i], s[i], c[i]);
}
// Test hook: ptshade::sampleMaterial on n samples (kd[3] + eta per sample-independent material; rnd/normal/wo: 3 floats per sample).
extern "C" void racc_pt_test_sample_material(const float* ke, const float* rnd, const float* normal, const float* wo, uint32_t n,
                                             float* wi, float* colour, int* alive) {
    ptshade::Materials mat{};
    for (int m = 0; m < 4; ++m) { mat.kd[m][0] = ke[0]; mat.kd[m][1] = ke[1]; mat.kd[m][2] = ke[2]; mat.eta[m] = ke[3]; }
    for (uint32_t i = 0; i < n; ++i) {
        ptshade::Vec w{0.f, 0.f, 0.f};
        alive[i] = ptshade::sampleMaterial(mat, 0, ptshade::Vec{normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]}, ptshade::Vec{wo[3 * i], wo[3 * i + 1], wo[3 * i + 2]},
                                           rnd[3 * i], rnd[3 * i + 1], rnd[3 * i + 2], w, colour + 3 * i) ? 1 : 0;
        wi[3 * i] = w.x; wi[3 * i + 1] = w.y; wi[3 * i + 2] = w.z;
    }
}
// Test hook: n random surface interactions (seeded) through ptshade::shadeSurface and, eight at a time, through ptshade::simd::shadeSurface8;
// returns the number of interactions whose outcome (alive flag, next ray, next payload) differs in any bit, or -1 if this build has no 8-wide form.
extern "C" long long racc_pt_test_shade8(uint32_t n, uint32_t seed, uint32_t* alive_count) {
#ifdef RACC_PT_SHADE8
    ptshade::Materials mat{};
    const float kds[4][3] = {{0.8f, 0.7f, 0.6f}, {0.1f, 0.9f, 0.2f}, {0.0f, 0.0f, 0.0f}, {0.5f, 0.5f, 0.5f}};
    const float etas[4] = {1.5f, 1.2f, 2.4f, 0.7f};      // (0.7: total internal reflection for grazing directions)
    for (int m = 0; m < 4; ++m) { for (int c = 0; c < 3; ++c) mat.kd[m][c] = kds[m][c]; mat.eta[m] = etas[m]; }
    uint32_t state = seed * 2654435761u + 12345u;
    auto rnd = [&] { state = ptshade::pcg(state + 0x9E3779B9u); return float(state >> 8) * (1.0f / 16777216.0f); };
    auto unit = [&](float out[3]) { float x, y, z, l; do { x = rnd() * 2 - 1; y = rnd() * 2 - 1; z = rnd() * 2 - 1; l = x * x + y * y + z * z; } while (l < 1e-3f); l = 1.0f / sqrtf(l); out[0] = x * l; out[1] = y * l; out[2] = z * l; };
    long long bad = 0; uint32_t aliveTotal = 0;
    for (uint32_t base = 0; base + 8 <= n; base += 8) {
        ptshade::RayRec ray[8]; ptshade::HitRec hit[8]; LightPath path[8]; uint32_t sample[8]; ptshade::ShadeTri tri[8];
        const ptshade::RayRec* rp[8]; const ptshade::HitRec* hp[8]; const LightPath* pp[8]; const ptshade::ShadeTri* tp[8];
        for (int k = 0; k < 8; ++k) {
            for (int c = 0; c < 3; ++c) ray[k].origin[c] = rnd() * 200 - 100;
            unit(ray[k].dir); ray[k].minT = 1e-3f; ray[k].maxT = 1e6f;
            hit[k].triangle = 0; hit[k].t = rnd() * 300; hit[k].u = rnd(); hit[k].v = rnd() * (1 - hit[k].u);
            for (int c = 0; c < 3; ++c) path[k].weight[c] = rnd() < 0.1f ? rnd() * 0.02f : rnd();
            path[k].pixelDepth = (uint32_t(rnd() * 2073600) & 0xFFFFFFu) | (uint32_t(rnd() * 6) << 24);
            sample[k] = uint32_t(rnd() * 64);
            unit(tri[k].n0); unit(tri[k].n1); unit(tri[k].n2); unit(tri[k].ng);
            if (rnd() < 0.05f) { tri[k].n0[0] = tri[k].n1[0] = tri[k].n2[0] = 0.0f; }      // (normals with a small x: the other tangent frame)
            if (rnd() < 0.01f) hit[k].t = INFINITY;                                         // a non-finite origin: the path dies
            tri[k].material = uint32_t(rnd() * 4) & 3u; tri[k].pad[0] = tri[k].pad[1] = tri[k].pad[2] = 0;
            rp[k] = &ray[k]; hp[k] = &hit[k]; pp[k] = &path[k]; tp[k] = &tri[k];
        }
        ptshade::RayRec nr8[8]; LightPath np8[8];
        const unsigned alive8 = ptshade::simd::shadeSurface8(mat, rp, hp, pp, sample, tp, nr8, np8);
        for (int k = 0; k < 8; ++k) {
            ptshade::RayRec nr; LightPath np;
            const bool alive = ptshade::shadeSurface(mat, ray[k], hit[k], path[k], sample[k], tri[k].n0, tri[k].n1, tri[k].n2, ptshade::Vec{tri[k].ng[0], tri[k].ng[1], tri[k].ng[2]}, tri[k].material, nr, np);
            const bool a8 = ((alive8 >> k) & 1u) != 0;
            if (alive != a8 || (alive && (std::memcmp(&nr, &nr8[k], sizeof(nr)) != 0 || std::memcmp(&np, &np8[k], sizeof(np)) != 0))) ++bad;
            aliveTotal += alive ? 1u : 0u;
        }
    }
    // ... and primary rays: 64 x 8 pixels of four samples, the eight-wide form against primaryRay
    ptshade::Camera cam{{1.5f, 20.0f, -60.0f}, {0.0011f, 0.0f, 0.0002f}, {0.0f, -0.0011f, 0.0001f}, {-0.9f, 0.55f, 0.8f}};
    for (uint32_t smp = 0; smp < 4; ++smp)
        for (uint32_t y = 0; y < 64; ++y) {
            ptshade::RayRec r8[8]; LightPath p8[8];
            const uint32_t x = (y * 8u) % 1912u, pixel = y * 1920u + x;
            ptshade::simd::primaryRay8(cam, x, y + 7u, pixel, smp + seed, r8, p8);
            for (uint32_t k = 0; k < 8; ++k) {
                ptshade::RayRec r1; LightPath p1;
                ptshade::primaryRay(cam, x + k, y + 7u, pixel + k, smp + seed, r1, p1);
                if (std::memcmp(&r1, &r8[k], sizeof(r1)) != 0 || std::memcmp(&p1, &p8[k], sizeof(p1)) != 0) ++bad;
            }
        }
    if (alive_count) *alive_count = aliveTotal;
    return bad;
#else
    (void)n; (void)seed; if (alive_count) *alive_count = 0;
    return -1;
#endif
}
extern "C" void racc_pt_test_uniform(uint32_t pixel, uint32_t sample, uint32_t depth, uint32_t stream, uint32_t n, float* out) {
    for (uint32_t i = 0; i < n; ++i) out[i] = ptshade::uniformKeyed(ptshade::pathKey(pixel + i, sample), depth, stream);
}

extern "C" int racc_pt_render_file(const char* scene_bin, int device, uint32_t width, uint32_t height,
                                   uint32_t spp_first, uint32_t spp_count, uint32_t max_depth,
                                   uint32_t cpu_threads, double* rgb_sum, racc_pt_stats* stats) {
    if (!scene_bin || !rgb_sum || !width || !height || !spp_count) return -1;
    if (uint64_t(width) * height > (1u << 24)) { std::fprintf(stderr, "racc_pt: the payload keeps the pixel index in 24 bits (LightPath.h:16)\n"); return -1; }
    ptscene::Scene sc;
    if (int rc = ptscene::load(scene_bin, width, height, sc)) return rc;
    const ptscene::SceneHeader& hdr = sc.hdr;
    const uint32_t T = hdr.triangleCount, V = hdr.vertexCount;
    Renderer r;
    r.scene = &sc;
    r.view = ptshade::SceneView{sc.indices.data(), sc.triangleMaterials.data(), sc.normals.data(), sc.vertices.data(), T};
    r.tris.resize(T);
    {
        const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nt; ++t)
            pool.emplace_back([&r, T, nt, t] { for (uint32_t k = uint32_t(uint64_t(T) * t / nt); k < uint32_t(uint64_t(T) * (t + 1) / nt); ++k) ptshade::buildShadeTri(r.view, k, r.tris[k]); });
        for (std::thread& th : pool) th.join();
    }
    r.width = width; r.height = height; r.tilesX = width / 128; r.tilesY = height / 128;
    r.maxDepth = max_depth ? max_depth : hdr.maxDepth;
    r.sampleFirst = spp_first; r.sampleCount = spp_count;
    std::vector<std::atomic<int64_t>> frame(size_t(width) * height * 3);
    for (auto& a : frame) a.store(0, std::memory_order_relaxed);
    r.frame.swap(frame);

    racc::init();
    racc::GpuContext gpu = racc::gpuContextForDevice(device);
    if (!gpu) { std::fprintf(stderr, "racc_pt: no gfx950 device %d\n", device); return -3; }
    racc::Configuration cfg = racc::defaultConfiguration(gpu);
    if (cpu_threads) cfg.cpuThreads = cpu_threads;
    racc::Context* ctx = racc::createContext(cfg);
    if (!ctx) return -3;
    r.info = racc::info(ctx);
    r.payload.resize(size_t(r.info.rayStreamCount) * r.info.rayStreamSize);
    r.sampleIndex.resize(size_t(r.info.rayStreamCount) * r.info.rayStreamSize);
    racc::Scene* scene = racc::createScene(ctx, reinterpret_cast<const racc::Vertex*>(sc.vertices.data()), V, sc.indices.data(), T * 3);
    racc::Environment* environment = racc::createEnvironment(ctx, reinterpret_cast<const racc::Color*>(sc.env.data()), hdr.environmentWidth, hdr.environmentHeight);
    if (!scene || !environment) { racc::destroy(ctx); return -3; }

    racc::RenderCallbacks cb = {&r, Renderer::spawn, Renderer::shade};
    const auto t0 = std::chrono::steady_clock::now();
    const racc::Stats st = racc::render(ctx, scene, environment, cb);
    const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (size_t i = 0; i < r.frame.size(); ++i) rgb_sum[i] = double(r.frame[i].load(std::memory_order_relaxed)) / kFixed;
    if (stats) {
        stats->rays_traced = st.raysTraced; stats->primary_rays = r.primaries.load(); stats->seconds = seconds;
        stats->tiles_x = r.tilesX; stats->tiles_y = r.tilesY; stats->max_depth = r.maxDepth; stats->threads = r.info.threadCount;
        stats->triangles = T; stats->reserved = 0;
    }
    racc::destroy(environment); racc::destroy(scene); racc::destroy(ctx); racc::deinit();
    return 0;
}
