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
// What differs, deliberately: scalar C++ instead of 8-wide AVX2; a counter-based RNG keyed by (pixel, sample, depth)
// instead of rand()-seeded streams (Camera.cpp:58, PathTracingRenderer.cpp:102), so a frame is reproducible; and a
// 64-bit fixed-point frame buffer, so the image does not depend on the order in which worker threads shade.
// All intersection work goes through racc::render (include/RayAccelerator.h), i.e. the MI355X engine.

#include "RayAccelerator.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

#pragma pack(push, 1)
struct SceneHeader {   // Renderer/main.cpp:118-133
    uint32_t maxDepth, vertexCount, triangleCount;
    uint16_t viewportWidth, viewportHeight, environmentWidth, environmentHeight;
    float origin[3], target[3], up[3], fov;
};
#pragma pack(pop)
static_assert(sizeof(SceneHeader) == 60, "scene header layout");

struct LightPath { float weight[3]; uint32_t pixel; };   // LightPath.h:14-17

struct Vec { float x, y, z; };
inline Vec operator+(Vec a, Vec b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline Vec operator-(Vec a, Vec b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline Vec operator*(Vec a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(Vec a, Vec b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec cross(Vec a, Vec b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline Vec normalize(Vec a) { return a * (1.0f / std::sqrt(dot(a, a))); }

inline uint32_t pcg(uint32_t x) {
    x = x * 747796405u + 2891336453u;
    const uint32_t w = ((x >> ((x >> 28) + 4)) ^ x) * 277803737u;
    return (w >> 22) ^ w;
}
inline float uniform(uint32_t pixel, uint32_t sample, uint32_t depth, uint32_t stream) {
    const uint32_t h = pcg(pcg(pcg(pixel) ^ (sample * 0x9E3779B9u)) ^ (depth * 0x85EBCA6Bu + stream * 0xC2B2AE35u));
    return float(h >> 8) * (1.0f / 16777216.0f);
}

constexpr double kFixed = 1048576.0;   // 2^20: frame-buffer resolution of one accumulated contribution

struct Renderer {
    std::vector<racc::Vertex> vertices;
    std::vector<uint32_t> indices;
    std::vector<uint16_t> triangleMaterials;
    std::vector<float> normals;            // xyzw per vertex
    float kd[4][3], eta[4];
    Vec camOrigin, camRight, camUp, camView;
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
        for (uint32_t y = 0; y < 128; ++y)
            for (uint32_t x = 0; x < 128; ++x) {
                const uint32_t pixel = (ty + y) * r->width + tx + x;
                const float px = float(tx + x) + uniform(pixel, sample, 0, 1), py = float(ty + y) + uniform(pixel, sample, 0, 2);
                const Vec d = normalize(r->camView + r->camRight * px + r->camUp * py);
                racc::Ray& ray = out->rays[out->count];
                ray.origin[0] = r->camOrigin.x; ray.origin[1] = r->camOrigin.y; ray.origin[2] = r->camOrigin.z; ray.minT = 0.0f;
                ray.dir[0] = d.x; ray.dir[1] = d.y; ray.dir[2] = d.z; ray.maxT = 1e6f;
                lp[out->count] = LightPath{{1.0f, 1.0f, 1.0f}, pixel};
                r->sampleOf(out->index, out->count) = sample;
                ++out->count;
            }
        r->primaries += 128 * 128;
        return job != jobs - 1;
    }

    std::vector<uint32_t> sampleIndex;     // [stream][slot]: which sample a path belongs to (RNG key)
    uint32_t& sampleOf(uint32_t stream, uint32_t slot) { return sampleIndex[size_t(stream) * info.rayStreamSize + slot]; }

    // Materials.cpp:39-151, scalar.  Returns false if the path dies.
    bool sampleMaterial(unsigned m, Vec n, Vec wo, float r1, float r2, float r3, Vec& wi, float colour[3]) const {
        const float cosi = std::max(dot(n, wo), 0.0f);
        const Vec refl = n * (2.0f * cosi) - wo;
        const float e = eta[m];
        const float k = 1.0f + e * e * (cosi * cosi - 1.0f);
        float fresnel = 1.0f;                      // total internal reflection (Materials.cpp:83: blendv on the sign of k)
        if (k >= 0.0f) {
            const float cost = std::sqrt(k);
            const float rper = (e * cosi - cost) / (e * cosi + cost);
            const float rpar = -(e * cost - cosi) / (e * cost + cosi);
            fresnel = 0.5f * (rpar * rpar + rper * rper);
        }
        Vec bu = std::fabs(n.x) > 0.1f ? Vec{-n.z, 0.0f, n.x} : Vec{0.0f, -n.z, n.y};   // Materials.cpp:86-93
        bu = normalize(bu);
        const Vec bv = cross(n, bu);
        const float phi = 6.2831853f * r1, s = std::sqrt(r2), c = std::sqrt(1.0f - r2);
        const Vec diffuse = normalize(n * c + (bu * std::cos(phi) + bv * std::sin(phi)) * s);
        const float s0 = 3.0f * fresnel, s1 = kd[m][0] + kd[m][1] + kd[m][2], sum = s0 + s1;    // Materials.cpp:121-128
        const bool pickDiffuse = r3 * sum >= s0;
        wi = pickDiffuse ? diffuse : refl;
        float rgb[3];
        for (int ch = 0; ch < 3; ++ch) rgb[ch] = pickDiffuse ? kd[m][ch] : fresnel;
        const float denom = rgb[0] + rgb[1] + rgb[2];
        if (!(denom > 0.0f)) return false;
        const float scale = sum / denom;                                                        // Materials.cpp:138
        for (int ch = 0; ch < 3; ++ch) colour[ch] = rgb[ch] * scale;
        return true;
    }

    static void shade(void* data, unsigned, const racc::RayStream* in, unsigned start, unsigned end, racc::RayStream* out) {
        Renderer* r = static_cast<Renderer*>(data);
        const LightPath* lin = r->payload.data() + size_t(in->index) * r->info.rayStreamSize;
        LightPath* lout = r->payload.data() + size_t(out->index) * r->info.rayStreamSize;
        for (unsigned i = start; i < end; ++i) {
            const racc::Ray& ray = in->rays[i];
            const racc::Result& hit = in->results[i];
            const LightPath& lp = lin[i];
            const uint32_t pixel = lp.pixel & 0xFFFFFFu, depth = lp.pixel >> 24;
            if (hit.triangle == racc::invalidTriangle) {                    // PathTracingRenderer.cpp:505-563
                const float env[3] = {hit.miss.r, hit.miss.g, hit.miss.b};
                for (int ch = 0; ch < 3; ++ch) {
                    const double v = double(env[ch]) * double(lp.weight[ch]);
                    if (std::isfinite(v)) r->frame[size_t(pixel) * 3 + ch].fetch_add(int64_t(std::llround(v * kFixed)), std::memory_order_relaxed);
                }
                continue;
            }
            if (depth >= r->maxDepth || hit.triangle >= r->indices.size() / 3) continue;        // :113-114
            const uint32_t sample = r->sampleOf(in->index, i);
            const uint32_t* tri = &r->indices[size_t(hit.triangle) * 3];
            const float u = hit.hit.u, v = hit.hit.v, w = 1.0f - u - v;                          // :218-227: w,u,v weight index 0,1,2
            const float* n0 = &r->normals[size_t(tri[0]) * 4];
            const float* n1 = &r->normals[size_t(tri[1]) * 4];
            const float* n2 = &r->normals[size_t(tri[2]) * 4];
            Vec n = normalize(Vec{n0[0] * w + n1[0] * u + n2[0] * v, n0[1] * w + n1[1] * u + n2[1] * v, n0[2] * w + n1[2] * u + n2[2] * v});
            const racc::Vertex &a = r->vertices[tri[0]], &b = r->vertices[tri[1]], &c = r->vertices[tri[2]];
            Vec ng = normalize(cross(Vec{b.x - a.x, b.y - a.y, b.z - a.z}, Vec{c.x - a.x, c.y - a.y, c.z - a.z}));
            const Vec d{ray.dir[0], ray.dir[1], ray.dir[2]};
            const Vec wo = d * -1.0f;
            if (dot(ng, wo) < 0.0f) ng = ng * -1.0f;        // geometric normal toward the viewer side
            if (dot(n, wo) < 0.0f) n = n * -1.0f;
            Vec wi;
            float colour[3];
            const unsigned m = r->triangleMaterials[hit.triangle] & 3u;
            if (!r->sampleMaterial(m, n, wo, uniform(pixel, sample, depth + 1, 3), uniform(pixel, sample, depth + 1, 4), uniform(pixel, sample, depth + 1, 5), wi, colour)) continue;
            const float wgt[3] = {lp.weight[0] * colour[0], lp.weight[1] * colour[1], lp.weight[2] * colour[2]};
            if (!(wgt[0] > 0.01f || wgt[1] > 0.01f || wgt[2] > 0.01f)) continue;               // :394-399
            if (!(dot(wi, ng) > 0.0f)) continue;                                                 // :401-403 (no transmission)
            const Vec p = Vec{ray.origin[0], ray.origin[1], ray.origin[2]} + d * hit.hit.t + ng * 1e-4f;   // :410-412
            if (!(std::isfinite(p.x + p.y + p.z) && std::isfinite(wi.x + wi.y + wi.z))) continue;         // :416-418
            racc::Ray& o = out->rays[out->count];
            o.origin[0] = p.x; o.origin[1] = p.y; o.origin[2] = p.z; o.minT = 1e-3f;
            o.dir[0] = wi.x; o.dir[1] = wi.y; o.dir[2] = wi.z; o.maxT = 1e6f;
            lout[out->count] = LightPath{{wgt[0], wgt[1], wgt[2]}, pixel | ((depth + 1) << 24)};   // :414
            r->sampleOf(out->index, out->count) = sample;
            ++out->count;
        }
    }
};

}  // namespace

extern "C" int racc_pt_render_file(const char* scene_bin, int device, uint32_t width, uint32_t height,
                                   uint32_t spp_first, uint32_t spp_count, uint32_t max_depth,
                                   uint32_t cpu_threads, double* rgb_sum, racc_pt_stats* stats) {
    if (!scene_bin || !rgb_sum || !width || !height || !spp_count) return -1;
    if (uint64_t(width) * height > (1u << 24)) { std::fprintf(stderr, "racc_pt: the payload keeps the pixel index in 24 bits (LightPath.h:16)\n"); return -1; }
    FILE* f = std::fopen(scene_bin, "rb");
    SceneHeader hdr;
    if (!f || std::fread(&hdr, sizeof(hdr), 1, f) != 1) { if (f) std::fclose(f); std::fprintf(stderr, "racc_pt: cannot read %s\n", scene_bin); return -2; }
    Renderer r;
    const uint32_t T = hdr.triangleCount, V = hdr.vertexCount;
    r.indices.resize(size_t(T) * 3); r.triangleMaterials.resize(T); r.vertices.resize(V); r.normals.resize(size_t(V) * 4);
    std::vector<racc::Color> env(size_t(hdr.environmentWidth) * hdr.environmentHeight);
    bool ok = std::fread(r.indices.data(), 12, T, f) == T;
    ok = ok && std::fread(r.triangleMaterials.data(), 2, T, f) == T;
    ok = ok && std::fseek(f, long(T) * 16, SEEK_CUR) == 0;                       // per-triangle normals: recomputed from the vertices
    ok = ok && std::fread(r.vertices.data(), 16, V, f) == V;
    ok = ok && std::fread(r.normals.data(), 16, V, f) == V;
    ok = ok && std::fseek(f, long(V) * 8, SEEK_CUR) == 0;                        // texture coordinates: unused by the 4 materials
    ok = ok && std::fread(env.data(), 16, env.size(), f) == env.size();
    std::fclose(f);
    if (!ok) { std::fprintf(stderr, "racc_pt: short scene file\n"); return -2; }

    const float mats[4][4] = {{0.8f, 0.8f, 0.8f, 1.0f / 1.4f}, {0.1f, 0.1f, 0.1f, 1.0f / 1.4f},    // main.cpp:165-168
                              {0.6f, 0.6f, 0.6f, 1.0f / 1.2f}, {0.3f, 0.3f, 0.3f, 1.0f / 1.2f}};
    for (int m = 0; m < 4; ++m) { for (int ch = 0; ch < 3; ++ch) r.kd[m][ch] = mats[m][ch]; r.eta[m] = mats[m][3]; }
    r.width = width; r.height = height; r.tilesX = width / 128; r.tilesY = height / 128;
    r.maxDepth = max_depth ? max_depth : hdr.maxDepth;
    r.sampleFirst = spp_first; r.sampleCount = spp_count;
    {   // Camera::lookAt, Camera.cpp:13-26
        const Vec o{hdr.origin[0], hdr.origin[1], hdr.origin[2]};
        const Vec fwd = normalize(Vec{hdr.target[0], hdr.target[1], hdr.target[2]} - o);
        const Vec right = normalize(cross(fwd, Vec{hdr.up[0], hdr.up[1], hdr.up[2]}));
        const Vec up = cross(right, fwd);
        const float ey = std::tan(0.5f * hdr.fov * 3.14159265f / 180.0f), ex = ey * float(width) / float(height);
        r.camOrigin = o;
        r.camRight = right * (-2.0f / float(width) * ex);
        r.camUp = up * (-2.0f / float(height) * ey);
        r.camView = fwd + right * ex + up * ey;
    }
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
    racc::Scene* scene = racc::createScene(ctx, r.vertices.data(), V, r.indices.data(), T * 3);
    racc::Environment* environment = racc::createEnvironment(ctx, env.data(), hdr.environmentWidth, hdr.environmentHeight);
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
