// pt_device.hip — device-resident path-tracing consumer of the intersect path (BASELINE.json configs[4]).
//
// The reference's renderer (Renderer/PathTracingRenderer.cpp) shades on CPU threads through the spawn/shade callbacks
// and ships every ray batch to the GPU and back; pathtracer.cpp keeps that shape (it is what a user of the reference's
// API writes) and is bound by host shading: 16 shading threads keep the MI355X 14 % busy.  This consumer is the
// MI355X-shaped alternative: rays, hits and path payloads never leave HBM.  Per batch of samples
//     ptGenKernel      camera rays for every pixel x sample of the batch        (Camera.cpp:55-85)
//     repeat: racc_hip_intersect_device (the engine, device-pointer entry of the C-ABI)
//             ptShadeKernel: misses add weight x radiance to the fixed-point frame buffer, hits sample the material and
//             write the next ray through a workgroup-aggregated (ballots + LDS + one atomic per 1024 rays) compaction   (…Renderer.cpp:72-566)
// until no path is alive.  The per-ray arithmetic is pt_shade.h, the same source the host consumer compiles, the RNG is
// keyed by (pixel, sample, depth) and the frame buffer is integer, so both consumers render the SAME image bit for bit
// (tests/test_gpu_pathtracer.py) — the host consumer is this kernel's oracle.
// Compiled with -fgpu-flush-denormals-to-zero: the host callbacks run with FTZ/DAZ set (Threading.h:78-79).

#include <hip/hip_runtime.h>

#include "pt_scene.h"
#include "pt_shade.h"
#include "racc_hip.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" {

typedef struct racc_pt_stats {     // same layout as pathtracer.cpp's
    uint64_t rays_traced;
    uint64_t primary_rays;
    double seconds;                // wall time of the render loop (scene build/upload excluded, as in racc_pt_render_file)
    uint32_t tiles_x, tiles_y;
    uint32_t max_depth;
    uint32_t threads;              // 0: nothing is shaded on the host
    uint32_t triangles;
    uint32_t reserved;             // bounce rounds executed
} racc_pt_stats;

// Renders samples [spp_first, spp_first + spp_count) of every pixel and writes the SUM of their radiance (not the mean)
// as rgb doubles, row-major width*height*3.  samples_per_batch 0 => 8.  Returns 0 on success.
int racc_ptdev_render_file(const char* scene_bin, int device, uint32_t width, uint32_t height,
                           uint32_t spp_first, uint32_t spp_count, uint32_t max_depth /* 0 = from the file */,
                           uint32_t samples_per_batch, double* rgb_sum, racc_pt_stats* stats);

// Test hook: the next racc_ptdev_render_file call copies the ray buffer the engine was handed in bounce round `round` (0 = the
// first batch's primaries; rounds count over all pipelines in issue order) and the hit records it returned — at most `capacity`
// of each — to the two host arrays (Ray 32 B, Result 16 B), so that a test can re-trace exactly what the device consumer traced
// with the oracle.  *count = records copied (0 if the render had fewer rounds).  One-shot; capacity 0 disarms.
void racc_ptdev_capture_round(uint32_t round, void* rays_out, void* hits_out, uint32_t capacity, uint32_t* count);
}

namespace {

using namespace ptshade;

struct Capture { uint32_t round = 0; void* rays = nullptr; void* hits = nullptr; uint32_t capacity = 0; uint32_t* count = nullptr; } g_capture;

// Pixels are walked in 8x8 blocks so that the 64 rays of a wave start out as one coherent bundle.
__global__ void __launch_bounds__(256) ptGenKernel(Camera cam, uint32_t width, uint32_t regionW, uint32_t regionH,
                                                   uint32_t sampleFirst, uint32_t samples,
                                                   RayRec* rays, PathRec* paths, uint32_t* sampleIdx) {
    const uint32_t perSample = regionW * regionH;
    const uint64_t total = uint64_t(perSample) * samples;
    for (uint64_t i = uint64_t(blockIdx.x) * 256u + threadIdx.x; i < total; i += uint64_t(gridDim.x) * 256u) {
        const uint32_t s = uint32_t(i / perSample), pi = uint32_t(i % perSample);
        const uint32_t block = pi >> 6, within = pi & 63u, blocksX = regionW >> 3;
        const uint32_t x = (block % blocksX) * 8u + (within & 7u), y = (block / blocksX) * 8u + (within >> 3);
        const uint32_t pixel = y * width + x;
        RayRec ray; PathRec path;
        primaryRay(cam, x, y, pixel, sampleFirst + s, ray, path);
        rays[i] = ray; paths[i] = path; sampleIdx[i] = sampleFirst + s;
    }
}

// Workgroups of 1024 threads: the compaction counter is ONE address, and same-address atomics retire at roughly one per
// 10 ns on this part — one atomic per wave (4.8 M of them for a 1080p x 64 spp frame) made this kernel cost 50 ms, a third
// of the frame.  Wave ballots -> 16 per-wave counts in LDS -> one atomic per workgroup: 16x fewer.
constexpr int kShadeBlock = 1024;

__global__ void __launch_bounds__(kShadeBlock) ptShadeKernel(const ShadeTri* tris, uint32_t triangleCount, Materials mat, uint32_t maxDepth,
                                                     const RayRec* rays, const HitRec* hits, const PathRec* paths, const uint32_t* sampleIdx,
                                                     uint32_t count, RayRec* outRays, PathRec* outPaths, uint32_t* outSample,
                                                     uint32_t* outCount, unsigned long long* frame) {
    __shared__ uint32_t waveCount[kShadeBlock / 64];
    __shared__ uint32_t blockFirst;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t base = blockIdx.x * uint32_t(kShadeBlock); base < count; base += gridDim.x * uint32_t(kShadeBlock)) {
        const uint32_t i = base + threadIdx.x;
        bool alive = false;
        RayRec nextRay; PathRec nextPath; uint32_t sample = 0;
        if (i < count) {
            const HitRec hit = hits[i];
            const PathRec path = paths[i];
            if (hit.triangle == 0xFFFFFFFFu) {
                long long add[3]; bool valid[3];
                missContribution(hit, path, add, valid);
                const uint32_t pixel = path.pixelDepth & 0xFFFFFFu;
                for (int ch = 0; ch < 3; ++ch)
                    if (valid[ch] && add[ch] != 0) atomicAdd(frame + size_t(pixel) * 3 + ch, (unsigned long long)add[ch]);
            } else if ((path.pixelDepth >> 24) < maxDepth && hit.triangle < triangleCount) {     // PathTracingRenderer.cpp:113-114
                sample = sampleIdx[i];
                const float4* tp = reinterpret_cast<const float4*>(tris + hit.triangle);
                const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2], q3 = tp[3];                      // one 64 B record
                const float n0[3] = {q0.x, q0.y, q0.z}, n1[3] = {q0.w, q1.x, q1.y}, n2[3] = {q1.z, q1.w, q2.x};
                alive = shadeSurface(mat, rays[i], hit, path, sample, n0, n1, n2, Vec{q2.y, q2.z, q2.w}, __float_as_uint(q3.x), nextRay, nextPath);
            }
        }
        // workgroup-aggregated compaction: ballot ranks inside a wave, per-wave counts through LDS, one atomic per workgroup
        const unsigned long long mask = __ballot(alive);
        if (lane == 0) waveCount[wave] = uint32_t(__popcll(mask));
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
            for (int w = 0; w < kShadeBlock / 64; ++w) total += waveCount[w];
            blockFirst = total ? atomicAdd(outCount, total) : 0u;
        }
        __syncthreads();
        if (alive) {
            uint32_t slot = blockFirst + uint32_t(__popcll(mask & ((1ull << lane) - 1ull)));
            for (uint32_t w = 0; w < wave; ++w) slot += waveCount[w];
            outRays[slot] = nextRay; outPaths[slot] = nextPath; outSample[slot] = sample;
        }
        __syncthreads();     // waveCount / blockFirst are rewritten by the next iteration
    }
}

#define PT_HIP(call)                                                                                             \
    do {                                                                                                         \
        hipError_t e_ = (call);                                                                                  \
        if (e_ != hipSuccess) { std::fprintf(stderr, "racc_ptdev: %s: %s\n", #call, hipGetErrorString(e_)); rc = -4; goto done; } \
    } while (0)
#define PT_RACC(call)                                                                                            \
    do {                                                                                                         \
        if ((call) != RACC_HIP_OK) { std::fprintf(stderr, "racc_ptdev: %s: %s\n", #call, racc_hip_last_error()); rc = -3; goto done; } \
    } while (0)

}  // namespace

extern "C" void racc_ptdev_capture_round(uint32_t round, void* rays_out, void* hits_out, uint32_t capacity, uint32_t* count) {
    g_capture = Capture{round, rays_out, hits_out, (rays_out && hits_out) ? capacity : 0u, count};
    if (count) *count = 0;
}

extern "C" int racc_ptdev_render_file(const char* scene_bin, int device, uint32_t width, uint32_t height,
                                      uint32_t spp_first, uint32_t spp_count, uint32_t max_depth,
                                      uint32_t samples_per_batch, double* rgb_sum, racc_pt_stats* stats) {
    if (!scene_bin || !rgb_sum || !width || !height || !spp_count) return -1;
    if (uint64_t(width) * height > (1u << 24)) { std::fprintf(stderr, "racc_ptdev: the payload keeps the pixel index in 24 bits (LightPath.h:16)\n"); return -1; }
    ptscene::Scene sc;
    if (int lrc = ptscene::load(scene_bin, width, height, sc)) return lrc;
    const uint32_t T = sc.hdr.triangleCount, V = sc.hdr.vertexCount;
    const uint32_t tilesX = width / 128, tilesY = height / 128;          // TiledRenderer.cpp:20-22: whole 128x128 tiles only
    const uint32_t regionW = tilesX * 128, regionH = tilesY * 128;
    const uint32_t depthLimit = max_depth ? max_depth : sc.hdr.maxDepth;
    const uint64_t perSample = uint64_t(regionW) * regionH;
    uint32_t S = samples_per_batch ? samples_per_batch : 8;   // 8 x 1.97M rays at 1080p: 1.9 GB of buffers, 6 host syncs per 8 spp
    if (S > spp_count) S = spp_count;
    while (S > 1 && perSample * S > (1ull << 30)) --S;                    // ray counts are 32-bit in the C-ABI

    int rc = 0;
    racc_hip_ctx* ctx = nullptr; racc_host_scene* host = nullptr; racc_hip_scene* scene = nullptr; racc_hip_env* env = nullptr;
    // Four pipelines, each a lane of the engine with its own HIP stream and buffers: sample batches are independent, so while
    // one pipeline's launch drains (its last long rays) the others' launches fill the machine (1080p x 64 spp, Grays/s end to
    // end: 2 pipelines 3.40, 3: 3.50, 4: 3.53-3.57, 6: 3.56).
    constexpr int kPipes = 4;
    struct Pipe {
        hipStream_t stream = nullptr; hipEvent_t done = nullptr;
        RayRec* rays[2] = {nullptr, nullptr}; PathRec* paths[2] = {nullptr, nullptr}; uint32_t* sample[2] = {nullptr, nullptr};
        HitRec* hits = nullptr; uint32_t* dCount = nullptr; uint32_t* hCount = nullptr;
        int cur = 0; uint32_t n = 0; bool busy = false;
    } pipe[kPipes];
    ShadeTri* dTris = nullptr;
    unsigned long long* dFrame = nullptr;
    uint64_t raysTraced = 0, primaries = 0; uint32_t rounds = 0;
    double seconds = 0.0;
    const size_t cap = size_t(perSample) * S;
    const size_t frameWords = size_t(width) * height * 3;
    const int pipes = (spp_count + S - 1) / S > 1 ? kPipes : 1;
    if (perSample == 0) {   // nothing to render (viewport smaller than one tile): an all-zero image, like the host consumer
        for (size_t i = 0; i < frameWords; ++i) rgb_sum[i] = 0.0;
        goto fill;
    }
    {
        racc_hip_options opts{};
        opts.struct_size = sizeof(opts);
        opts.lanes = kPipes;
        PT_RACC(racc_hip_create(device, &opts, &ctx));
        PT_RACC(racc_host_scene_build(sc.vertices.data(), V, sc.indices.data(), T * 3, &host));
        const void *nodes = nullptr, *pairs = nullptr; const uint32_t* remap = nullptr;
        uint32_t nNodes = 0, nPairsPadded = 0, nPairs = 0, nRemap = 0;
        PT_RACC(racc_host_scene_blobs(host, &nodes, &nNodes, &pairs, &nPairsPadded, &nPairs, &remap, &nRemap));
        PT_RACC(racc_hip_scene_upload(ctx, nodes, nNodes, pairs, nPairsPadded, remap, nRemap, &scene));
        racc_host_scene_free(host); host = nullptr;
        PT_RACC(racc_hip_env_upload(ctx, sc.env.data(), sc.hdr.environmentWidth, sc.hdr.environmentHeight, &env));

        PT_HIP(hipSetDevice(device));
        {
            const SceneView hostView{sc.indices.data(), sc.triangleMaterials.data(), sc.normals.data(), sc.vertices.data(), T};
            std::vector<ShadeTri> tris(T);
            for (uint32_t t = 0; t < T; ++t) buildShadeTri(hostView, t, tris[t]);
            PT_HIP(hipMalloc(&dTris, size_t(T) * sizeof(ShadeTri)));
            PT_HIP(hipMemcpy(dTris, tris.data(), size_t(T) * sizeof(ShadeTri), hipMemcpyHostToDevice));
        }
        for (int p = 0; p < pipes; ++p) {
            Pipe& P = pipe[p];
            PT_HIP(hipStreamCreateWithFlags(&P.stream, hipStreamNonBlocking));
            PT_HIP(hipEventCreateWithFlags(&P.done, hipEventDisableTiming));
            for (int k = 0; k < 2; ++k) {
                PT_HIP(hipMalloc(&P.rays[k], cap * sizeof(RayRec)));
                PT_HIP(hipMalloc(&P.paths[k], cap * sizeof(PathRec)));
                PT_HIP(hipMalloc(&P.sample[k], cap * 4));
            }
            PT_HIP(hipMalloc(&P.hits, cap * sizeof(HitRec)));
            PT_HIP(hipMalloc(&P.dCount, 4));
            PT_HIP(hipHostMalloc(&P.hCount, 4));
        }
        PT_HIP(hipMalloc(&dFrame, frameWords * 8));
        PT_HIP(hipMemset(dFrame, 0, frameWords * 8));
        PT_HIP(hipStreamSynchronize(nullptr));      // hipMemset is asynchronous; the pipelines' streams do not wait for the null stream
        PT_HIP(hipDeviceSynchronize());

        hipDeviceProp_t prop;
        PT_HIP(hipGetDeviceProperties(&prop, device));
        const uint32_t maxBlocks = uint32_t(prop.multiProcessorCount) * 8u;
        const auto t0 = std::chrono::steady_clock::now();
        uint32_t s0 = 0;            // next sample batch to start
        int live = 0;
        // One bounce of pipeline P: trace, shade + compact into the other buffer, read the survivor count back; `done` fires
        // when the count has landed.
        auto bounce = [&](int p) -> int {
            Pipe& P = pipe[p];
            if (racc_hip_intersect_device(ctx, scene, env, P.rays[P.cur], P.hits, P.n, uint32_t(p), P.stream) != RACC_HIP_OK) return -3;
            if (g_capture.capacity && g_capture.round == rounds) {      // test hook: hand this round's rays and hits to the host
                const uint32_t n = P.n < g_capture.capacity ? P.n : g_capture.capacity;
                if (racc_hip_stream_synchronize(ctx, P.stream) != RACC_HIP_OK) return -3;
                if (hipMemcpy(g_capture.rays, P.rays[P.cur], size_t(n) * sizeof(RayRec), hipMemcpyDeviceToHost) != hipSuccess) return -4;
                if (hipMemcpy(g_capture.hits, P.hits, size_t(n) * sizeof(HitRec), hipMemcpyDeviceToHost) != hipSuccess) return -4;
                if (g_capture.count) *g_capture.count = n;
                g_capture.capacity = 0;
            }
            raysTraced += P.n; ++rounds;
            if (hipMemsetAsync(P.dCount, 0, 4, P.stream) != hipSuccess) return -4;
            const uint32_t need = (P.n + uint32_t(kShadeBlock) - 1u) / uint32_t(kShadeBlock);
            const uint32_t blocks = need < uint32_t(prop.multiProcessorCount) * 2u ? need : uint32_t(prop.multiProcessorCount) * 2u;
            hipLaunchKernelGGL(ptShadeKernel, dim3(blocks), dim3(kShadeBlock), 0, P.stream, dTris, T, sc.mat, depthLimit,
                               P.rays[P.cur], P.hits, P.paths[P.cur], P.sample[P.cur], P.n, P.rays[P.cur ^ 1], P.paths[P.cur ^ 1], P.sample[P.cur ^ 1],
                               P.dCount, dFrame);
            if (hipGetLastError() != hipSuccess) return -4;
            if (hipMemcpyAsync(P.hCount, P.dCount, 4, hipMemcpyDeviceToHost, P.stream) != hipSuccess) return -4;
            if (hipEventRecord(P.done, P.stream) != hipSuccess) return -4;
            return 0;
        };
        for (;;) {
            for (int p = 0; p < pipes; ++p) {          // start a new sample batch on every idle pipeline
                Pipe& P = pipe[p];
                if (P.busy || s0 >= spp_count) continue;
                const uint32_t ns = (spp_count - s0 < S) ? spp_count - s0 : S;
                P.n = uint32_t(perSample * ns); P.cur = 0;
                const uint32_t blocks = (P.n + 255u) / 256u < maxBlocks ? (P.n + 255u) / 256u : maxBlocks;
                hipLaunchKernelGGL(ptGenKernel, dim3(blocks), dim3(256), 0, P.stream, sc.cam, width, regionW, regionH, spp_first + s0, ns,
                                   P.rays[0], P.paths[0], P.sample[0]);
                PT_HIP(hipGetLastError());
                primaries += P.n;
                s0 += ns;
                if (int e = bounce(p)) { std::fprintf(stderr, "racc_ptdev: bounce failed: %s\n", racc_hip_last_error()); rc = e; goto done; }
                P.busy = true; ++live;
            }
            if (!live) break;
            // wait for whichever pipeline finishes its bounce first (poll; block on one when it is the only one)
            int ready = -1;
            while (ready < 0) {
                for (int p = 0; p < pipes && ready < 0; ++p)
                    if (pipe[p].busy) {
                        const hipError_t q = live == 1 ? hipEventSynchronize(pipe[p].done) : hipEventQuery(pipe[p].done);
                        if (q == hipSuccess) ready = p;
                        else if (q != hipErrorNotReady) { std::fprintf(stderr, "racc_ptdev: %s\n", hipGetErrorString(q)); rc = -4; goto done; }
                    }
            }
            Pipe& P = pipe[ready];
            P.n = *P.hCount;
            P.cur ^= 1;
            if (P.n) {
                if (int e = bounce(ready)) { std::fprintf(stderr, "racc_ptdev: bounce failed: %s\n", racc_hip_last_error()); rc = e; goto done; }
            } else {
                P.busy = false; --live;
            }
        }
        // (not hipDeviceSynchronize: the engine's call also reports a traversal watchdog trip — stale hit records would
        //  otherwise be shaded as if valid and the frame returned as complete)
        PT_RACC(racc_hip_synchronize(ctx));
        seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::vector<long long> frame(frameWords);
        PT_HIP(hipMemcpy(frame.data(), dFrame, frameWords * 8, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < frameWords; ++i) rgb_sum[i] = double(frame[i]) / kFixed;
    }
fill:
    if (stats) {
        stats->rays_traced = raysTraced; stats->primary_rays = primaries; stats->seconds = seconds;
        stats->tiles_x = tilesX; stats->tiles_y = tilesY; stats->max_depth = depthLimit; stats->threads = 0;
        stats->triangles = T; stats->reserved = rounds;
    }
done:
    if (host) racc_host_scene_free(host);
    if (rc != 0) hipDeviceSynchronize();     // nothing may still be writing into the buffers freed below
    for (int p = 0; p < kPipes; ++p) {
        Pipe& P = pipe[p];
        for (int k = 0; k < 2; ++k) { if (P.rays[k]) hipFree(P.rays[k]); if (P.paths[k]) hipFree(P.paths[k]); if (P.sample[k]) hipFree(P.sample[k]); }
        if (P.hits) hipFree(P.hits);
        if (P.dCount) hipFree(P.dCount);
        if (P.hCount) hipHostFree(P.hCount);
        if (P.done) hipEventDestroy(P.done);
        if (P.stream) hipStreamDestroy(P.stream);
    }
    if (dFrame) hipFree(dFrame);
    if (dTris) hipFree(dTris);
    if (env) racc_hip_env_free(ctx, env);
    if (scene) racc_hip_scene_free(ctx, scene);
    if (ctx) racc_hip_destroy(ctx);
    return rc;
}
