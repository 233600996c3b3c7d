// fake_engine — a GPU-free stand-in for libracc_hip.so's C-ABI, for ONE purpose: running the racc:: scheduler
// (rayaccel_amd/csrc/racc_api.cpp: CPU workers, GPU submission threads, the four stream lists) under -fsanitize=thread on a host
// without a GPU (`make tsan`; ThreadSanitizer and the ROCm runtime do not share an address space: the instrumented driver segfaults
// at HIP start-up on the GPU box).  TEST INFRASTRUCTURE: it traces nothing.  racc_hip_intersect_streams sleeps a pseudo-random
// 50-400 us (so launches of different submission threads interleave in many ways) and fills every Result from a hash of the Ray:
// two thirds "hit" a valid triangle id at t = 1, one third miss — enough for the test driver's shade callback to spawn bounces
// to full depth.  No parity claim is made with it; tests/test_gpu_render.py holds the real engine to the oracle.
#include "racc_hip.h"

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

struct racc_hip_ctx { std::atomic<uint64_t> launches{0}; };
struct racc_hip_scene { uint32_t triangles; };
struct racc_hip_env { int unused; };
struct racc_host_scene { std::vector<uint8_t> nodes, pairs; std::vector<uint32_t> remap; uint32_t triangles; };

static uint32_t hash32(uint32_t x) { x = x * 747796405u + 2891336453u; uint32_t w = ((x >> ((x >> 28) + 4)) ^ x) * 277803737u; return (w >> 22) ^ w; }

extern "C" {
const char* racc_hip_last_error(void) { return "fake engine"; }
int racc_hip_device_count(int* count) { *count = 4; return RACC_HIP_OK; }
int racc_hip_create(int, const racc_hip_options*, racc_hip_ctx** out) { *out = new racc_hip_ctx(); return RACC_HIP_OK; }
int racc_hip_destroy(racc_hip_ctx* c) { delete c; return RACC_HIP_OK; }
int racc_hip_register_host(racc_hip_ctx*, void*, uint64_t) { return RACC_HIP_OK; }
int racc_hip_unregister_host(racc_hip_ctx*, void*) { return RACC_HIP_OK; }
int racc_hip_scene_upload(racc_hip_ctx*, const void*, uint32_t, const void*, uint32_t, const uint32_t* remap, uint32_t remap_count, racc_hip_scene** out) {
    uint32_t tris = 1;
    for (uint32_t i = 0; i < remap_count; ++i) { const uint32_t t = remap[i] & 0x3FFFFFFFu; if (t + 1 > tris) tris = t + 1; }
    *out = new racc_hip_scene{tris};
    return RACC_HIP_OK;
}
int racc_hip_scene_free(racc_hip_ctx*, racc_hip_scene* s) { delete s; return RACC_HIP_OK; }
int racc_hip_env_upload(racc_hip_ctx*, const float*, uint32_t, uint32_t, racc_hip_env** out) { *out = new racc_hip_env{0}; return RACC_HIP_OK; }
int racc_hip_env_free(racc_hip_ctx*, racc_hip_env* e) { delete e; return RACC_HIP_OK; }
int racc_hip_intersect_streams(racc_hip_ctx* ctx, const racc_hip_scene* scene, const racc_hip_env*, uint32_t n_streams, const void* const* rays,
                               void* const* results, const uint32_t* counts, uint32_t) {
    const uint64_t k = ctx->launches.fetch_add(1);
    std::this_thread::sleep_for(std::chrono::microseconds(50 + hash32(uint32_t(k)) % 350));
    for (uint32_t s = 0; s < n_streams; ++s) {
        const uint32_t* in = static_cast<const uint32_t*>(rays[s]);
        uint32_t* out = static_cast<uint32_t*>(results[s]);
        for (uint32_t i = 0; i < counts[s]; ++i) {
            uint32_t h = 0;
            for (int w = 0; w < 8; ++w) h = hash32(h ^ in[size_t(i) * 8 + w]);
            const float one = 1.0f, third = 0.25f;
            uint32_t rec[4] = { (h % 3u) ? (h >> 4) % scene->triangles : 0xFFFFFFFFu, 0, 0, 0 };
            std::memcpy(&rec[1], &one, 4); std::memcpy(&rec[2], &third, 4); std::memcpy(&rec[3], &third, 4);
            std::memcpy(out + size_t(i) * 4, rec, 16);
        }
    }
    return RACC_HIP_OK;
}
// the scene "build": one 64 B node, one 48 B pair, a remap that names every triangle once (what racc_hip_scene_upload above reads)
int racc_host_scene_build(const float*, uint32_t, const uint32_t*, uint32_t index_count, racc_host_scene** out) {
    racc_host_scene* h = new racc_host_scene();
    h->triangles = index_count / 3;
    h->nodes.assign(64, 0); h->pairs.assign(48, 0);
    h->remap.resize(size_t(h->triangles) + (h->triangles & 1u));
    for (uint32_t i = 0; i < h->remap.size(); ++i) h->remap[i] = i < h->triangles ? i : 0u;
    *out = h;
    return RACC_HIP_OK;
}
int racc_host_scene_free(racc_host_scene* h) { delete h; return RACC_HIP_OK; }
int racc_host_scene_blobs(const racc_host_scene* h, const void** nodes64, uint32_t* node_count, const void** pairs48, uint32_t* pair_count_padded,
                          uint32_t* pair_count, const uint32_t** remap, uint32_t* remap_count) {
    *nodes64 = h->nodes.data(); *node_count = 1; *pairs48 = h->pairs.data(); *pair_count_padded = 1; *pair_count = 1;
    *remap = h->remap.data(); *remap_count = uint32_t(h->remap.size());
    return RACC_HIP_OK;
}
}
