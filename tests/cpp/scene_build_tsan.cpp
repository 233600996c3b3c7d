// ThreadSanitizer harness of the host scene build (rayaccel_amd/csrc/scene_build.cpp): the thread-pool BVH2 build and the parallel phases
// of the quality mode (TreeOptimizer: sibling subtrees optimised by different threads share the node above them — ADVICE r05).
// Built by `make -C rayaccel_amd/csrc tsan` (g++ -fsanitize=thread, no GPU code), run by tests/test_host_build.py.
// Prints a checksum of the blobs for 1 and for N threads: they must agree (the build is deterministic for any thread count).
#include "racc_hip.h"

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

static uint32_t rnd(uint32_t& s) { s = s * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28) + 4)) ^ s) * 277803737u; return (w >> 22) ^ w; }

int main(int argc, char** argv) {
    const uint32_t grid = argc > 1 ? uint32_t(std::atoi(argv[1])) : 96u;      // grid x grid quads + floating quads: ~20k triangles, a dozen parallel subtrees
    const uint32_t threads = argc > 2 ? uint32_t(std::atoi(argv[2])) : 6u;
    std::vector<float> v;
    std::vector<uint32_t> idx;
    uint32_t seed = 12345u;
    const uint32_t n = grid + 1;
    for (uint32_t i = 0; i < n; ++i)
        for (uint32_t j = 0; j < n; ++j) {
            const float h = float(rnd(seed) & 1023u) * (1.0f / 256.0f);
            v.insert(v.end(), {float(j), h, float(i), 1.0f});
        }
    for (uint32_t i = 0; i < grid; ++i)
        for (uint32_t j = 0; j < grid; ++j) {
            const uint32_t a = i * n + j, b = a + 1, c = a + n + 1, d = a + n;
            idx.insert(idx.end(), {a, c, b, a, d, c});
        }
    for (uint32_t q = 0; q < grid * 8u; ++q) {      // thin quads floating above the field: what the re-insertion moves around
        const float x = float(rnd(seed) % (grid * 16u)) / 16.0f, z = float(rnd(seed) % (grid * 16u)) / 16.0f, y = 6.0f + float(rnd(seed) & 255u) / 16.0f;
        const float w = 0.2f + float(rnd(seed) & 63u) / 16.0f;
        const uint32_t base = uint32_t(v.size() / 4);
        v.insert(v.end(), {x, y, z, 1.0f, x + w, y, z, 1.0f, x + w, y + 0.3f, z + 0.2f, 1.0f, x, y + 0.3f, z + 0.2f, 1.0f});
        idx.insert(idx.end(), {base, base + 1, base + 2, base, base + 2, base + 3});
    }
    std::vector<float> aligned(v.size() + 4);
    float* vp = aligned.data();
    while (reinterpret_cast<uintptr_t>(vp) % 16) ++vp;
    for (size_t i = 0; i < v.size(); ++i) vp[i] = v[i];
    uint64_t sums[2] = {0, 0};
    for (int run = 0; run < 2; ++run) {
        racc_host_build_options opt{};
        opt.struct_size = sizeof(opt); opt.quality = 1; opt.threads = run ? threads : 1u;
        racc_host_scene* s = nullptr;
        if (racc_host_scene_build_ex(vp, uint32_t(v.size() / 4), idx.data(), uint32_t(idx.size()), &opt, &s) != 0) { std::fprintf(stderr, "build failed: %s\n", racc_hip_last_error()); return 1; }
        const void *nodes, *pairs; const uint32_t* remap; uint32_t nn, np, npr, nr;
        racc_host_scene_blobs(s, &nodes, &nn, &pairs, &np, &npr, &remap, &nr);
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](const void* p, size_t bytes) { const unsigned char* c = static_cast<const unsigned char*>(p); for (size_t i = 0; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ull; } };
        mix(nodes, size_t(nn) * 64); mix(pairs, size_t(np) * 48); mix(remap, size_t(nr) * 4);
        sums[run] = h;
        std::printf("threads %u: %u nodes, %u pairs, checksum %016llx\n", opt.threads, nn, npr, (unsigned long long)h);
        racc_host_scene_free(s);
    }
    return sums[0] == sums[1] ? 0 : 2;
}
