// The host scene build at the format's pair limit (rayaccel_amd/csrc/scene_build.cpp), with the limit lowered to 2,000 pairs so that a test
// reaches it: RACC_SCENE_MAX_PAIRS is the 2^24 of a leaf reference's pair index (Scene.cpp:294-312).  Unconnected triangles: every triangle
// is a pair of its own.  1,900 triangles fit as they are but not with the default 10 % of extra references (spatial splits): the build must
// fall back to one reference per triangle and succeed with the very blobs a build without splits gives; 2,100 triangles fit neither way.
// Built and run by tests/test_host_build.py (g++, no GPU code).
#define RACC_SCENE_MAX_PAIRS 2000u
#include "../../rayaccel_amd/csrc/scene_build.cpp"

#include <cstdio>

static uint32_t rnd(uint32_t& s) { s = s * 747796405u + 2891336453u; uint32_t w = ((s >> ((s >> 28) + 4)) ^ s) * 277803737u; return (w >> 22) ^ w; }

static int build(uint32_t triangles, uint32_t splitPercent, bool useOptions, uint64_t* checksum, uint32_t* pairCount) {
    std::vector<float> v((size_t(triangles) * 12) + 4);
    float* vp = v.data();
    while (reinterpret_cast<uintptr_t>(vp) % 16) ++vp;
    uint32_t seed = 4321u;
    for (uint32_t t = 0; t < triangles; ++t) {
        const float c[3] = { float(rnd(seed) % 4096u) / 32.0f, float(rnd(seed) % 4096u) / 32.0f, float(rnd(seed) % 4096u) / 32.0f };
        for (int k = 0; k < 3; ++k) {
            float* p = vp + (size_t(t) * 3 + k) * 4;
            for (int a = 0; a < 3; ++a) p[a] = c[a] + float(rnd(seed) % 1024u) / 64.0f;      // triangles up to 16 units across: boxes overlap, many get cut
            p[3] = 1.0f;
        }
    }
    std::vector<uint32_t> idx(size_t(triangles) * 3);
    for (size_t i = 0; i < idx.size(); ++i) idx[i] = uint32_t(i);
    racc_host_build_options opt{};
    opt.struct_size = sizeof(opt); opt.quality = 1; opt.threads = 2; opt.split_percent = splitPercent;
    racc_host_scene* s = nullptr;
    const int rc = useOptions ? racc_host_scene_build_ex(vp, triangles * 3, idx.data(), uint32_t(idx.size()), &opt, &s)
                              : racc_host_scene_build(vp, triangles * 3, idx.data(), uint32_t(idx.size()), &s);
    if (rc != 0) return rc;
    const void *nodes, *pairs; const uint32_t* remap; uint32_t nn, np, npr, nr;
    racc_host_scene_blobs(s, &nodes, &nn, &pairs, &np, &npr, &remap, &nr);
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t bytes) { const unsigned char* c = static_cast<const unsigned char*>(p); for (size_t i = 0; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ull; } };
    mix(nodes, size_t(nn) * 64); mix(pairs, size_t(np) * 48); mix(remap, size_t(nr) * 4);
    *checksum = h; *pairCount = npr;
    racc_host_scene_free(s);
    return 0;
}

int main() {
    uint64_t a = 0, b = 0, c = 0; uint32_t pa = 0, pb = 0, pc = 0;
    int rc = build(1500, 0, true, &a, &pa);                                  // room for the splits: more pairs than triangles
    std::printf("1500 triangles, library's budget: rc %d, %u pairs\n", rc, pa);
    if (rc != 0 || pa <= 1500 || pa >= 2000) return 1;
    rc = build(1900, RACC_HOST_BUILD_NO_SPLITS, true, &a, &pa);
    std::printf("1900 triangles, no splits: rc %d, %u pairs\n", rc, pa);
    if (rc != 0 || pa != 1900) return 2;
    rc = build(1900, 0, true, &b, &pb);                                      // the library's budget does not fit: falls back
    std::printf("1900 triangles, library's budget: rc %d, %u pairs, %s\n", rc, pb, a == b ? "the blobs of the build without splits" : "OTHER BLOBS");
    if (rc != 0 || pb != 1900 || a != b) return 3;
    rc = build(1900, 0, false, &c, &pc);                                     // ... and so does a caller without options (racc::createScene)
    if (rc != 0 || pc != 1900 || a != c) return 4;
    rc = build(1900, 25, true, &b, &pb);                                     // a budget the caller named: the same fallback
    if (rc != 0 || pb != 1900 || a != b) return 5;
    rc = build(2100, 0, true, &b, &pb);
    std::printf("2100 triangles: rc %d (%s)\n", rc, racc_hip_last_error());
    if (rc != RACC_HIP_ERR_LIMIT) return 6;
    return 0;
}
