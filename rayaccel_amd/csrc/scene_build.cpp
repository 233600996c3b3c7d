// scene_build.cpp — host-side scene build for the MI355X intersect engine.
//
// Produces the reference-format blobs the traversal consumes (64 B inner nodes,
// 48 B triangle pairs, pair->triangle remap) from an indexed triangle mesh:
//   full-sweep SAH BVH2        ≙ createBvh2/build, RayAccelerator/Bvh2.cpp:257-535,772-907
//   leaf triangle-pair merge   ≙ mergeTriangle,    RayAccelerator/Scene.cpp:109-181,237-261
//   inner-node flatten         ≙                   RayAccelerator/Scene.cpp:275-332
//   pair padding               ≙                   RayAccelerator/Scene.cpp:334-338
// Design differences from the reference (none change the tree it would build
// single-threaded): subtrees are built by a pool of threads as in the reference
// (Bvh2.cpp:511-535), but under provisional node ids; a final pass renumbers the
// finished tree in the order a single-threaded depth-first build allocates ids, so
// the output does not depend on thread timing (the reference's does); one
// surface-area expression everywhere (fma(dx,dy,fma(dx,dz,dy*dz)), Bvh2.cpp:339);
// exact 1/area instead of rcpss.
// No GPU code here; this file is plain C++ and is also what racc::createScene uses.

#include "racc_hip.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <sched.h>

namespace {

void set_error(const char* msg);

struct Bvh2Node {            // Bvh2.h:15-22
    uint32_t kind, parent, first, last;
    float bbMin[3]; uint32_t pad0;
    float bbMax[3]; uint32_t pad1;
};
struct GpuNode {             // Scene.cpp:73-78
    uint32_t kind, parent, first, last;
    float leftMin[3], leftMax[3], rightMin[3], rightMax[3];
};
struct TrianglePair {        // Scene.cpp:83-87
    float e1[3], e3x, e2[3], e3y, p0[3], e3z;
};
static_assert(sizeof(Bvh2Node) == 48 && sizeof(GpuNode) == 64 && sizeof(TrianglePair) == 48, "layout");

struct Box8 {                // (-min.xyzw, max.xyzw): union is a component-wise max (Bvh2.cpp:587-621)
    float v[8];
    void grow(const Box8& o) { for (int k = 0; k < 8; ++k) v[k] = o.v[k] > v[k] ? o.v[k] : v[k]; }
    float halfArea() const {
        const float dx = v[4] + v[0], dy = v[5] + v[1], dz = v[6] + v[2];
        return std::fmaf(dx, dy, std::fmaf(dx, dz, dy * dz));
    }
};

inline uint32_t float_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float bits_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

class Bvh2Builder {
public:
    Bvh2Builder(const float* vertices, const uint32_t* indices, uint32_t triangleCount, unsigned threads)
        : verts_(vertices), idx_(indices), T_(triangleCount) { threads_ = threads ? threads : 1; }

    void run(std::vector<Bvh2Node>& nodes, std::vector<uint32_t>& triangles) {
        boxes_.resize(T_);
        for (auto& s : sorted_) s.resize(T_);
        scratch_.resize(T_);
        for (auto& pc : prefixCost_) pc.resize(T_);
        goesLeft_.resize(T_);

        Box8 scene;
        for (float& f : scene.v) f = -std::numeric_limits<float>::infinity();
        for (uint32_t t = 0; t < T_; ++t) {
            const float* a = verts_ + size_t(idx_[3 * t + 0]) * 4;
            const float* b = verts_ + size_t(idx_[3 * t + 1]) * 4;
            const float* c = verts_ + size_t(idx_[3 * t + 2]) * 4;
            Box8& bx = boxes_[t];
            for (int k = 0; k < 4; ++k) {
                bx.v[k] = -std::min(std::min(a[k], b[k]), c[k]);
                bx.v[4 + k] = std::max(std::max(a[k], b[k]), c[k]);
            }
            scene.grow(bx);
        }
        const auto tA = std::chrono::steady_clock::now();
        sortAxes();
        const auto tB = std::chrono::steady_clock::now();

        // Build under provisional ids (children get whatever pair of slots the atomic counter hands out) ...
        std::vector<Bvh2Node> tmp(size_t(T_) * 2 + 1);
        nodes_ = tmp.data();
        Bvh2Node& root = nodes_[0];
        root = Bvh2Node{};
        root.kind = 0; root.parent = 0xFFFFFFFFu; root.first = 0; root.last = T_;
        storeBounds(root, scene);
        splits_.store(0);
        buildParallel();
        const auto tC = std::chrono::steady_clock::now();

        // ... then renumber: replay the id allocation of a single-threaded depth-first build (a split node's children get
        // the next two ids when the node is visited, the left subtree is finished before the right one).
        const uint32_t splits = splits_.load();
        nodes.assign(size_t(splits) * 2 + 1, Bvh2Node{});
        nodes[0] = tmp[0];
        std::vector<std::pair<uint32_t, uint32_t>> work;      // (provisional id, final id)
        work.emplace_back(0u, 0u);
        uint32_t next = 0;
        while (!work.empty()) {
            const uint32_t t = work.back().first, f = work.back().second;
            work.pop_back();
            if (!tmp[t].kind) continue;                       // leaf: first/last are a triangle range, nothing to relocate
            ++next;
            const uint32_t left = next * 2 - 1, right = next * 2;
            nodes[left] = tmp[tmp[t].first];  nodes[left].parent = f;
            nodes[right] = tmp[tmp[t].last];  nodes[right].parent = f;
            nodes[f].first = left; nodes[f].last = right;
            work.emplace_back(tmp[t].last, right);
            work.emplace_back(tmp[t].first, left);
        }
        triangles = sorted_[0];
        if (std::getenv("RACC_PROFILE"))
            std::fprintf(stderr, "RayAccelerator profile: bvh2 with %u threads: sort %.3f s, sweep/partition %.3f s, renumber %.3f s\n", threads_,
                         std::chrono::duration<double>(tB - tA).count(), std::chrono::duration<double>(tC - tB).count(),
                         std::chrono::duration<double>(std::chrono::steady_clock::now() - tC).count());
    }

private:
    // Order-preserving float key + stable 4-pass LSD radix (Bvh2.cpp:128-184,743-749).
    void sortAxes() {
        auto one = [this](int axis) {
            std::vector<uint64_t> keys(T_), tmp(T_);
            for (uint32_t t = 0; t < T_; ++t) {
                const float mid = (boxes_[t].v[4 + axis] - boxes_[t].v[axis]) * 0.5f;   // (min+max)/2
                uint32_t e = float_bits(mid);
                e ^= (int32_t(e) < 0) ? 0xFFFFFFFFu : 0x80000000u;
                keys[t] = (uint64_t(e) << 32) | t;
            }
            uint64_t* src = keys.data();
            uint64_t* dst = tmp.data();
            for (int pass = 0; pass < 4; ++pass) {
                const int shift = 32 + 8 * pass;
                uint32_t bucket[257] = {};
                for (uint32_t t = 0; t < T_; ++t) ++bucket[((src[t] >> shift) & 0xFF) + 1];
                for (int b = 1; b < 257; ++b) bucket[b] += bucket[b - 1];
                for (uint32_t t = 0; t < T_; ++t) dst[bucket[(src[t] >> shift) & 0xFF]++] = src[t];
                std::swap(src, dst);
            }
            for (uint32_t t = 0; t < T_; ++t) sorted_[axis][t] = uint32_t(src[t]);
        };
        if (threads_ > 1 && T_ > 4096) {
            std::thread t1(one, 1), t2(one, 2);
            one(0);
            t1.join(); t2.join();
        } else {
            for (int axis = 0; axis < 3; ++axis) one(axis);
        }
    }

    // Task pool (≙ the reference's ThreadPool-driven recursion, Bvh2.cpp:511-535): a node with many triangles is one task
    // whose children become new tasks; a node below the cutoff is finished depth-first by the thread that took it.
    void buildParallel() {
        constexpr uint32_t kCutoff = 8192;
        std::mutex m;
        std::condition_variable cv;
        std::vector<uint32_t> queue;
        uint32_t running = 0;
        queue.push_back(0);
        auto worker = [&]() {
            std::vector<uint32_t> local;
            std::unique_lock<std::mutex> lock(m);
            for (;;) {
                while (queue.empty() && running != 0) cv.wait(lock);
                if (queue.empty()) { cv.notify_all(); return; }
                const uint32_t n = queue.back();
                queue.pop_back();
                ++running;
                lock.unlock();
                const uint32_t count = nodes_[n].last - nodes_[n].first;
                uint32_t kids[2]; int nk = 0;
                if (count > kCutoff) {
                    if (splitNode(n)) { kids[0] = nodes_[n].first; kids[1] = nodes_[n].last; nk = 2; }
                } else {
                    local.assign(1, n);
                    while (!local.empty()) {
                        const uint32_t c = local.back();
                        local.pop_back();
                        if (splitNode(c)) { local.push_back(nodes_[c].last); local.push_back(nodes_[c].first); }
                    }
                }
                lock.lock();
                for (int k = 0; k < nk; ++k) queue.push_back(kids[k]);
                --running;
                cv.notify_all();
            }
        };
        std::vector<std::thread> pool;
        for (unsigned i = 1; i < threads_; ++i) pool.emplace_back(worker);
        worker();
        for (std::thread& t : pool) t.join();
    }

    static void storeBounds(Bvh2Node& n, const Box8& b) {
        n.bbMin[0] = -b.v[0]; n.bbMin[1] = -b.v[1]; n.bbMin[2] = -b.v[2]; n.pad0 = float_bits(-b.v[3]);
        n.bbMax[0] = b.v[4];  n.bbMax[1] = b.v[5];  n.bbMax[2] = b.v[6];  n.pad1 = float_bits(b.v[7]);
    }

    Box8 unionOf(const std::vector<uint32_t>& order, uint32_t first, uint32_t last) const {
        Box8 b = boxes_[order[first]];
        for (uint32_t i = first + 1; i < last; ++i) b.grow(boxes_[order[i]]);
        return b;
    }

    // Stable partition of one axis list by goesLeft_ (Bvh2.cpp:217-240).
    void partitionAxis(int axis, uint32_t first, uint32_t last) {
        uint32_t* order = sorted_[axis].data();
        uint32_t l = first, r = first;           // scratch_[first..last) belongs to this node: tasks never share a slice
        for (uint32_t i = first; i < last; ++i) {
            const uint32_t t = order[i];
            if (goesLeft_[t]) order[l++] = t; else scratch_[r++] = t;
        }
        std::copy(scratch_.begin() + first, scratch_.begin() + r, order + l);
    }

    // Returns true if the node became an inner node (Bvh2.cpp:257-509).
    bool splitNode(uint32_t nodeIndex) {
        Bvh2Node& node = nodes_[nodeIndex];
        const uint32_t first = node.first, last = node.last, count = last - first;
        Box8 bounds;
        if (nodeIndex != 0) {
            bounds = unionOf(sorted_[0], first, last);
            storeBounds(node, bounds);
        } else {
            bounds.v[0] = -node.bbMin[0]; bounds.v[1] = -node.bbMin[1]; bounds.v[2] = -node.bbMin[2]; bounds.v[3] = 0;
            bounds.v[4] = node.bbMax[0];  bounds.v[5] = node.bbMax[1];  bounds.v[6] = node.bbMax[2];  bounds.v[7] = 0;
        }
        if (count <= 2) return false;                                   // Bvh2.cpp:272

        const float parentArea = bounds.halfArea();
        int axis = -1;
        uint32_t pivot = 0;
        bool forceMedian = false;

        if (parentArea > 0.0f) {
            float best = std::numeric_limits<float>::infinity();
            // One axis of the full sweep: the best pivot on `dim` that beats `bestIn`, or none (Bvh2.cpp:326-445).
            auto sweep = [&](int dim, float bestIn, float& bestOut, uint32_t& found) {
                const uint32_t* order = sorted_[dim].data();
                float* prefix = prefixCost_[dim].data();
                float bestAxis = bestIn;
                // prefix sweep: cost of [first..i] on the left; stop once it alone exceeds the best so far
                Box8 b = boxes_[order[first]];
                uint32_t i = first;
                for (; i + 1 < last; ++i) {
                    b.grow(boxes_[order[i]]);
                    prefix[i] = b.halfArea() * float(int(i - first + 1));
                    if (prefix[i] > bestAxis) break;                    // Bvh2.cpp:346-351
                }
                // suffix sweep from the stop point; pivot p splits [first,p) | [p,last)
                Box8 sfx = unionOf(sorted_[dim], i, last);
                found = 0xFFFFFFFFu;
                for (uint32_t p = i; p > first; --p) {
                    sfx.grow(boxes_[order[p]]);
                    const float right = sfx.halfArea() * float(int(last - p));
                    const float sah = prefix[p - 1] + right;
                    if (sah < bestAxis) { bestAxis = sah; found = p; }
                    if (right > bestAxis) break;                        // Bvh2.cpp:418,431-432
                }
                bestOut = bestAxis;
            };
            {
                for (int dim = 0; dim < 3; ++dim) {
                    float bo; uint32_t found;
                    sweep(dim, best, bo, found);
                    if (found != 0xFFFFFFFFu) { best = bo; pivot = found; axis = dim; }
                }
            }
            const float cost = 2.0f + 1.0f * (1.0f / parentArea) * best;    // Bvh2.cpp:462-465
            if (cost > float(int(count)) * 1.0f) forceMedian = true;
        } else {
            forceMedian = true;
        }
        if (axis < 0) forceMedian = true;          // no pivot won (only possible with non-finite costs): never index sorted_[-1]
        if (forceMedian) {                                               // Bvh2.cpp:467-485
            if (count < 127) return false;
            axis = 0;
            pivot = (first + last) >> 1;
        }

        const uint32_t* ref = sorted_[axis].data();                      // Bvh2.cpp:242-253
        for (uint32_t i = first; i < pivot; ++i) goesLeft_[ref[i]] = 1;
        for (uint32_t i = pivot; i < last; ++i) goesLeft_[ref[i]] = 0;
        partitionAxis((axis + 1) % 3, first, last);
        partitionAxis((axis + 2) % 3, first, last);

        const uint32_t slot = splits_.fetch_add(1) + 1;                  // Bvh2.cpp:489-509 (provisional ids, see run())
        const uint32_t left = slot * 2 - 1, right = slot * 2;
        node.kind = uint32_t(axis) + 1; node.first = left; node.last = right;
        Bvh2Node& l = nodes_[left];
        Bvh2Node& r = nodes_[right];
        l = Bvh2Node{}; r = Bvh2Node{};
        l.parent = nodeIndex; l.first = first; l.last = pivot;
        r.parent = nodeIndex; r.first = pivot; r.last = last;
        return true;
    }

    const float* verts_;
    const uint32_t* idx_;
    uint32_t T_;
    Bvh2Node* nodes_ = nullptr;
    std::atomic<uint32_t> splits_{0};
    unsigned threads_ = 1;
    std::vector<Box8> boxes_;
    std::vector<uint32_t> sorted_[3];
    std::vector<uint32_t> scratch_;
    std::vector<float> prefixCost_[3];
    std::vector<uint8_t> goesLeft_;
};

// A reversed shared edge between two index triples (Scene.cpp:109-120).
bool sharedEdge(const uint32_t* a, const uint32_t* b, unsigned& ea, unsigned& eb) {
    for (ea = 0; ea < 3; ++ea)
        for (eb = 0; eb < 3; ++eb)
            if (a[ea] == b[(eb + 1) % 3] && a[(ea + 1) % 3] == b[eb]) return true;
    return false;
}

TrianglePair packPair(const float* p0, const float* p1, const float* p2, const float* p3) {
    TrianglePair q;
    for (int k = 0; k < 3; ++k) { q.e1[k] = p0[k] - p1[k]; q.e2[k] = p2[k] - p0[k]; q.p0[k] = p0[k]; }
    q.e3x = p3[0] - p0[0]; q.e3y = p3[1] - p0[1]; q.e3z = p3[2] - p0[2];
    return q;
}

}  // namespace

namespace {
// Threads for the build: RACC_BUILD_THREADS, else the CPUs this process may use (affinity mask capped by the cgroup quota).
unsigned buildThreads() {
    if (const char* e = std::getenv("RACC_BUILD_THREADS")) { const long v = std::atol(e); if (v > 0) return unsigned(v > 256 ? 256 : v); }
    unsigned n = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = unsigned(CPU_COUNT(&set));
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long long quota = 0, period = 0;
        if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
            const unsigned q = unsigned((quota + period - 1) / period);
            if (q && q < n) n = q;
        }
        std::fclose(f);
    }
    return n ? (n > 64 ? 64 : n) : 1u;
}
}  // namespace

struct racc_host_scene {
    std::vector<Bvh2Node> bvh;
    std::vector<uint32_t> triangles;
    std::vector<GpuNode> nodes;
    std::vector<TrianglePair> pairs;   // padded
    std::vector<uint32_t> remap;
    uint32_t pairCount = 0;            // unpadded
    uint32_t triangleCount = 0;
};

namespace {

int flatten(racc_host_scene& s, const float* vertices, const uint32_t* indices) {
    const uint32_t nodeCount = uint32_t(s.bvh.size());
    if (!s.bvh[0].kind) { set_error("scene needs an inner root (at least 3 triangles), Kernels.h:164"); return RACC_HIP_ERR_LIMIT; }
    if (s.triangleCount >= (1u << 30)) { set_error("triangle ids must be < 2^30 (Scene.cpp:132-133)"); return RACC_HIP_ERR_LIMIT; }

    // leaf ranges in pair units, filled while merging
    std::vector<uint32_t> leafFirst(nodeCount, 0), leafLast(nodeCount, 0);
    s.pairs.clear(); s.pairs.reserve(s.triangleCount / 2 + 64);
    s.remap.clear(); s.remap.reserve(s.triangleCount + 64);
    std::vector<uint32_t> pool;
    for (uint32_t n = 0; n < nodeCount; ++n) {
        const Bvh2Node& node = s.bvh[n];
        if (node.kind) continue;
        if (node.last - node.first > 127) { set_error("leaf with more than 127 triangles (Scene.cpp:298)"); return RACC_HIP_ERR_LIMIT; }
        pool.assign(s.triangles.begin() + node.first, s.triangles.begin() + node.last);
        leafFirst[n] = uint32_t(s.pairs.size());
        while (!pool.empty()) {                                         // Scene.cpp:251-256
            const uint32_t t0 = pool.front();
            pool.erase(pool.begin());
            const uint32_t* a = indices + size_t(t0) * 3;
            bool merged = false;
            for (size_t c = 0; c < pool.size(); ++c) {
                const uint32_t* b = indices + size_t(pool[c]) * 3;
                unsigned ea, eb;
                if (!sharedEdge(a, b, ea, eb)) continue;
                s.remap.push_back(t0 | (ea << 30));
                s.remap.push_back(pool[c] | ((eb + 1) << 30));
                s.pairs.push_back(packPair(vertices + size_t(a[ea]) * 4, vertices + size_t(a[(ea + 1) % 3]) * 4,
                                           vertices + size_t(a[(ea + 2) % 3]) * 4, vertices + size_t(b[(eb + 2) % 3]) * 4));
                pool.erase(pool.begin() + c);
                merged = true;
                break;
            }
            if (!merged) {                                              // Scene.cpp:160-180: p3 = p1 => n2 = 0
                s.remap.push_back(t0);
                s.remap.push_back(0);
                const float* p1 = vertices + size_t(a[1]) * 4;
                s.pairs.push_back(packPair(vertices + size_t(a[0]) * 4, p1, vertices + size_t(a[2]) * 4, p1));
            }
        }
        leafLast[n] = uint32_t(s.pairs.size());
    }
    s.pairCount = uint32_t(s.pairs.size());
    if (s.pairCount >= (1u << 24)) { set_error("more than 2^24 triangle pairs (Scene.cpp:298)"); return RACC_HIP_ERR_LIMIT; }

    std::vector<uint32_t> innerIndex(nodeCount, 0);
    s.nodes.clear(); s.nodes.reserve(nodeCount / 2 + 1);
    auto childRef = [&](uint32_t child) -> uint32_t {
        if (s.bvh[child].kind) return child | 0x80000000u;              // patched below
        return ((leafLast[child] - leafFirst[child]) << 24) | leafFirst[child];
    };
    for (uint32_t n = 0; n < nodeCount; ++n) {
        const Bvh2Node& node = s.bvh[n];
        if (!node.kind) continue;
        innerIndex[n] = uint32_t(s.nodes.size());
        GpuNode g;
        g.kind = node.kind; g.parent = node.parent;
        g.first = childRef(node.first); g.last = childRef(node.last);
        const Bvh2Node& l = s.bvh[node.first];
        const Bvh2Node& r = s.bvh[node.last];
        for (int k = 0; k < 3; ++k) {
            g.leftMin[k] = l.bbMin[k]; g.leftMax[k] = l.bbMax[k];
            g.rightMin[k] = r.bbMin[k]; g.rightMax[k] = r.bbMax[k];
        }
        s.nodes.push_back(g);
    }
    for (GpuNode& g : s.nodes) {
        if (g.first & 0x80000000u) g.first = 0x80000000u | innerIndex[g.first & 0x7FFFFFFFu];
        if (g.last & 0x80000000u) g.last = 0x80000000u | innerIndex[g.last & 0x7FFFFFFFu];
    }
    do { s.pairs.push_back(s.pairs[0]); } while ((s.pairs.size() * 3) % 32 != 0);   // Scene.cpp:334-338
    return RACC_HIP_OK;
}

}  // namespace

extern "C" {

int racc_host_scene_build(const float* vertices, uint32_t vertex_count,
                          const uint32_t* indices, uint32_t index_count,
                          racc_host_scene** out) {
    if (!out) { set_error("out is NULL"); return RACC_HIP_ERR_INVALID; }
    *out = nullptr;
    if (!vertices || !indices) { set_error("vertices/indices is NULL"); return RACC_HIP_ERR_INVALID; }
    if (index_count % 3 != 0) { set_error("index_count must be a multiple of 3 (Scene.cpp:186)"); return RACC_HIP_ERR_INVALID; }
    if (reinterpret_cast<uintptr_t>(vertices) % 16 != 0) { set_error("vertices must be 16-byte aligned (Scene.cpp:187)"); return RACC_HIP_ERR_INVALID; }
    const uint32_t T = index_count / 3;
    if (T < 3) { set_error("scene needs at least 3 triangles (root must be an inner node)"); return RACC_HIP_ERR_LIMIT; }
    for (uint32_t i = 0; i < index_count; ++i)
        if (indices[i] >= vertex_count) { set_error("vertex index out of range"); return RACC_HIP_ERR_INVALID; }
    // The reference assumes finite geometry; a NaN/inf coordinate (or extents whose surface area overflows float) turns the
    // SAH costs into NaN and leaves the sweep without a split axis.  Refuse it here instead of building garbage.
    for (uint32_t i = 0; i < index_count; ++i) {
        const float* v = vertices + size_t(indices[i]) * 4;
        if (!(std::fabs(v[0]) < 1e18f && std::fabs(v[1]) < 1e18f && std::fabs(v[2]) < 1e18f)) {
            set_error("vertex coordinate is not finite (or beyond 1e18: surface areas would overflow binary32)");
            return RACC_HIP_ERR_INVALID;
        }
    }
    try {
        racc_host_scene* s = new racc_host_scene();
        s->triangleCount = T;
        const bool prof = std::getenv("RACC_PROFILE") != nullptr;
        const auto t0 = std::chrono::steady_clock::now();
        Bvh2Builder(vertices, indices, T, buildThreads()).run(s->bvh, s->triangles);
        const auto t1 = std::chrono::steady_clock::now();
        const int rc = flatten(*s, vertices, indices);
        if (prof) std::fprintf(stderr, "RayAccelerator profile: scene build %u triangles: bvh2 %.3f s, pack+flatten %.3f s\n", T,
                               std::chrono::duration<double>(t1 - t0).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - t1).count());
        if (rc != RACC_HIP_OK) { delete s; return rc; }
        *out = s;
        return RACC_HIP_OK;
    } catch (const std::bad_alloc&) {
        set_error("RayAccelerator: Unable to allocate memory.");
        return RACC_HIP_ERR_NOMEM;
    }
}

int racc_host_scene_free(racc_host_scene* scene) {
    delete scene;
    return RACC_HIP_OK;
}

int racc_host_scene_blobs(const racc_host_scene* s,
                          const void** nodes64, uint32_t* node_count,
                          const void** pairs48, uint32_t* pair_count_padded, uint32_t* pair_count,
                          const uint32_t** remap, uint32_t* remap_count) {
    if (!s) { set_error("scene is NULL"); return RACC_HIP_ERR_INVALID; }
    if (nodes64) *nodes64 = s->nodes.data();
    if (node_count) *node_count = uint32_t(s->nodes.size());
    if (pairs48) *pairs48 = s->pairs.data();
    if (pair_count_padded) *pair_count_padded = uint32_t(s->pairs.size());
    if (pair_count) *pair_count = s->pairCount;
    if (remap) *remap = s->remap.data();
    if (remap_count) *remap_count = uint32_t(s->remap.size());
    return RACC_HIP_OK;
}

int racc_host_scene_bvh2(const racc_host_scene* s,
                         const void** nodes48, uint32_t* node_count,
                         const uint32_t** triangles, uint32_t* triangle_count) {
    if (!s) { set_error("scene is NULL"); return RACC_HIP_ERR_INVALID; }
    if (nodes48) *nodes48 = s->bvh.data();
    if (node_count) *node_count = uint32_t(s->bvh.size());
    if (triangles) *triangles = s->triangles.data();
    if (triangle_count) *triangle_count = s->triangleCount;
    return RACC_HIP_OK;
}

}  // extern "C"

// ---- thread-local error text shared by the whole library -------------------------------------
namespace {
thread_local char g_error[512] = "";
void set_error(const char* msg) {
    std::strncpy(g_error, msg ? msg : "", sizeof(g_error) - 1);
    g_error[sizeof(g_error) - 1] = 0;
}
}  // namespace

extern "C" const char* racc_hip_last_error(void) { return g_error; }
extern "C" void racc_hip_set_error_(const char* msg) { set_error(msg); }   // used by racc_hip.hip
