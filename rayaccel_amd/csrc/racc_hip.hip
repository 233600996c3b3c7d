// racc_hip.hip — gfx950 (MI355X) wavefront BVH2 traversal + the C-ABI around it.
//
// Replaces the reference's OpenCL `traversal` kernel (RayAccelerator/Kernels.h:139-242) and its launch path
// (RayAccelerator/RayAccelerator.cpp:378-404).  Not a translation: the reference runs one work-item per ray in
// work-groups of 8 with a private int[64] stack and one blocking launch per <=27k-ray stream.  Here:
//   * persistent waves (wave64) pull rays from a global cursor in chunks and keep their lanes full: finished lanes are
//     detected with a wave ballot, ranked with mbcnt (prefix sum over the ballot) and re-loaded with the next rays of
//     the wave's chunk — active-ray compaction at wavefront width;
//   * each iteration the wave VOTES on what to run: an inner-node step for every lane that holds an inner node, or a
//     triangle-pair step for every lane that holds a leaf ("vote-scheduled while-while"); Moller-Trumbore (the
//     reference's Embree-style pair test, Kernels.h:36-115) is fused into that leaf step;
//   * divergent waves fetch their 64 B node records quad-cooperatively through LDS-DMA (four lanes read one record's 64
//     contiguous bytes); the per-ray traversal stack lives in LDS as [level][thread] (12 entries, conflict-free), deeper
//     entries in a global spill — unlike the reference's unchecked stack[64] it cannot overflow;
//   * hit epilogues (remap gather + barycentric rotation) are batched into the refill step; miss radiance is evaluated by
//     a second, streaming kernel (envShadeKernel);
//   * launches of different lanes (HIP stream + ray cursor + spill area each) overlap: one's drain runs beside the next
//     one's bulk.
// The shipped kernel, traverseKernelV8 (hot loop in hand-scheduled assembly), is in racc_kernel_v8.inc; the seven earlier
// generations are in tools/experimental/racc_kernels_experimental.inc and exist only in a `make EXPERIMENTAL=1` build.
// Arithmetic is IEEE binary32 with explicit fmaf only (built with -ffp-contract=off, no fast-math), the same evaluation
// order as oracle/racc_oracle.c, so primId/t/u/v are bit-identical to the CPU restatement for every finite ray.  The
// traversal ORDER is the reference's (nearer child first, far child pushed only if both hit, pairs of a leaf in order),
// which is what makes ties resolve identically.

#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <atomic>
#include <mutex>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <limits>
#include <new>
#include <queue>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "racc_hip.h"

extern "C" void racc_hip_set_error_(const char* msg);

namespace {

#include "racc_device.inc"

#include "racc_kernel_v8.inc"

#include "racc_kernel_v9.inc"

#include "racc_kernel_v10.inc"

#ifdef RACC_EXPERIMENTAL
#include "racc_kernels_experimental.inc"      // V1..V7: earlier generations and ablations, tools/experimental/, `make EXPERIMENTAL=1`
#endif

// ------------------------------------------------------------------------------------------ host side

// Second (tiny, streaming) kernel of the V2 path: every miss record holds the ray direction; replace it by the
// probe-image radiance (Kernels.h:213-222).  16 B read per ray, 16 B written per miss.
__global__ void __launch_bounds__(256) envShadeKernel(float4* results, uint32_t count, const float4* env, uint32_t envW, uint32_t envH) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
        const float4 rec = results[i];
        if (__float_as_uint(rec.x) != kInvalidTriangle) continue;
        results[i] = isnan(rec.y) ? make_float4(rec.x, 0.0f, 0.0f, 0.0f) : envSample(env, envW, envH, rec.y, rec.z, rec.w);
    }
}

// Chained launches: writes launch `idx`'s descriptor into the ring (device memory: a drained wave reads it at L2 speed; host-mapped
// memory cost every such wave six PCIe round trips) and links it behind its predecessor.  One thread, on the context's control stream.
__global__ void chainPublishKernel(ChainDesc* ring, uint32_t idx, const float4* rays, float4* results, uint32_t* cursor, uint32_t count,
                                   uint32_t dynBase, int pred) {
    ChainDesc& d = ring[idx];
    d.rays = rays; d.results = results; d.cursor = cursor; d.count = count; d.dynBase = dynBase;
    __hip_atomic_store(&d.next, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (pred >= 0) __hip_atomic_store(&ring[pred].next, idx + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

thread_local char g_msg[512];

int fail(int code, const char* what, hipError_t e = hipSuccess) {
    if (e != hipSuccess) snprintf(g_msg, sizeof(g_msg), "racc_hip: %s: %s", what, hipGetErrorString(e));
    else snprintf(g_msg, sizeof(g_msg), "racc_hip: %s", what);
    racc_hip_set_error_(g_msg);
    return code;
}

#define HIP_TRY(call, what)                                                      \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) return fail(RACC_HIP_ERR_DEVICE, what, e_);        \
    } while (0)

struct Lane {
    hipStream_t stream = nullptr;
    uint32_t* cursor = nullptr;          // 2 words, zero-initialised; the kernel re-arms it
    uint32_t* spill = nullptr;
    size_t spillWords = 0;
    void* dRays = nullptr;               // staging for the host-buffer path
    void* dResults = nullptr;
    uint32_t capacity = 0;
    std::vector<hipEvent_t> events;
    hipStream_t copyIn = nullptr, copyOut = nullptr;   // host-buffer path: copies of slice k+1 / k-1 run beside the kernel of slice k
    std::vector<hipEvent_t> pipeEvents;
    racc_hip_launch_info info{};
    bool pendingEnv = false;             // last traversal launch parked miss directions: envShade must follow
    // A lane owns ONE ray cursor / ticket / spill area, so its launches must not overlap: every launch is followed by
    // `done` on the stream it went to, and a launch that goes to a different stream than the previous one waits for it.
    std::mutex mutex;                    // host-side: one thread at a time enqueues on a lane
    hipEvent_t done = nullptr;
    hipStream_t lastStream = nullptr;
    std::atomic<bool> everLaunched{false};
    // opts.time_kernels: an event pair around every traversal kernel, read back by racc_hip_read_kernel_times
    std::vector<hipEvent_t> ring;        // 2 * kTimeRing events
    uint32_t ringHead = 0, ringCount = 0;
    // host-buffer path, sliced: the kernels of consecutive slices go to the lane and its two helpers in turn, so that one
    // slice's drain runs beside the next slice's bulk (each with half a grid), created on first use
    Lane* helper[3] = {nullptr, nullptr, nullptr};      // (the third only when the runtime has >= 8 hardware queues)
    uint32_t forceWavesPerSimd = 0;      // != 0: grid size of this (helper-rotated) launch
    hipEvent_t chainKernelEnd = nullptr; // chained launches: recorded right after the lane's latest chained traversal kernel ...
    hipEvent_t chainKernelEndEv = nullptr;   // ... or, with time_kernels, that launch's timing end event (not owned)
    bool chainKernelValid = false;
};
constexpr uint32_t kTimeRing = 256;

}  // namespace

struct racc_hip_ctx {
    int device = 0;
    int numCUs = 0;
    racc_hip_options opts{};
    Lane lanes[RACC_HIP_MAX_LANES];
    uint32_t* hostTrips = nullptr;       // host-mapped: kernels bump it when a wave hits the iteration limit
    uint32_t* devTrips = nullptr;        // the device alias of the same word
    std::atomic<uint32_t> seenTrips{0};
    std::atomic<uint32_t> nextLane{0};   // RACC_HIP_LANE_AUTO: round robin
    uint32_t autoLanes = 3;              // ... over this many lanes
    uint32_t overlapWaves = 2;           // waves per SIMD of a launch that finds another lane's launch running
    int hwQueues = 4;                    // GPU_MAX_HW_QUEUES as this process's HIP runtime was started with
    // chained launches (launchTraverse): a ring of descriptors in host-mapped memory, one fresh cursor word per launch
    static constexpr uint32_t kChainRing = 256;
    ChainDesc* chainDev = nullptr;       // the ring of descriptors (device memory, written by chainPublishKernel)
    hipStream_t chainStream = nullptr;   // control stream of the publish kernels
    uint32_t* chainCursors = nullptr;    // device: kChainRing x 16 words (cursor at word 0), all zero between ring laps
    uint32_t chainHead = 0;              // launches so far
    std::mutex chainMutex;
    struct { const racc_hip_scene* scene = nullptr; const racc_hip_env* env = nullptr; const void* kernel = nullptr; uint32_t idx = 0; Lane* lane = nullptr; bool valid = false; } chainLast;
    bool chainEnabled = true;            // RACC_CHAIN=0 switches it off
    bool raysBypassL1 = true;            // chained kernels load rays with system-scope loads (RACC_RAY_SCOPE=0: plain loads, A/B only)
    uint32_t maxIters = 1u << 24;        // RACC_MAX_ITERS overrides (tests)
};

struct racc_hip_scene {
    float4* nodes = nullptr;
    float4* nodesWide = nullptr;    // the same tree collapsed into 4-wide 128 B records (collapseWide)
    float4* nodesWideQ = nullptr;   // ... and those compressed to 64 B: child boxes quantised to 8 bits per plane on the box around them (quantiseWide)
    uint32_t wideCount = 0;
    uint32_t wideStack = 0;         // upper bound of a ray's stack entries in the wide tree
    float4* nodesSoa = nullptr;     // only when the context asks for the SoA ablation variant
    float4* pairs = nullptr;
    uint32_t* remap = nullptr;
    racc_hip_scene_info info{};
};

struct racc_hip_env {
    float4* pixels = nullptr;
    uint32_t width = 0, height = 0;
};

namespace {

struct GpuNodeHost { uint32_t kind, parent, first, last; float box[12]; };

// Walks the blob from the root: every child index / pair range must be in bounds and every inner node
// reachable at most once (a DAG or cycle would make traversal unbounded).  Also measures the height.
int validateScene(const GpuNodeHost* nodes, uint32_t nodeCount, uint32_t pairCount, uint32_t remapCount,
                  racc_hip_scene_info& info) {
    if (!nodeCount) return fail(RACC_HIP_ERR_LIMIT, "scene has no inner node (needs >= 3 triangles; root must be inner, Kernels.h:164)");
    if (nodeCount >= 0x7FFFFFFFu || pairCount > (1u << 24)) return fail(RACC_HIP_ERR_LIMIT, "node/pair count exceeds the reference format (Scene.cpp:294-312)");
    // the kernel addresses a node record by a 32-bit byte offset (index << 6): 2^26 inner nodes = 4 GiB of records, i.e. scenes
    // of ~190 M triangles; the reference format's own limit on pairs (2^24, above) is reached long before
    if (nodeCount > (1u << 26)) return fail(RACC_HIP_ERR_LIMIT, "more than 2^26 inner nodes: beyond the 32-bit record offsets of this kernel");
    std::vector<uint8_t> seen(nodeCount, 0);
    std::vector<std::pair<uint32_t, uint32_t>> work;   // (node, depth)
    work.emplace_back(0u, 1u);
    seen[0] = 1;
    uint32_t height = 0, maxLeaf = 0;
    while (!work.empty()) {
        const auto [n, depth] = work.back();
        work.pop_back();
        height = depth > height ? depth : height;
        const uint32_t kids[2] = { nodes[n].first, nodes[n].last };
        for (uint32_t c : kids) {
            if (c & 0x80000000u) {
                const uint32_t ci = c & 0x7FFFFFFFu;
                if (ci >= nodeCount) return fail(RACC_HIP_ERR_INVALID, "scene blob: child index out of range");
                if (seen[ci]) return fail(RACC_HIP_ERR_INVALID, "scene blob: inner node referenced twice (not a tree)");
                seen[ci] = 1;
                work.emplace_back(ci, depth + 1);
            } else {
                const uint32_t first = c & 0xFFFFFFu, cnt = c >> 24;
                if (cnt == 0) return fail(RACC_HIP_ERR_INVALID, "scene blob: leaf with zero pairs");
                if (first + cnt > pairCount) return fail(RACC_HIP_ERR_INVALID, "scene blob: leaf pair range out of bounds");
                if ((first + cnt) * 2ull > remapCount) return fail(RACC_HIP_ERR_INVALID, "scene blob: remap shorter than the pairs it indexes");
                maxLeaf = cnt > maxLeaf ? cnt : maxLeaf;
            }
        }
    }
    info.inner_height = height;
    info.max_leaf_pairs = maxLeaf;
    // A push happens at most once per inner node on the current root path.
    info.spill_levels = 0;   // filled per kernel variant at launch: max(0, height - LDS_LEVELS)
    return RACC_HIP_OK;
}

// Device node order: the kCacheMax nodes with the largest own bounding-box area first (a node's box is always
// larger than its children's, so this is a connected top of the tree and every prefix [0,K) of it is the best K
// by that measure — ~50 % of all node visits for K = 1024 on battlefield-synth, vs 43 % for plain BFS levels),
// the rest in the reference's order.  Child references are rewritten; results cannot depend on node numbering.
void reorderNodes(const GpuNodeHost* in, uint32_t n, std::vector<GpuNodeHost>& out) {
    std::vector<uint32_t> newToOld;
    newToOld.reserve(n);
    std::vector<uint8_t> placed(n, 0);
    auto area = [](const float* b) {   // b = min[3], max[3]
        const double x = double(b[3]) - b[0], y = double(b[4]) - b[1], z = double(b[5]) - b[2];
        return x * y + x * z + y * z;
    };
    using Item = std::pair<double, uint32_t>;   // (area, -index) max-heap: larger area first, lower index on ties
    std::priority_queue<Item> heap;
    heap.emplace(1e300, ~0u);
    const uint32_t top = n < uint32_t(kCacheMax) ? n : uint32_t(kCacheMax);
    while (!heap.empty() && newToOld.size() < top) {
        const uint32_t node = ~heap.top().second;
        heap.pop();
        newToOld.push_back(node);
        placed[node] = 1;
        if (in[node].first & 0x80000000u) heap.emplace(area(in[node].box + 0), ~(in[node].first & 0x7FFFFFFFu));
        if (in[node].last & 0x80000000u) heap.emplace(area(in[node].box + 6), ~(in[node].last & 0x7FFFFFFFu));
    }
    for (uint32_t i = 0; i < n; ++i) if (!placed[i]) newToOld.push_back(i);
    std::vector<uint32_t> oldToNew(n);
    for (uint32_t i = 0; i < n; ++i) oldToNew[newToOld[i]] = i;
    out.resize(n);
    for (uint32_t i = 0; i < n; ++i) {
        const GpuNodeHost& g = in[newToOld[i]];
        GpuNodeHost d{};   // device record: see slabPair
        d.kind = (g.first & 0x80000000u) ? (0x80000000u | oldToNew[g.first & 0x7FFFFFFFu]) : g.first;      // word 0: first child
        d.parent = (g.last & 0x80000000u) ? (0x80000000u | oldToNew[g.last & 0x7FFFFFFFu]) : g.last;        // word 1: last child
        const float* b = g.box;   // leftMin[3], leftMax[3], rightMin[3], rightMax[3]
        const float planes[12] = { b[0], b[3], b[1], b[4], b[2], b[5], b[6], b[9], b[7], b[10], b[8], b[11] };
        std::memcpy(d.box, planes, sizeof(planes));
        out[i] = d;
    }
}

// 4-wide device format (racc_kernel_v9.inc).  The two children of a BVH2 node are the first candidates; the inner candidate
// with the largest surface area is replaced, in place (spatial neighbours stay neighbours), by its own two children until
// there are four or only leaves are left.  Boxes are copied, never recomputed: every box a ray is tested against is one
// the reference tests it against.  Nodes are numbered breadth first.  An unused slot holds a box at +inf (never entered:
// its entry distance is +inf or its exit distance -inf) and the ref of a real leaf, so that even a ray whose arithmetic
// has gone non-finite can only be sent to geometry that exists.
struct WideNode { uint32_t ref[4]; float plane[6][4]; uint32_t pad[4]; };      // planes: lo.x hi.x lo.y hi.y lo.z hi.z, four children each
static_assert(sizeof(WideNode) == 128, "one L1 line");

void collapseWide(const GpuNodeHost* in, uint32_t n, std::vector<WideNode>& out, uint32_t& stackBound) {
    struct Cand { uint32_t ref; float mn[3], mx[3]; };
    auto area = [](const Cand& c) {
        const double x = double(c.mx[0]) - c.mn[0], y = double(c.mx[1]) - c.mn[1], z = double(c.mx[2]) - c.mn[2];
        return x * y + x * z + y * z;
    };
    std::vector<uint32_t> map(n, 0xFFFFFFFFu), queue, depth;      // BVH2 index -> wide index; BFS queue of BVH2 indices; depth of each wide node
    queue.reserve(n); depth.reserve(n); out.clear(); out.reserve(n / 2 + 1);
    queue.push_back(0); depth.push_back(1); map[0] = 0;
    uint32_t height = 0;
    const uint32_t firstLeaf = (in[0].first & 0x80000000u) ? 0x01000000u : in[0].first;      // a real leaf: pair 0 always exists (validated)
    for (size_t qh = 0; qh < queue.size(); ++qh) {
        const GpuNodeHost& g = in[queue[qh]];
        height = depth[qh] > height ? depth[qh] : height;
        Cand c[4]; int k = 2;
        c[0].ref = g.first; std::memcpy(c[0].mn, g.box + 0, 12); std::memcpy(c[0].mx, g.box + 3, 12);
        c[1].ref = g.last;  std::memcpy(c[1].mn, g.box + 6, 12); std::memcpy(c[1].mx, g.box + 9, 12);
        while (k < 4) {
            int best = -1; double bestA = -1.0;
            for (int i = 0; i < k; ++i)
                if ((c[i].ref & 0x80000000u) && area(c[i]) > bestA) { bestA = area(c[i]); best = i; }
            if (best < 0) break;
            const GpuNodeHost& m = in[c[best].ref & 0x7FFFFFFFu];
            for (int i = k; i > best + 1; --i) c[i] = c[i - 1];
            c[best].ref = m.first; std::memcpy(c[best].mn, m.box + 0, 12); std::memcpy(c[best].mx, m.box + 3, 12);
            c[best + 1].ref = m.last; std::memcpy(c[best + 1].mn, m.box + 6, 12); std::memcpy(c[best + 1].mx, m.box + 9, 12);
            ++k;
        }
        WideNode w{};
        for (int i = 0; i < 4; ++i) {
            if (i < k) {
                uint32_t r = c[i].ref;
                if (r & 0x80000000u) {
                    const uint32_t t = r & 0x7FFFFFFFu;
                    map[t] = uint32_t(queue.size());
                    queue.push_back(t); depth.push_back(depth[qh] + 1);
                    r = 0x80000000u | map[t];
                }
                w.ref[i] = r;
                for (int ax = 0; ax < 3; ++ax) {      // the reference's slab test takes min/max of the two plane distances: an inverted box acts as its mirror image
                    w.plane[2 * ax][i] = c[i].mn[ax] < c[i].mx[ax] ? c[i].mn[ax] : c[i].mx[ax];
                    w.plane[2 * ax + 1][i] = c[i].mn[ax] < c[i].mx[ax] ? c[i].mx[ax] : c[i].mn[ax];
                }
            } else {
                w.ref[i] = firstLeaf;
                for (int p = 0; p < 6; ++p) w.plane[p][i] = std::numeric_limits<float>::infinity();
            }
        }
        out.push_back(w);
    }
    stackBound = 3u * height + 1u;      // at most three entries per level of the path
}

// Compressed 4-wide device format (racc_kernel_v10.inc): the record of collapseWide in 64 bytes.  Frame = the box around the
// used children: origin = its lower corner, scale = extent / 255, raised until fmaf(255, scale, origin) reaches the upper corner.
// A lower plane takes the largest byte whose decoded value — fmaf(float(byte), scale, origin), the kernel's expression, one
// rounding — does not exceed it, an upper plane the smallest byte whose decoded value is not below it: the decoded box contains
// the reference's box, checked here with that very expression.  Unused slots: lower = 255, upper = 0 on every axis (entry beyond
// exit on every axis with an extent: never entered) and the ref of a real leaf.
struct WideNodeQ { uint32_t ref[4]; float org[3]; float sclX; float sclY, sclZ; uint32_t q[6]; };      // q: lo.x hi.x lo.y hi.y lo.z hi.z, byte i = child i
static_assert(sizeof(WideNodeQ) == 64, "half an L1 line, like the binary record");

int quantiseWide(const std::vector<WideNode>& in, std::vector<WideNodeQ>& out) {
    out.resize(in.size());
    for (size_t n = 0; n < in.size(); ++n) {
        const WideNode& w = in[n];
        WideNodeQ r{};
        std::memcpy(r.ref, w.ref, sizeof(r.ref));
        bool used[4];
        for (int i = 0; i < 4; ++i) used[i] = !std::isinf(w.plane[0][i]);
        for (int i = 0; i < 4; ++i)
            for (int pl = 0; pl < 6; ++pl)
                if (used[i] && !std::isfinite(w.plane[pl][i])) return fail(RACC_HIP_ERR_INVALID, "scene blob: a child box is not finite (the compressed 4-wide format cannot hold it)");
        float scl[3];
        for (int ax = 0; ax < 3; ++ax) {
            float lo = std::numeric_limits<float>::infinity(), hi = -lo;
            for (int i = 0; i < 4; ++i) if (used[i]) { lo = std::fmin(lo, w.plane[2 * ax][i]); hi = std::fmax(hi, w.plane[2 * ax + 1][i]); }
            float scale = (hi - lo) / 255.0f;
            if (!(scale > 0.0f)) scale = std::numeric_limits<float>::min();
            while (std::fmaf(255.0f, scale, lo) < hi) scale = std::nextafter(scale, std::numeric_limits<float>::infinity());
            if (!std::isfinite(scale) || !std::isfinite(lo)) return fail(RACC_HIP_ERR_LIMIT, "scene blob: a node's extent overflows binary32");
            r.org[ax] = lo; scl[ax] = scale;
            uint32_t qlo = 0, qhi = 0;
            for (int i = 0; i < 4; ++i) {
                int a = 255, b = 0;      // unused slot: inverted
                if (used[i]) {
                    const float pl = w.plane[2 * ax][i], ph = w.plane[2 * ax + 1][i];
                    a = int(std::floor((pl - lo) / scale)); b = int(std::ceil((ph - lo) / scale));
                    a = a < 0 ? 0 : a > 255 ? 255 : a; b = b < 0 ? 0 : b > 255 ? 255 : b;
                    while (a > 0 && std::fmaf(float(a), scale, lo) > pl) --a;
                    while (a < 255 && std::fmaf(float(a + 1), scale, lo) <= pl) ++a;
                    while (b < 255 && std::fmaf(float(b), scale, lo) < ph) ++b;
                    while (b > 0 && std::fmaf(float(b - 1), scale, lo) >= ph) --b;
                    if (std::fmaf(float(a), scale, lo) > pl || std::fmaf(float(b), scale, lo) < ph) return fail(RACC_HIP_ERR_LIMIT, "scene blob: a child box cannot be quantised conservatively");
                }
                qlo |= uint32_t(a) << (8 * i); qhi |= uint32_t(b) << (8 * i);
            }
            r.q[2 * ax] = qlo; r.q[2 * ax + 1] = qhi;
        }
        r.sclX = scl[0]; r.sclY = scl[1]; r.sclZ = scl[2];
        out[n] = r;
    }
    return RACC_HIP_OK;
}

int ensureSpill(racc_hip_ctx* ctx, Lane& lane, uint32_t gridThreads, uint32_t levels) {
    const size_t words = size_t(gridThreads) * (levels ? levels : 1u);
    if (lane.spillWords >= words) return RACC_HIP_OK;
    (void)ctx;
    if (lane.spill) { HIP_TRY(hipFree(lane.spill), "hipFree(spill)"); lane.spill = nullptr; lane.spillWords = 0; }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&lane.spill), words * sizeof(uint32_t)), "hipMalloc(spill)");
    lane.spillWords = words;
    return RACC_HIP_OK;
}

uint32_t optOr(uint32_t v, uint32_t dflt) { return v ? v : dflt; }

#ifndef RACC_EXPERIMENTAL
constexpr int kRecFields = 20;     // (V3's LDS record, racc_kernels_experimental.inc: only sizes its table rows here)
#endif

struct Variant {
    int id;                    // the racc_hip_options::kernel_variant number that selects this row
    int block, ldsLevels, cacheNodes;
    void (*kernel)(const TraverseArgs);
    bool noSpill = false;      // kernel has no global spill path: only valid while tree height <= ldsLevels
    bool deferEnv = false;     // kernel parks miss directions; envShadeKernel must follow
    int slots = 1;             // ray slots per lane (V4: 2); ldsLevels counts all of them
    int reserved = 0;          // LDS levels the kernel keeps for itself (V5: the sentinel; V6: sentinel + trash level)
    int stagePerWave = 0;      // bytes of LDS-DMA stage per wave (V6 COOP)
    bool wide = false;         // traverses the 4-wide device format (V9, V10)
    void (*kernelChained)(const TraverseArgs) = nullptr;      // the instantiation whose waves can move on to the next launch of a chain (V8, V10)
    bool quant = false;        // ... its 64-byte compressed form (V10)
    int stackLevels() const { return (ldsLevels > 32 ? ldsLevels - (kRecFields + 1) : ldsLevels / slots) - reserved; }   // V3 rows fold the record words into ldsLevels
};
// kernel_variant n selects the row with id n; 0 selects the default (V8).  LDS per workgroup = cacheNodes*64 + ldsLevels*block*4
// (+ the LDS-DMA stage).  Ids 1-40 are earlier generations and ablations: they live outside the product tree
// (tools/experimental/) and exist only in a `make EXPERIMENTAL=1` build; otherwise racc_hip_create refuses those numbers.
const Variant kVariants[] = {
#ifdef RACC_EXPERIMENTAL
#include "racc_variants_experimental.inc"
#endif
    {41, 256, 10, 0, traverseKernelV8<256, 9, false>, false, true, 1, 2, 4 * 1040, false, traverseKernelV8<256, 9, false, true, true>},           // 41: V8 with an 8-entry LDS stack (exercises the DEEP door and the spill: 2.6 % of the bench rays)
    {42, 256, 14, 0, traverseKernelV8<256, 13, true>, false, true, 1, 2, 4 * 1040, false, nullptr},           // 42: variant 43 + statistics (debug; only the C++ parts count: refills, rays loaded, waves)
    {43, 256, 14, 0, traverseKernelV8<256, 13, false>, false, true, 1, 2, 4 * 1040, false, traverseKernelV8<256, 13, false, true, true>},          // 43: V8, 12-entry LDS stack: the default
    {44, 256, 14, 0, traverseKernelV8<256, 13, false, false>, false, false, 1, 2, 4 * 1040, false, traverseKernelV8<256, 13, false, false, true>},  // 44: variant 43 with the probe-image lookup in its own epilogue (no envShadeKernel launch)
    {45, 256, 19, 0, traverseKernelV9<256, 19, false, true>, false, true, 1, 1, 8 * 1040, true},    // 45: V9 (4-wide nodes, hot loop in assembly), 18-entry LDS stack: 3 workgroups per CU
    {46, 256, 7, 0, traverseKernelV9<256, 7, false, false>, false, true, 1, 1, 8 * 1040, true},     // 46: V9 in plain C++, 6-entry LDS stack + spill: 4 workgroups per CU (exercises the spill)
    {47, 256, 19, 0, traverseKernelV9<256, 19, true, false>, false, true, 1, 1, 8 * 1040, true},    // 47: V9 in plain C++ + statistics (debug)
    {48, 256, 19, 0, traverseKernelV9<256, 19, false, false>, false, true, 1, 1, 8 * 1040, true},   // 48: V9 in plain C++ (A/B of the assembly block)
    {49, 256, 8, 0, traverseKernelV9<256, 8, false, true>, false, true, 1, 1, 8 * 1040, true},      // 49: variant 45 with a 7-entry LDS stack (exercises the DEEP door and the spill)
    {50, 256, 15, 0, traverseKernelV10<256, 15, false, true>, false, true, 1, 1, 4 * 1040, true, traverseKernelV10<256, 15, false, true, true>, true},     // 50: V10 (compressed 4-wide nodes, 64 B; hot loop in assembly), 14-entry LDS stack: 5 workgroups per CU
    {51, 256, 15, 0, traverseKernelV10<256, 15, false, false>, false, true, 1, 1, 4 * 1040, true, traverseKernelV10<256, 15, false, false, true>, true},   // 51: V10 in plain C++ (A/B and cross-check of the assembly block)
    {52, 256, 15, 0, traverseKernelV10<256, 15, true, false>, false, true, 1, 1, 4 * 1040, true, nullptr, true},                                         // 52: V10 in plain C++ + statistics (debug)
    {53, 256, 8, 0, traverseKernelV10<256, 8, false, true>, false, true, 1, 1, 4 * 1040, true, traverseKernelV10<256, 8, false, true, true>, true},         // 53: variant 50 with a 7-entry LDS stack (exercises the DEEP door and the spill)
};
constexpr int kSoaVariant = 27;
constexpr int kWideVariant = 45;
constexpr int kDefaultVariant = 43;   // V8, 12-entry LDS stack + global spill (any tree height)
constexpr int kSpillFallback = kDefaultVariant;    // used when a tree is taller than an LDS-only variant's stack
constexpr uint32_t kLdsPerCU = 160u * 1024u;

const Variant* variantById(uint32_t id) {
    for (const Variant& v : kVariants) if (uint32_t(v.id) == id && v.kernel) return &v;
    return nullptr;
}

// kernel_variant 0 (default): V8 with a 12-entry LDS stack.  On battlefield-synth 99.99 % of the rays never go deeper
// (mean 4.7, max 15); the rest of any tree's height lives in the global spill.
const Variant& pickVariant(const racc_hip_ctx* ctx, uint32_t treeHeight) {
    (void)treeHeight;
    if (const Variant* v = variantById(ctx->opts.kernel_variant)) return *v;
    return *variantById(kDefaultVariant);
}

// mayChain: the batch is resident and final NOW (a device-resident batch issued on one of the engine's own streams), so waves of
// the launch issued before it may start on it before its own kernel does (DESIGN.md §3 "Chained launches").
int launchTraverse(racc_hip_ctx* ctx, Lane& lane, hipStream_t stream, const racc_hip_scene* scene, const racc_hip_env* env,
                   const void* dRays, void* dResults, uint32_t count, bool mayChain = false) {
    if (!count) return RACC_HIP_OK;
    const Variant* vp = &pickVariant(ctx, scene->info.inner_height);
    if (count < ctx->opts.wide_below && !vp->wide) vp = variantById(kWideVariant);      // small launch: the shorter dependent chain wins
    if (vp->noSpill && scene->info.inner_height > uint32_t(vp->stackLevels())) vp = variantById(kSpillFallback);   // tall tree
    const Variant& v = *vp;
    const uint32_t ldsBytes = uint32_t(v.cacheNodes) * 64u + uint32_t(v.ldsLevels) * uint32_t(v.block) * 4u + (v.ldsLevels > 32 ? 272u : 0u) +
                              uint32_t(v.stagePerWave) * uint32_t(v.block / 64);
    // Persistent grid: as many waves as the LDS allows (5 per SIMD with the default kernel) when the GPU is otherwise idle.
    // When another lane's launch is still running — a caller issuing batch after batch — a launch takes 2 per SIMD (1 with six lanes in rotation):
    // two or three launches are then co-resident, each one's drain (its last, longest rays: ~0.13 ms during which most of
    // its waves have nothing left) runs beside the others' bulk instead of leaving the machine empty.  Measured on 1M-ray
    // diffuse batches (tools/gpu_overlap.py): 0.378 ms per batch one at a time; back to back over 3 lanes 0.294 with full
    // grids, 0.271 with 3 waves per SIMD each, 0.266 with 2 (0.33 / 0.28 with four / five lanes in rotation: three it is).
    // ---- chained launches: is the launch issued just before this one (any lane) still running on the same scene?
    const bool chain = mayChain && v.kernelChained && ctx->chainEnabled && ctx->chainDev != nullptr && count < 0x80000000u;      // (bit 31 of a ray index tags the batch)
    std::unique_lock<std::mutex> chainGuard(ctx->chainMutex, std::defer_lock);
    uint32_t chainIdx = 0;
    int chainPred = -1;
    if (chain) {
        chainGuard.lock();
        chainIdx = ctx->chainHead % racc_hip_ctx::kChainRing;
        if (chainIdx == 0 && ctx->chainHead != 0) {      // a lap of the ring: every cursor word must be zero again before it is handed out
            // Every chained kernel was enqueued on a lane's own stream while chainMutex was held, so draining those streams here (the
            // mutex is held again) ends every kernel that can still look at the ring.  Not the lanes' `done` events: another host
            // thread records its lane's `done` only after it has left launchTraverse — this thread would see the previous one.
            for (uint32_t i = 0; i < ctx->opts.lanes; ++i)
                HIP_TRY(hipStreamSynchronize(ctx->lanes[i].stream), "hipStreamSynchronize(chain lap)");
            HIP_TRY(hipStreamSynchronize(ctx->chainStream), "hipStreamSynchronize(chain)");
            HIP_TRY(hipMemset(ctx->chainCursors, 0, size_t(racc_hip_ctx::kChainRing) * 64), "hipMemset(chain cursors)");
            // ... and no descriptor may keep a link of the lap before: a kernel can look at its own descriptor before the publish
            // kernel of its launch has run, and must then find "no next", not last lap's successor
            HIP_TRY(hipMemset(ctx->chainDev, 0, sizeof(ChainDesc) * racc_hip_ctx::kChainRing), "hipMemset(chain ring)");
            ctx->chainLast.valid = false;
        }
        if (ctx->chainLast.valid && ctx->chainLast.scene == scene && ctx->chainLast.env == env && ctx->chainLast.kernel == reinterpret_cast<const void*>(v.kernelChained) &&
            ctx->chainLast.lane->everLaunched && hipEventQuery(ctx->chainLast.lane->done) == hipErrorNotReady)      // (`done`: its kernel, every kernel before it, its miss shading)
            chainPred = int(ctx->chainLast.idx);
        (void)hipGetLastError();      // hipErrorNotReady is not an error
    }
    uint32_t wavesPerSimd = ctx->opts.waves_per_simd ? ctx->opts.waves_per_simd : lane.forceWavesPerSimd;
    if (!wavesPerSimd) {
        wavesPerSimd = 6u;
        // (only for batches whose drain is a visible share of the launch: the device-resident path tracer's 16M-ray bounces run
        //  3.42 Grays/s end to end with full grids, 3.08 with halved ones)
        for (uint32_t i = 0; i < ctx->opts.lanes && count <= (2u << 20); ++i) {
            const Lane& other = ctx->lanes[i];
            if (&other != &lane && other.everLaunched.load(std::memory_order_relaxed) && hipEventQuery(other.done) == hipErrorNotReady) { wavesPerSimd = ctx->overlapWaves; break; }
        }
        (void)hipGetLastError();      // hipErrorNotReady is not an error
    }
    const uint32_t wavesPerBlock = uint32_t(v.block) / 64u;
    uint32_t blocksPerCU = (wavesPerSimd * 4u) / wavesPerBlock;
    if (blocksPerCU < 1u) blocksPerCU = 1u;
    if (blocksPerCU > kLdsPerCU / ldsBytes) blocksPerCU = kLdsPerCU / ldsBytes;
    uint32_t blocks = uint32_t(ctx->numCUs) * blocksPerCU;
    const uint32_t blocksNeeded = (count + uint32_t(v.block) - 1) / uint32_t(v.block);
    if (blocks > blocksNeeded) blocks = blocksNeeded;
    const uint32_t gridThreads = blocks * uint32_t(v.block);
    const uint32_t stackBound = v.wide ? scene->wideStack : scene->info.inner_height;
    const uint32_t spillLevels = (stackBound > uint32_t(v.stackLevels()) ? stackBound - uint32_t(v.stackLevels()) : 0u) * uint32_t(v.slots);
    if (int rc = ensureSpill(ctx, lane, uint32_t(ctx->numCUs) * 2048u, spillLevels)) return rc;

    TraverseArgs a;
    a.rays = static_cast<const float4*>(dRays);
    a.results = static_cast<float4*>(dResults);
    a.count = count;
    a.nodes = scene->nodes; a.pairs = scene->pairs; a.remap = scene->remap;
    a.cacheCount = scene->info.node_count < uint32_t(v.cacheNodes) ? scene->info.node_count : uint32_t(v.cacheNodes);
    a.nodeBytes = scene->info.node_count * 64u;
    a.nodesSoa = scene->nodesSoa; a.nodeCount = scene->info.node_count;
    if (v.wide && !(v.quant ? scene->nodesWideQ : scene->nodesWide)) return fail(RACC_HIP_ERR_INVALID, "a 4-wide kernel needs a scene uploaded through a context created with that kernel_variant (45-53) or wide_below");
    if (v.quant) { a.nodes = scene->nodesWideQ; a.nodeCount = scene->wideCount; a.nodeBytes = scene->wideCount * 64u; }
    else if (v.wide) { a.nodes = scene->nodesWide; a.nodeCount = scene->wideCount; a.nodeBytes = scene->wideCount * 128u; }
    if (v.id == kSoaVariant && !scene->nodesSoa) return fail(RACC_HIP_ERR_INVALID, "the SoA ablation variant needs a scene uploaded through a context created with that variant");
    a.pairBytes = scene->info.pair_count * 48u;
    a.env = env ? env->pixels : nullptr;
    a.envW = env ? env->width : 0; a.envH = env ? env->height : 0;
    a.cursor = lane.cursor;
    a.chain = nullptr; a.chainRing = nullptr; a.rearm = 1u; a.raysBypassL1 = 0u;
    if (chain) {
        a.cursor = ctx->chainCursors + size_t(chainIdx) * 16;
        a.rearm = 0u;
        a.raysBypassL1 = ctx->raysBypassL1 ? 1u : 0u;
        a.chain = ctx->chainDev + chainIdx; a.chainRing = ctx->chainDev;
    }
    a.spill = lane.spill;
    a.spillStride = gridThreads;
    a.chunk = optOr(ctx->opts.chunk, 64u * uint32_t(v.slots));
    if (a.chunk > 65536u) a.chunk = 65536u;      // grid waves x chunk (the statically assigned first chunks) must stay far below 2^32
    // tools/gpu_policy_sweep.py.  Round 2: 12-20 idle lanes beat 32 by 2 % (4M-ray launch) to 3 % (1M-ray launches back to back); 44: -15 %.
    // Round 3, after the refill lost its scratch round trips: binary kernel refill at 12 / leaf step at 10 waiting lanes
    // 0.2255-0.2261 ms back to back, 1.062-1.064 per 4M rays against 0.2309-0.2319 / 1.070-1.079 with 20 / 12; the wide kernels
    // stay at 20 / 6 (12 / 6: 0.2185 vs 0.2176; 20 / 10: 0.2153).
    a.refillMin = optOr(ctx->opts.refill_min, v.wide ? 20u : 12u);
    a.leafMin = optOr(ctx->opts.leaf_min, v.wide ? 6u : 10u);
    a.maxIters = ctx->maxIters;
    a.trips = ctx->devTrips;
    a.tailActive = ctx->opts.tail_active ? (ctx->opts.tail_active > 64u ? 0u : ctx->opts.tail_active) : 32u;   // >64 disables
    a.regroup = optOr(ctx->opts.regroup_period, 8u);
    a.thinReps = optOr(ctx->opts.thin_reps, 8u);
    a.innerReps = optOr(ctx->opts.inner_reps, v.wide ? 2u : 3u);
    a.coopNum = ctx->opts.coop_same_pct ? (ctx->opts.coop_same_pct > 100u ? 0u : ctx->opts.coop_same_pct) : 20u;   // > 100 disables the cooperative fetch
    a.coopDen = 100u;
    a.leafInCpp = ctx->opts.leaf_step == 2u ? 1u : 0u;
    a.noFusedStep = ctx->opts.leaf_step == 3u ? 1u : 0u;
    a.noDrainPrefetch = ctx->opts.drain_prefetch == 1u ? 0u : 1u;      // off by default: measured -3 % on a 64k-ray batch, +4..10 % on 256k-1M
    a.stats = reinterpret_cast<unsigned long long*>(lane.cursor + 8);
    // the lane's cursor / ticket / spill serve one launch at a time: a launch on another stream than the lane's previous
    // one first waits for that one (same stream: stream order already does it)
    if (lane.everLaunched && lane.lastStream != stream) HIP_TRY(hipStreamWaitEvent(stream, lane.done, 0), "hipStreamWaitEvent(lane)");
    const bool timed = ctx->opts.time_kernels != 0u && !lane.ring.empty();
    if (timed) HIP_TRY(hipEventRecord(lane.ring[2 * lane.ringHead], stream), "hipEventRecord");
    hipLaunchKernelGGL(chain ? v.kernelChained : v.kernel, dim3(blocks), dim3(v.block), 0, stream, a);
    HIP_TRY(hipGetLastError(), "launch traverseKernel");
    if (timed) {
        HIP_TRY(hipEventRecord(lane.ring[2 * lane.ringHead + 1], stream), "hipEventRecord");
        lane.ringHead = (lane.ringHead + 1u) % kTimeRing;
        if (lane.ringCount < kTimeRing) ++lane.ringCount;
    }
    if (chain) {
        // the descriptor, and the link behind the predecessor: from here on the waves of every earlier launch of the chain may take
        // this batch's rays, so whatever follows on this stream (the miss shading, the lane's `done`) must see those kernels ended
        hipLaunchKernelGGL(chainPublishKernel, dim3(1), dim3(1), 0, ctx->chainStream, ctx->chainDev, chainIdx, a.rays, a.results, a.cursor, count,
                           blocks * uint32_t(v.block / 64) * a.chunk, chainPred);
        HIP_TRY(hipGetLastError(), "launch chainPublishKernel");
        // Every kernel issued before this one may have worked on this batch.  Those on this stream have ended when this one starts;
        // of every other lane the latest one is waited for (the earlier ones on its stream ended before it): two waits with three
        // lanes, and no chain of events from launch to launch that would serialise the ends of a long sequence.
        if (timed) lane.chainKernelEndEv = lane.ring[2 * ((lane.ringHead + kTimeRing - 1u) % kTimeRing) + 1];      // the timing pair's end event serves
        else {
            if (!lane.chainKernelEnd) HIP_TRY(hipEventCreateWithFlags(&lane.chainKernelEnd, hipEventDisableTiming), "hipEventCreate");
            HIP_TRY(hipEventRecord(lane.chainKernelEnd, stream), "hipEventRecord(chain kernel)");
            lane.chainKernelEndEv = lane.chainKernelEnd;
        }
        lane.chainKernelValid = true;
        if (chainPred >= 0)
            for (uint32_t i = 0; i < ctx->opts.lanes; ++i) {
                Lane& other = ctx->lanes[i];
                if (&other != &lane && other.chainKernelValid) HIP_TRY(hipStreamWaitEvent(stream, other.chainKernelEndEv, 0), "hipStreamWaitEvent(chain)");
            }
        ctx->chainLast.scene = scene; ctx->chainLast.env = env; ctx->chainLast.kernel = reinterpret_cast<const void*>(v.kernelChained);
        ctx->chainLast.idx = chainIdx; ctx->chainLast.lane = &lane; ctx->chainLast.valid = true;
        ++ctx->chainHead;
    }
    lane.pendingEnv = v.deferEnv && env != nullptr;
    lane.info.grid_blocks = blocks;
    lane.info.block_threads = uint32_t(v.block);
    lane.info.lds_bytes_per_block = ldsBytes;
    lane.info.waves_per_simd = blocksPerCU * wavesPerBlock / 4u;
    return RACC_HIP_OK;
}

// Marks the end of a lane's launch on `stream` (after its last kernel): the next launch of this lane on another stream waits here.
int markLaneDone(Lane& lane, hipStream_t stream) {
    HIP_TRY(hipEventRecord(lane.done, stream), "hipEventRecord(lane done)");
    lane.lastStream = stream;
    lane.everLaunched = true;
    return RACC_HIP_OK;
}

int launchEnvShadeOnly(racc_hip_ctx* ctx, Lane& lane, hipStream_t stream, const racc_hip_env* env, void* dResults, uint32_t count);

int launchEnvShade(racc_hip_ctx* ctx, Lane& lane, hipStream_t stream, const racc_hip_env* env, void* dResults, uint32_t count) {
    if (int rc = launchEnvShadeOnly(ctx, lane, stream, env, dResults, count)) return rc;
    return count ? markLaneDone(lane, stream) : RACC_HIP_OK;
}

int launchEnvShadeOnly(racc_hip_ctx* ctx, Lane& lane, hipStream_t stream, const racc_hip_env* env, void* dResults, uint32_t count) {
    if (!lane.pendingEnv || !env || !count) return RACC_HIP_OK;
    lane.pendingEnv = false;
    uint32_t blocks = (count + 255u) / 256u;
    const uint32_t cap = uint32_t(ctx->numCUs) * 8u;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(envShadeKernel, dim3(blocks), dim3(256), 0, stream, static_cast<float4*>(dResults), count, env->pixels, env->width, env->height);
    HIP_TRY(hipGetLastError(), "launch envShadeKernel");
    return RACC_HIP_OK;
}

hipError_t initLane(Lane& l, bool timeKernels) {
    hipError_t e = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&l.cursor), 256);
    if (e == hipSuccess) e = hipMemset(l.cursor, 0, 256);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&l.done, hipEventDisableTiming);
    if (e == hipSuccess && timeKernels) {
        l.ring.resize(2 * kTimeRing, nullptr);
        for (hipEvent_t& ev : l.ring) if (e == hipSuccess) e = hipEventCreate(&ev);
    }
    return e;
}

void freeLane(Lane& l) {
    for (Lane*& h : l.helper) if (h) { freeLane(*h); delete h; h = nullptr; }
    if (l.stream) hipStreamSynchronize(l.stream);
    for (hipEvent_t ev : l.events) hipEventDestroy(ev);
    for (hipEvent_t ev : l.pipeEvents) hipEventDestroy(ev);
    for (hipEvent_t ev : l.ring) if (ev) hipEventDestroy(ev);
    if (l.done) hipEventDestroy(l.done);
    if (l.chainKernelEnd) hipEventDestroy(l.chainKernelEnd);
    if (l.copyIn) hipStreamDestroy(l.copyIn);
    if (l.copyOut) hipStreamDestroy(l.copyOut);
    if (l.cursor) hipFree(l.cursor);
    if (l.spill) hipFree(l.spill);
    if (l.dRays) hipFree(l.dRays);
    if (l.dResults) hipFree(l.dResults);
    if (l.stream) hipStreamDestroy(l.stream);
}

int checkLane(racc_hip_ctx* ctx, uint32_t lane) {
    if (!ctx) return fail(RACC_HIP_ERR_INVALID, "ctx is NULL");
    if (lane >= ctx->opts.lanes) return fail(RACC_HIP_ERR_INVALID, "lane out of range");
    return RACC_HIP_OK;
}

int ensureStaging(Lane& lane, uint32_t count) {
    if (lane.capacity >= count) return RACC_HIP_OK;
    if (lane.dRays) { HIP_TRY(hipFree(lane.dRays), "hipFree(staging rays)"); lane.dRays = nullptr; }
    if (lane.dResults) { HIP_TRY(hipFree(lane.dResults), "hipFree(staging results)"); lane.dResults = nullptr; }
    lane.capacity = 0;
    uint64_t cap64 = 32768;
    while (cap64 < count) cap64 <<= 1;
    const uint32_t cap = cap64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : uint32_t(cap64);
    HIP_TRY(hipMalloc(&lane.dRays, size_t(cap) * 32), "hipMalloc(staging rays)");
    HIP_TRY(hipMalloc(&lane.dResults, size_t(cap) * 16), "hipMalloc(staging results)");
    lane.capacity = cap;
    return RACC_HIP_OK;
}

// After a synchronisation: did a wave of any launch since the last check give up at the iteration limit?  (Only a scene
// blob that passed validation and still does not terminate, or an absurd RACC_MAX_ITERS, can do that; its results are
// incomplete and the caller must hear about it.)
int checkWatchdog(racc_hip_ctx* ctx) {
    const uint32_t t = *static_cast<volatile uint32_t*>(ctx->hostTrips);
    uint32_t seen = ctx->seenTrips.load();
    while (seen != t) {
        if (ctx->seenTrips.compare_exchange_weak(seen, t))
            return fail(RACC_HIP_ERR_DEVICE, "traversal watchdog: a wave exceeded the iteration limit, results of the launch are incomplete (corrupt scene blob?)");
    }
    return RACC_HIP_OK;
}

}  // namespace

extern "C" {

#ifdef RACC_EXPERIMENTAL
const char* racc_hip_version(void) { return "racc-hip 0.2 (gfx950, experimental kernels included)"; }
#else
const char* racc_hip_version(void) { return "racc-hip 0.2 (gfx950)"; }
#endif

int racc_hip_lane_count(const racc_hip_ctx* ctx, uint32_t* lanes, uint32_t* auto_lanes) {
    if (!ctx) return fail(RACC_HIP_ERR_INVALID, "ctx is NULL");
    if (lanes) *lanes = ctx->opts.lanes;
    if (auto_lanes) *auto_lanes = ctx->autoLanes;
    return RACC_HIP_OK;
}

int racc_hip_variant_available(uint32_t kernel_variant) {
    if (kernel_variant == 0u) return 1;
    return variantById(kernel_variant) ? 1 : 0;
}

int racc_hip_device_count(int* count) {
    if (!count) return fail(RACC_HIP_ERR_INVALID, "count is NULL");
    *count = 0;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(RACC_HIP_ERR_NO_DEVICE, "hipGetDeviceCount", e);
    *count = n;
    return RACC_HIP_OK;
}

int racc_hip_create(int device, const racc_hip_options* opts, racc_hip_ctx** out) {
    if (!out) return fail(RACC_HIP_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(RACC_HIP_ERR_NO_DEVICE, "no HIP device (there is no CPU fallback)", e);
    if (device < 0 || device >= n) return fail(RACC_HIP_ERR_INVALID, "device ordinal out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device), "hipGetDeviceProperties");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_msg, sizeof(g_msg), "device %d is %s; this engine is built for gfx950 only", device, prop.gcnArchName);
        return fail(RACC_HIP_ERR_NO_DEVICE, g_msg);
    }
    HIP_TRY(hipSetDevice(device), "hipSetDevice");
    racc_hip_ctx* ctx = new (std::nothrow) racc_hip_ctx();
    if (!ctx) return fail(RACC_HIP_ERR_NOMEM, "out of host memory");
    ctx->device = device;
    ctx->numCUs = prop.multiProcessorCount;
    if (opts) {
        const size_t n_copy = opts->struct_size && opts->struct_size < sizeof(racc_hip_options) ? opts->struct_size : sizeof(racc_hip_options);
        memcpy(&ctx->opts, opts, n_copy);
    }
    ctx->opts.struct_size = sizeof(racc_hip_options);
    if (ctx->opts.kernel_variant && !variantById(ctx->opts.kernel_variant)) {
        delete ctx;
        return fail(RACC_HIP_ERR_INVALID, "kernel_variant is not in this build (experimental kernels: make EXPERIMENTAL=1)");
    }
    // How many launches to keep in flight.  HIP streams share hardware queues (GPU_MAX_HW_QUEUES, 4 unless the environment says
    // otherwise), and kernels of two streams on one hardware queue do not overlap.  Measured on 1M-ray diffuse batches back to
    // back (tools/gpu_overlap.py), ms per batch: 4 queues: 3 lanes x 2 waves per SIMD 0.260, 4 x 2 0.322, 6 x 1 0.349;
    // 8 queues: 3 x 2 0.260, 4 x 2 0.253, 6 x 1 0.245-0.251, 7 x 1 0.247, 8 x 1 0.32; 16 queues: 6 x 1 0.248, 8 x 1 0.254.
    // So: six thin launches when the runtime was given >= 8 hardware queues, three otherwise.
    {
        const char* q = std::getenv("GPU_MAX_HW_QUEUES");
        const int hwQueues = q ? std::atoi(q) : 4;
        ctx->hwQueues = hwQueues;
        const bool defaultLanes = ctx->opts.lanes == 0;
        ctx->chainEnabled = ctx->opts.chain_launches != 2u;
        if (const char* c = std::getenv("RACC_CHAIN")) ctx->chainEnabled = std::atoi(c) != 0;
        // chained launches (launchTraverse) need no thin grids: three lanes are enough to keep successors queued (measured, 1M-ray
        // batches: 20 in a row 0.285 ms each with 3 lanes, 0.292 with 2, 0.32 with 4, 0.42 with 6; 200 in a row 0.242 / 0.245 / - / 0.273)
        ctx->autoLanes = ctx->chainEnabled ? 3u : (hwQueues >= 8 ? 6u : 3u);
        if (const char* r = std::getenv("RACC_AUTO_LANES")) if (std::atoi(r) > 0) ctx->autoLanes = uint32_t(std::atoi(r));
        if (defaultLanes && ctx->autoLanes > 4u) ctx->opts.lanes = ctx->autoLanes;
        if (!ctx->opts.lanes) ctx->opts.lanes = 4;                       // RayAccelerator.cpp:436
        if (ctx->opts.lanes > RACC_HIP_MAX_LANES) ctx->opts.lanes = RACC_HIP_MAX_LANES;
        if (ctx->autoLanes > ctx->opts.lanes) ctx->autoLanes = ctx->opts.lanes;
        ctx->overlapWaves = ctx->autoLanes >= 5u ? 1u : 2u;
    }
    if (ctx->opts.waves_per_simd > 8) ctx->opts.waves_per_simd = 8;
    if (ctx->opts.refill_min > 64) ctx->opts.refill_min = 64;
    if (ctx->opts.leaf_min > 64) ctx->opts.leaf_min = 64;
    for (uint32_t i = 0; i < ctx->opts.lanes; ++i) {
        const hipError_t e3 = initLane(ctx->lanes[i], ctx->opts.time_kernels != 0u);
        if (e3 != hipSuccess) { racc_hip_destroy(ctx); return fail(RACC_HIP_ERR_DEVICE, "lane setup", e3); }
    }
    {
        hipError_t e1 = hipHostMalloc(reinterpret_cast<void**>(&ctx->hostTrips), 64, hipHostMallocMapped);
        if (e1 == hipSuccess) { *ctx->hostTrips = 0; e1 = hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->devTrips), ctx->hostTrips, 0); }
        if (e1 != hipSuccess) { racc_hip_destroy(ctx); return fail(RACC_HIP_ERR_DEVICE, "watchdog word", e1); }
        if (const char* m = std::getenv("RACC_MAX_ITERS")) { const long long v = std::atoll(m); if (v > 0 && v < (1ll << 31)) ctx->maxIters = uint32_t(v); }
    }
    {
        ctx->chainEnabled = ctx->opts.chain_launches != 2u;
        if (const char* c = std::getenv("RACC_CHAIN")) ctx->chainEnabled = std::atoi(c) != 0;
        if (const char* c = std::getenv("RACC_RAY_SCOPE")) ctx->raysBypassL1 = std::atoi(c) != 0;
        hipError_t e1 = hipMalloc(reinterpret_cast<void**>(&ctx->chainDev), sizeof(ChainDesc) * racc_hip_ctx::kChainRing);
        if (e1 == hipSuccess) e1 = hipMemset(ctx->chainDev, 0, sizeof(ChainDesc) * racc_hip_ctx::kChainRing);
        if (e1 == hipSuccess) {      // highest priority: a publish kernel must not wait behind the persistent waves it is meant to feed
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            e1 = hipStreamCreateWithPriority(&ctx->chainStream, hipStreamNonBlocking, hi);
        }
        if (e1 == hipSuccess) e1 = hipMalloc(reinterpret_cast<void**>(&ctx->chainCursors), size_t(racc_hip_ctx::kChainRing) * 64);
        if (e1 == hipSuccess) e1 = hipMemset(ctx->chainCursors, 0, size_t(racc_hip_ctx::kChainRing) * 64);
        if (e1 != hipSuccess) { racc_hip_destroy(ctx); return fail(RACC_HIP_ERR_DEVICE, "chain ring", e1); }
    }
    *out = ctx;
    return RACC_HIP_OK;
}

int racc_hip_destroy(racc_hip_ctx* ctx) {
    if (!ctx) return RACC_HIP_OK;
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();              // launches given a caller's own stream (racc_hip_intersect_device) included
    if (ctx->hostTrips) hipHostFree(ctx->hostTrips);
    if (ctx->chainDev) hipFree(ctx->chainDev);
    if (ctx->chainStream) hipStreamDestroy(ctx->chainStream);
    if (ctx->chainCursors) hipFree(ctx->chainCursors);
    for (Lane& l : ctx->lanes) freeLane(l);
    delete ctx;
    return RACC_HIP_OK;
}

int racc_hip_scene_upload(racc_hip_ctx* ctx, const void* nodes64, uint32_t node_count,
                          const void* pairs48, uint32_t pair_count,
                          const uint32_t* remap, uint32_t remap_count, racc_hip_scene** out) {
    if (!ctx || !out) return fail(RACC_HIP_ERR_INVALID, "ctx/out is NULL");
    *out = nullptr;
    if (!nodes64 || !pairs48 || !remap) return fail(RACC_HIP_ERR_INVALID, "scene blob pointer is NULL");
    racc_hip_scene_info info{};
    if (int rc = validateScene(static_cast<const GpuNodeHost*>(nodes64), node_count, pair_count, remap_count, info)) return rc;
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    racc_hip_scene* s = new (std::nothrow) racc_hip_scene();
    if (!s) return fail(RACC_HIP_ERR_NOMEM, "out of host memory");
    const size_t nb = size_t(node_count) * 64, pb = size_t(pair_count) * 48, rb = size_t(remap_count) * 4;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&s->nodes), nb);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->pairs), pb);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&s->remap), rb ? rb : 4);
    std::vector<GpuNodeHost> ordered;
    reorderNodes(static_cast<const GpuNodeHost*>(nodes64), node_count, ordered);
    if (e == hipSuccess) e = hipMemcpy(s->nodes, ordered.data(), nb, hipMemcpyHostToDevice);
    if (e == hipSuccess && ctx->opts.kernel_variant == uint32_t(kSoaVariant)) {
        std::vector<float> planes(size_t(node_count) * 16);
        const float* rec = reinterpret_cast<const float*>(ordered.data());
        for (uint32_t i = 0; i < node_count; ++i)
            for (uint32_t p = 0; p < 4; ++p)
                std::memcpy(&planes[(size_t(p) * node_count + i) * 4], rec + size_t(i) * 16 + p * 4, 16);
        e = hipMalloc(reinterpret_cast<void**>(&s->nodesSoa), nb);
        if (e == hipSuccess) e = hipMemcpy(s->nodesSoa, planes.data(), nb, hipMemcpyHostToDevice);
    }
    size_t wb = 0;
    // the 4-wide copies of the tree only for contexts that can select a wide kernel (each costs device memory like the nodes)
    const Variant& own = pickVariant(ctx, info.inner_height);
    if (e == hipSuccess && (own.wide || ctx->opts.wide_below != 0u)) {
        std::vector<WideNode> wide;
        collapseWide(static_cast<const GpuNodeHost*>(nodes64), node_count, wide, s->wideStack);
        s->wideCount = uint32_t(wide.size());
        if (!own.quant || ctx->opts.wide_below != 0u) {
            wb = wide.size() * sizeof(WideNode);
            e = hipMalloc(reinterpret_cast<void**>(&s->nodesWide), wb);
            if (e == hipSuccess) e = hipMemcpy(s->nodesWide, wide.data(), wb, hipMemcpyHostToDevice);
        }
        if (e == hipSuccess && own.quant) {
            std::vector<WideNodeQ> packed;
            if (int rc = quantiseWide(wide, packed)) { racc_hip_scene_free(ctx, s); return rc; }
            const size_t qb = packed.size() * sizeof(WideNodeQ);
            wb += qb;
            e = hipMalloc(reinterpret_cast<void**>(&s->nodesWideQ), qb);
            if (e == hipSuccess) e = hipMemcpy(s->nodesWideQ, packed.data(), qb, hipMemcpyHostToDevice);
        }
    }
    if (e == hipSuccess) e = hipMemcpy(s->pairs, pairs48, pb, hipMemcpyHostToDevice);
    if (e == hipSuccess && rb) e = hipMemcpy(s->remap, remap, rb, hipMemcpyHostToDevice);
    if (e != hipSuccess) { racc_hip_scene_free(ctx, s); return fail(RACC_HIP_ERR_DEVICE, "scene upload", e); }
    info.node_count = node_count; info.pair_count = pair_count; info.remap_count = remap_count;
    info.device_bytes = nb + wb + pb + rb;
    {
        const Variant& v = pickVariant(ctx, info.inner_height);
        const uint32_t bound = v.wide ? s->wideStack : info.inner_height;
        info.spill_levels = bound > uint32_t(v.stackLevels()) ? bound - uint32_t(v.stackLevels()) : 0u;
    }
    s->info = info;
    *out = s;
    return RACC_HIP_OK;
}

int racc_hip_scene_free(racc_hip_ctx* ctx, racc_hip_scene* s) {
    if (!s) return RACC_HIP_OK;
    if (ctx) { std::lock_guard<std::mutex> g(ctx->chainMutex); if (ctx->chainLast.scene == s) ctx->chainLast.valid = false; }
    if (ctx) hipSetDevice(ctx->device);
    if (s->nodes) hipFree(s->nodes);
    if (s->nodesWide) hipFree(s->nodesWide);
    if (s->nodesWideQ) hipFree(s->nodesWideQ);
    if (s->nodesSoa) hipFree(s->nodesSoa);
    if (s->pairs) hipFree(s->pairs);
    if (s->remap) hipFree(s->remap);
    delete s;
    return RACC_HIP_OK;
}

int racc_hip_scene_get_info(const racc_hip_scene* scene, racc_hip_scene_info* info) {
    if (!scene || !info) return fail(RACC_HIP_ERR_INVALID, "scene/info is NULL");
    *info = scene->info;
    return RACC_HIP_OK;
}

int racc_hip_env_upload(racc_hip_ctx* ctx, const float* rgba, uint32_t width, uint32_t height, racc_hip_env** out) {
    if (!ctx || !out) return fail(RACC_HIP_ERR_INVALID, "ctx/out is NULL");
    *out = nullptr;
    if (!rgba || !width || !height) return fail(RACC_HIP_ERR_INVALID, "environment image is empty");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    racc_hip_env* env = new (std::nothrow) racc_hip_env();
    if (!env) return fail(RACC_HIP_ERR_NOMEM, "out of host memory");
    const size_t bytes = size_t(width) * height * 16;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&env->pixels), bytes);
    if (e == hipSuccess) e = hipMemcpy(env->pixels, rgba, bytes, hipMemcpyHostToDevice);
    if (e != hipSuccess) { racc_hip_env_free(ctx, env); return fail(RACC_HIP_ERR_DEVICE, "environment upload", e); }
    env->width = width; env->height = height;
    *out = env;
    return RACC_HIP_OK;
}

int racc_hip_env_free(racc_hip_ctx* ctx, racc_hip_env* env) {
    if (!env) return RACC_HIP_OK;
    if (ctx) { std::lock_guard<std::mutex> g(ctx->chainMutex); if (ctx->chainLast.env == env) ctx->chainLast.valid = false; }
    if (ctx) hipSetDevice(ctx->device);
    if (env->pixels) hipFree(env->pixels);
    delete env;
    return RACC_HIP_OK;
}

int racc_hip_register_host(racc_hip_ctx* ctx, void* ptr, uint64_t bytes) {
    if (!ctx || !ptr || !bytes) return fail(RACC_HIP_ERR_INVALID, "register_host: bad argument");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterPortable | hipHostRegisterMapped), "hipHostRegister");
    return RACC_HIP_OK;
}

int racc_hip_unregister_host(racc_hip_ctx* ctx, void* ptr) {
    if (!ctx || !ptr) return fail(RACC_HIP_ERR_INVALID, "unregister_host: bad argument");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    HIP_TRY(hipHostUnregister(ptr), "hipHostUnregister");
    return RACC_HIP_OK;
}

int racc_hip_register_stream(racc_hip_ctx* ctx, void* rays, void* results, uint32_t capacity) {
    if (!ctx || !rays || !results || !capacity) return fail(RACC_HIP_ERR_INVALID, "register_stream: bad argument");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    HIP_TRY(hipHostRegister(rays, size_t(capacity) * 32, hipHostRegisterPortable | hipHostRegisterMapped), "hipHostRegister(rays)");
    hipError_t e = hipHostRegister(results, size_t(capacity) * 16, hipHostRegisterPortable);
    if (e != hipSuccess) { hipHostUnregister(rays); return fail(RACC_HIP_ERR_DEVICE, "hipHostRegister(results)", e); }
    return RACC_HIP_OK;
}

int racc_hip_unregister_stream(racc_hip_ctx* ctx, void* rays, void* results) {
    if (!ctx) return fail(RACC_HIP_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    if (rays) HIP_TRY(hipHostUnregister(rays), "hipHostUnregister(rays)");
    if (results) HIP_TRY(hipHostUnregister(results), "hipHostUnregister(results)");
    return RACC_HIP_OK;
}

int racc_hip_intersect_async(racc_hip_ctx* ctx, const racc_hip_scene* scene, const racc_hip_env* env,
                             const void* rays, void* results, uint32_t count, uint32_t lane) {
    if (int rc = checkLane(ctx, lane)) return rc;
    if (!scene) return fail(RACC_HIP_ERR_INVALID, "scene is NULL");
    if (!count) return RACC_HIP_OK;
    if (!rays || !results) return fail(RACC_HIP_ERR_INVALID, "rays/results is NULL");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    Lane& l = ctx->lanes[lane];
    std::lock_guard<std::mutex> guard(l.mutex);
    HIP_TRY(hipStreamSynchronize(l.stream), "hipStreamSynchronize");     // staging buffers are reused per lane
    if (int rc = ensureStaging(l, count)) return rc;
    HIP_TRY(hipMemcpyAsync(l.dRays, rays, size_t(count) * 32, hipMemcpyHostToDevice, l.stream), "H2D rays");
    if (int rc = launchTraverse(ctx, l, l.stream, scene, env, l.dRays, l.dResults, count)) return rc;
    if (int rc = launchEnvShade(ctx, l, l.stream, env, l.dResults, count)) return rc;
    HIP_TRY(hipMemcpyAsync(results, l.dResults, size_t(count) * 16, hipMemcpyDeviceToHost, l.stream), "D2H results");
    return RACC_HIP_OK;
}

int racc_hip_wait(racc_hip_ctx* ctx, uint32_t lane) {
    if (!ctx) return fail(RACC_HIP_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    if (lane == RACC_HIP_LANE_AUTO) {         // every lane: its own stream and, through `done`, whatever stream its last launch went to
        for (uint32_t i = 0; i < ctx->opts.lanes; ++i) {
            Lane& l = ctx->lanes[i];
            std::lock_guard<std::mutex> guard(l.mutex);
            HIP_TRY(hipStreamSynchronize(l.stream), "hipStreamSynchronize");
            if (l.everLaunched) HIP_TRY(hipEventSynchronize(l.done), "hipEventSynchronize");
        }
        return checkWatchdog(ctx);
    }
    if (int rc = checkLane(ctx, lane)) return rc;
    {
        Lane& l = ctx->lanes[lane];
        std::lock_guard<std::mutex> guard(l.mutex);
        HIP_TRY(hipStreamSynchronize(l.stream), "hipStreamSynchronize");
        if (l.everLaunched) HIP_TRY(hipEventSynchronize(l.done), "hipEventSynchronize");
    }
    return checkWatchdog(ctx);
}

int racc_hip_intersect(racc_hip_ctx* ctx, const racc_hip_scene* scene, const racc_hip_env* env,
                       const void* rays, void* results, uint32_t count, uint32_t lane) {
    if (count >= 262144u) return racc_hip_intersect_streams(ctx, scene, env, 1, &rays, &results, &count, lane);   // sliced: copies beside kernels
    if (int rc = racc_hip_intersect_async(ctx, scene, env, rays, results, count, lane)) return rc;
    return racc_hip_wait(ctx, lane);
}

int racc_hip_intersect_streams(racc_hip_ctx* ctx, const racc_hip_scene* scene, const racc_hip_env* env,
                               uint32_t n_streams, const void* const* rays, void* const* results,
                               const uint32_t* counts, uint32_t lane) {
    if (int rc = checkLane(ctx, lane)) return rc;
    if (!scene) return fail(RACC_HIP_ERR_INVALID, "scene is NULL");
    if (!n_streams) return RACC_HIP_OK;
    if (!rays || !results || !counts) return fail(RACC_HIP_ERR_INVALID, "streams: NULL array");
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_streams; ++i) {
        if (counts[i] && (!rays[i] || !results[i])) return fail(RACC_HIP_ERR_INVALID, "streams: NULL rays/results");
        total += counts[i];
    }
    if (total > 0xFFFFFFFFull) return fail(RACC_HIP_ERR_LIMIT, "streams: more than 2^32-1 rays in one launch");
    if (!total) return RACC_HIP_OK;
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    Lane& l = ctx->lanes[lane];
    std::lock_guard<std::mutex> guard(l.mutex);
    HIP_TRY(hipStreamSynchronize(l.stream), "hipStreamSynchronize");
    if (int rc = ensureStaging(l, uint32_t(total))) return rc;
    // Copies `what` (0 = rays H2D, 1 = results D2H) of the global ray range [g0, g1) on stream st, stream by stream.
    auto copyRange = [&](int what, uint64_t g0, uint64_t g1, hipStream_t st) -> hipError_t {
        uint64_t off = 0;
        for (uint32_t i = 0; i < n_streams; ++i) {
            const uint64_t s0 = off, s1 = off + counts[i];
            off = s1;
            const uint64_t a0 = s0 > g0 ? s0 : g0, a1 = s1 < g1 ? s1 : g1;
            if (a0 >= a1) continue;
            hipError_t e;
            if (what == 0) e = hipMemcpyAsync(static_cast<char*>(l.dRays) + a0 * 32, static_cast<const char*>(rays[i]) + (a0 - s0) * 32, size_t(a1 - a0) * 32, hipMemcpyHostToDevice, st);
            else e = hipMemcpyAsync(static_cast<char*>(results[i]) + (a0 - s0) * 16, static_cast<char*>(l.dResults) + a0 * 16, size_t(a1 - a0) * 16, hipMemcpyDeviceToHost, st);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    };
    static const uint64_t kSlice = [] { const char* e = std::getenv("RACC_SLICE"); const long long v = e ? std::atoll(e) : 0; return v > 0 ? uint64_t(v) : uint64_t(262144); }();
    uint32_t slices = total >= 2 * kSlice ? uint32_t((total + kSlice - 1) / kSlice > 16 ? 16 : (total + kSlice - 1) / kSlice) : 1u;
    if (slices > 1) {   // only page-locked host memory copies asynchronously; pageable buffers would just pay for the extra launches
        hipPointerAttribute_t at{};
        uint32_t first = 0;
        while (first < n_streams && !counts[first]) ++first;
        if (hipPointerGetAttributes(&at, rays[first]) != hipSuccess || at.type != hipMemoryTypeHost) { (void)hipGetLastError(); slices = 1; }
        else if (hipPointerGetAttributes(&at, results[first]) != hipSuccess || at.type != hipMemoryTypeHost) { (void)hipGetLastError(); slices = 1; }
    }
    if (slices == 1) {
        HIP_TRY(copyRange(0, 0, total, l.stream), "H2D rays");
        if (int rc = launchTraverse(ctx, l, l.stream, scene, env, l.dRays, l.dResults, uint32_t(total))) return rc;
        if (int rc = launchEnvShade(ctx, l, l.stream, env, l.dResults, uint32_t(total))) return rc;
        HIP_TRY(copyRange(1, 0, total, l.stream), "D2H results");
        HIP_TRY(hipStreamSynchronize(l.stream), "hipStreamSynchronize");
        return checkWatchdog(ctx);
    }
    // The five streams of the pipeline — three for the slices' kernels, copy-in, copy-out — should sit on five hardware
    // queues.  HIP hands queues out in creation order: with >= 8 of them the kernels go to three helper lanes created here in
    // a row with the copy streams (the lane's own stream, created with the context, may share a queue with any of them: with
    // six lanes and 8 queues it did, 0.87 instead of 1.03 Grays/s); with the default 4 the lane's own stream and two helpers.
    const uint32_t nHelpers = ctx->hwQueues >= 8 ? 3u : 2u;
    if (nHelpers == 2u) {      // 4 queues: copy streams first (measured: the other order costs 5 %)
        if (!l.copyIn) HIP_TRY(hipStreamCreateWithFlags(&l.copyIn, hipStreamNonBlocking), "hipStreamCreate");
        if (!l.copyOut) HIP_TRY(hipStreamCreateWithFlags(&l.copyOut, hipStreamNonBlocking), "hipStreamCreate");
    }
    for (uint32_t i = 0; i < nHelpers; ++i)
        if (!l.helper[i]) {
            Lane*& h = l.helper[i];
            h = new (std::nothrow) Lane();
            if (!h) return fail(RACC_HIP_ERR_NOMEM, "out of host memory");
            const hipError_t e = initLane(*h, false);
            if (e != hipSuccess) { freeLane(*h); delete h; h = nullptr; return fail(RACC_HIP_ERR_DEVICE, "helper lane setup", e); }
        }
    Lane* const run[3] = {nHelpers == 3u ? l.helper[2] : &l, l.helper[0], l.helper[1]};
    if (!l.copyIn) HIP_TRY(hipStreamCreateWithFlags(&l.copyIn, hipStreamNonBlocking), "hipStreamCreate");
    if (!l.copyOut) HIP_TRY(hipStreamCreateWithFlags(&l.copyOut, hipStreamNonBlocking), "hipStreamCreate");
    // Cut into slices so that the PCIe copy of slice k+1 (in) and of slice k-1 (out) run beside the kernel of slice k (PCIe is
    // full duplex; a 1M-ray batch is 32 MiB in, 16 MiB out), the kernels on the lane and its two helpers in turn so that one
    // slice's drain runs beside the next one's bulk.  Measured on page-locked arrays: 1.04 Grays/s at 1M rays, 1.37 at 4M
    // (= 66 GB/s over the link, both directions together; the DMA engines deliver ~63).
    // (Measured and rejected: letting the kernel read page-locked rays straight from host memory — no copy-in stage at all —
    //  reaches 0.87 Grays/s at 1M and 4M rays: PCIe reads issued by the waves' refills run at ~28 GB/s, DMA copies at ~52.)
    while (l.pipeEvents.size() < size_t(slices) * 2) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
        l.pipeEvents.push_back(ev);
    }
    const uint64_t per = ((total + slices - 1) / slices + 63) / 64 * 64;
    std::vector<uint64_t> cut;          // slice boundaries (equal slices: a short first and/or last slice measured no gain)
    for (uint64_t g = 0; g < total; g += per) cut.push_back(g);
    cut.push_back(total);
    slices = uint32_t(cut.size() - 1);
    while (l.pipeEvents.size() < size_t(slices) * 2) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
        l.pipeEvents.push_back(ev);
    }
    for (uint32_t k = 0; k < slices; ++k) {
        const uint64_t g0 = cut[k], g1 = cut[k + 1];
        Lane& r = *run[k % 3u];           // the three take the slices' kernels in turn: a slice's drain overlaps the next one's bulk
        struct GridGuard { Lane& l; ~GridGuard() { l.forceWavesPerSimd = 0u; } } gridGuard{r};      // (also on the error returns below)
        r.forceWavesPerSimd = 3u;
        HIP_TRY(copyRange(0, g0, g1, l.copyIn), "H2D rays");
        HIP_TRY(hipEventRecord(l.pipeEvents[2 * k], l.copyIn), "hipEventRecord");
        HIP_TRY(hipStreamWaitEvent(r.stream, l.pipeEvents[2 * k], 0), "hipStreamWaitEvent");
        if (int rc = launchTraverse(ctx, r, r.stream, scene, env, static_cast<char*>(l.dRays) + g0 * 32, static_cast<char*>(l.dResults) + g0 * 16, uint32_t(g1 - g0))) return rc;
        if (int rc = launchEnvShade(ctx, r, r.stream, env, static_cast<char*>(l.dResults) + g0 * 16, uint32_t(g1 - g0))) return rc;
        HIP_TRY(hipEventRecord(l.pipeEvents[2 * k + 1], r.stream), "hipEventRecord");
        HIP_TRY(hipStreamWaitEvent(l.copyOut, l.pipeEvents[2 * k + 1], 0), "hipStreamWaitEvent");
        HIP_TRY(copyRange(1, g0, g1, l.copyOut), "D2H results");
    }
    for (Lane* h : l.helper) if (h) HIP_TRY(hipStreamSynchronize(h->stream), "hipStreamSynchronize");
    HIP_TRY(hipStreamSynchronize(l.copyOut), "hipStreamSynchronize");
    HIP_TRY(hipStreamSynchronize(l.stream), "hipStreamSynchronize");
    return checkWatchdog(ctx);
}

int racc_hip_intersect_device(racc_hip_ctx* ctx, const racc_hip_scene* scene, const racc_hip_env* env,
                              const void* d_rays, void* d_results, uint32_t count, uint32_t lane, void* stream) {
    if (!ctx) return fail(RACC_HIP_ERR_INVALID, "ctx is NULL");
    // round robin over the lanes in rotation (racc_hip_create): consecutive launches overlap
    if (lane == RACC_HIP_LANE_AUTO) lane = ctx->nextLane.fetch_add(1u) % ctx->autoLanes;
    if (int rc = checkLane(ctx, lane)) return rc;
    if (!scene) return fail(RACC_HIP_ERR_INVALID, "scene is NULL");
    if (!count) return RACC_HIP_OK;
    if (!d_rays || !d_results) return fail(RACC_HIP_ERR_INVALID, "d_rays/d_results is NULL");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    Lane& l = ctx->lanes[lane];
    std::lock_guard<std::mutex> guard(l.mutex);
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : l.stream;
    if (int rc = launchTraverse(ctx, l, st, scene, env, d_rays, d_results, count, /*mayChain=*/stream == nullptr)) return rc;
    return launchEnvShade(ctx, l, st, env, d_results, count);
}

int racc_hip_intersect_device_timed(racc_hip_ctx* ctx, const racc_hip_scene* scene, const racc_hip_env* env,
                                    const void* d_rays, void* d_results, uint32_t count,
                                    uint32_t lane, uint32_t iters, float* ms) {
    if (int rc = checkLane(ctx, lane)) return rc;
    if (!scene || !ms || !iters) return fail(RACC_HIP_ERR_INVALID, "timed: bad argument");
    if (!d_rays || !d_results || !count) return fail(RACC_HIP_ERR_INVALID, "timed: empty batch");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    Lane& l = ctx->lanes[lane];
    std::lock_guard<std::mutex> guard(l.mutex);
    while (l.events.size() < size_t(iters) * 2) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev), "hipEventCreate");
        l.events.push_back(ev);
    }
    for (uint32_t i = 0; i < iters; ++i) {
        HIP_TRY(hipEventRecord(l.events[2 * i], l.stream), "hipEventRecord");
        if (int rc = launchTraverse(ctx, l, l.stream, scene, env, d_rays, d_results, count)) return rc;
        HIP_TRY(hipEventRecord(l.events[2 * i + 1], l.stream), "hipEventRecord");      // brackets the traversal kernel alone
        if (int rc = launchEnvShade(ctx, l, l.stream, env, d_results, count)) return rc;
    }
    HIP_TRY(hipStreamSynchronize(l.stream), "hipStreamSynchronize");
    for (uint32_t i = 0; i < iters; ++i)
        HIP_TRY(hipEventElapsedTime(&ms[i], l.events[2 * i], l.events[2 * i + 1]), "hipEventElapsedTime");
    l.info.last_kernel_ms = ms[iters - 1];
    return checkWatchdog(ctx);
}

int racc_hip_read_kernel_times(racc_hip_ctx* ctx, uint32_t lane, float* ms, uint32_t capacity, uint32_t* n) {
    if (int rc = checkLane(ctx, lane)) return rc;
    if (!n || (capacity && !ms)) return fail(RACC_HIP_ERR_INVALID, "read_kernel_times: bad argument");
    *n = 0;
    if (!ctx->opts.time_kernels) return fail(RACC_HIP_ERR_INVALID, "read_kernel_times: the context was created without time_kernels");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    Lane& l = ctx->lanes[lane];
    std::lock_guard<std::mutex> guard(l.mutex);
    const uint32_t have = l.ringCount < capacity ? l.ringCount : capacity;
    for (uint32_t i = 0; i < have; ++i) {                     // oldest of the most recent `have` first
        const uint32_t slot = (l.ringHead + kTimeRing - have + i) % kTimeRing;
        HIP_TRY(hipEventSynchronize(l.ring[2 * slot + 1]), "hipEventSynchronize");
        HIP_TRY(hipEventElapsedTime(&ms[i], l.ring[2 * slot], l.ring[2 * slot + 1]), "hipEventElapsedTime");
    }
    *n = have;
    l.ringCount = 0;
    return RACC_HIP_OK;
}

int racc_hip_get_launch_info(racc_hip_ctx* ctx, uint32_t lane, racc_hip_launch_info* info) {
    if (int rc = checkLane(ctx, lane)) return rc;
    if (!info) return fail(RACC_HIP_ERR_INVALID, "info is NULL");
    *info = ctx->lanes[lane].info;
    return RACC_HIP_OK;
}

int racc_hip_read_stats(racc_hip_ctx* ctx, uint32_t lane, uint64_t* stats8 /* [16] */, int reset) {
    if (int rc = checkLane(ctx, lane)) return rc;
    if (!stats8) return fail(RACC_HIP_ERR_INVALID, "stats8 is NULL");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    Lane& l = ctx->lanes[lane];
    HIP_TRY(hipStreamSynchronize(l.stream), "hipStreamSynchronize");
    HIP_TRY(hipMemcpy(stats8, l.cursor + 8, 128, hipMemcpyDeviceToHost), "hipMemcpy stats");
    if (reset) HIP_TRY(hipMemset(l.cursor + 8, 0, 128), "hipMemset stats");
    return RACC_HIP_OK;
}

int racc_hip_malloc(racc_hip_ctx* ctx, uint64_t bytes, void** d_ptr) {
    if (!ctx || !d_ptr) return fail(RACC_HIP_ERR_INVALID, "malloc: bad argument");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 1), "hipMalloc");
    return RACC_HIP_OK;
}

int racc_hip_free(racc_hip_ctx* ctx, void* d_ptr) {
    if (!ctx) return fail(RACC_HIP_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    if (d_ptr) HIP_TRY(hipFree(d_ptr), "hipFree");
    return RACC_HIP_OK;
}

int racc_hip_stream_create(racc_hip_ctx* ctx, void** stream) {
    if (!ctx || !stream) return fail(RACC_HIP_ERR_INVALID, "stream_create: bad argument");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    hipStream_t st = nullptr;
    HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
    *stream = st;
    return RACC_HIP_OK;
}

int racc_hip_stream_synchronize(racc_hip_ctx* ctx, void* stream) {
    if (!ctx) return fail(RACC_HIP_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    HIP_TRY(hipStreamSynchronize(static_cast<hipStream_t>(stream)), "hipStreamSynchronize");
    return checkWatchdog(ctx);
}

int racc_hip_stream_destroy(racc_hip_ctx* ctx, void* stream) {
    if (!ctx) return fail(RACC_HIP_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    if (stream) HIP_TRY(hipStreamDestroy(static_cast<hipStream_t>(stream)), "hipStreamDestroy");
    return RACC_HIP_OK;
}

int racc_hip_memcpy_h2d(racc_hip_ctx* ctx, void* d_dst, const void* src, uint64_t bytes) {
    if (!ctx || !d_dst || !src) return fail(RACC_HIP_ERR_INVALID, "memcpy_h2d: bad argument");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    HIP_TRY(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice), "hipMemcpy H2D");
    return RACC_HIP_OK;
}

int racc_hip_memcpy_d2h(racc_hip_ctx* ctx, void* dst, const void* d_src, uint64_t bytes) {
    if (!ctx || !dst || !d_src) return fail(RACC_HIP_ERR_INVALID, "memcpy_d2h: bad argument");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    HIP_TRY(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H");
    return RACC_HIP_OK;
}

int racc_hip_memcpy_d2d_async(racc_hip_ctx* ctx, void* d_dst, const void* d_src, uint64_t bytes, void* stream) {
    if (!ctx || !d_dst || !d_src) return fail(RACC_HIP_ERR_INVALID, "memcpy_d2d_async: bad argument");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    HIP_TRY(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)), "hipMemcpyAsync D2D");
    return RACC_HIP_OK;
}

// ---- RCCL (librccl, = NCCL's API over xGMI), bound at run time: the engine itself has no link dependency on it ----------
namespace {
struct RcclApi {
    void* lib = nullptr;
    int (*getUniqueId)(void*) = nullptr;
    int (*commInitRank)(void**, int, racc_hip_comm_id, int) = nullptr;
    int (*allGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*commDestroy)(void*) = nullptr;
    const char* (*errorString)(int) = nullptr;
};
RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy the process already mapped (torch.distributed ships its own: one RCCL instance per process, not two) first,
        // then the ROCm one
        if (FILE* maps = std::fopen("/proc/self/maps", "r")) {
            char line[1024];
            while (!api.lib && std::fgets(line, sizeof(line), maps)) {
                char* path = std::strchr(line, '/');
                if (!path || !std::strstr(path, "librccl")) continue;
                path[std::strcspn(path, "\n")] = 0;
                api.lib = dlopen(path, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
            }
            std::fclose(maps);
        }
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
        for (const char* n : names) if (!api.lib) api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!api.lib) return;
        api.getUniqueId = reinterpret_cast<int (*)(void*)>(dlsym(api.lib, "ncclGetUniqueId"));
        api.commInitRank = reinterpret_cast<int (*)(void**, int, racc_hip_comm_id, int)>(dlsym(api.lib, "ncclCommInitRank"));
        api.allGather = reinterpret_cast<int (*)(const void*, void*, size_t, int, void*, hipStream_t)>(dlsym(api.lib, "ncclAllGather"));
        api.commDestroy = reinterpret_cast<int (*)(void*)>(dlsym(api.lib, "ncclCommDestroy"));
        api.errorString = reinterpret_cast<const char* (*)(int)>(dlsym(api.lib, "ncclGetErrorString"));
    });
    return (api.lib && api.getUniqueId && api.commInitRank && api.allGather && api.commDestroy) ? &api : nullptr;
}
int failRccl(const char* what, int rc) {
    RcclApi* r = rccl();
    snprintf(g_msg, sizeof(g_msg), "racc_hip: %s: %s", what, r && r->errorString ? r->errorString(rc) : "librccl not available");
    racc_hip_set_error_(g_msg);
    return RACC_HIP_ERR_DEVICE;
}
}  // namespace

struct racc_hip_comm {
    racc_hip_ctx* ctx = nullptr;
    void* comm = nullptr;          // ncclComm_t
    int rank = 0, nranks = 1;
};

int racc_hip_comm_unique_id(racc_hip_comm_id* id) {
    if (!id) return fail(RACC_HIP_ERR_INVALID, "id is NULL");
    RcclApi* r = rccl();
    if (!r) return failRccl("ncclGetUniqueId", 0);
    if (int rc = r->getUniqueId(id)) return failRccl("ncclGetUniqueId", rc);
    return RACC_HIP_OK;
}

int racc_hip_comm_init_rank(racc_hip_ctx* ctx, const racc_hip_comm_id* id, int rank, int nranks, racc_hip_comm** out) {
    if (!ctx || !id || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(RACC_HIP_ERR_INVALID, "comm_init_rank: bad argument");
    *out = nullptr;
    RcclApi* r = rccl();
    if (!r) return failRccl("ncclCommInitRank", 0);
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    racc_hip_comm* c = new (std::nothrow) racc_hip_comm();
    if (!c) return fail(RACC_HIP_ERR_NOMEM, "out of host memory");
    c->ctx = ctx; c->rank = rank; c->nranks = nranks;
    if (int rc = r->commInitRank(&c->comm, nranks, *id, rank)) { delete c; return failRccl("ncclCommInitRank", rc); }
    *out = c;
    return RACC_HIP_OK;
}

int racc_hip_allgather_results(racc_hip_comm* comm, const void* d_send, void* d_recv, uint32_t count_per_rank, void* stream) {
    if (!comm || !d_send || !d_recv) return fail(RACC_HIP_ERR_INVALID, "allgather_results: bad argument");
    if (!count_per_rank) return RACC_HIP_OK;
    RcclApi* r = rccl();
    if (!r) return failRccl("ncclAllGather", 0);
    HIP_TRY(hipSetDevice(comm->ctx->device), "hipSetDevice");
    // a Result is 16 bytes = 4 x u32 (ncclUint32 = 3); one message per rank, as large as the shard (ring collectives over
    // point-to-point xGMI are per-link bound: few large messages, never one per ray stream)
    if (int rc = r->allGather(d_send, d_recv, size_t(count_per_rank) * 4u, 3, comm->comm, static_cast<hipStream_t>(stream))) return failRccl("ncclAllGather", rc);
    return RACC_HIP_OK;
}

int racc_hip_comm_destroy(racc_hip_comm* comm) {
    if (!comm) return RACC_HIP_OK;
    RcclApi* r = rccl();
    if (r && comm->comm) { hipSetDevice(comm->ctx->device); r->commDestroy(comm->comm); }
    delete comm;
    return RACC_HIP_OK;
}

// ---- device groups: the GPUs of one node behind one handle --------------------------------------------------------------
// The path shards with no exchange (rays never interact, the scene is read-only, Scene.cpp:342-346): a group is n engine
// contexts, the scene and environment replicated on each, a batch cut into n contiguous shards (multiples of 64 rays, one
// wave's chunk) traced concurrently, results in place.  Entries of `devices` may repeat an ordinal (rehearsal on one GPU).
// One persistent host thread per member (a member = one GPU): every call hands its per-member work to that thread, so the n GPUs
// are driven concurrently without creating threads per call, and each member's HIP calls stay on one thread.
namespace {
struct GroupWorker {
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::deque<std::function<void()>> q;
    uint64_t posted = 0, finished = 0;
    bool stop = false;
    int rc = RACC_HIP_OK;              // first failure of a posted job since the last collect()
    std::string msg;
    void run() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return stop || !q.empty(); });
            if (q.empty()) return;      // stop, and nothing left
            std::function<void()> job = std::move(q.front());
            q.pop_front();
            lk.unlock();
            job();
            lk.lock();
            ++finished;
            cv.notify_all();
        }
    }
    void post(std::function<void()> job) {
        { std::lock_guard<std::mutex> g(m); q.push_back(std::move(job)); ++posted; }
        cv.notify_all();
    }
    void drain() { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return finished == posted; }); }
    void note(int code) {               // called from a job, on the worker thread: keeps the first failure and its thread-local text
        if (code == RACC_HIP_OK) return;
        std::lock_guard<std::mutex> g(m);
        if (rc == RACC_HIP_OK) { rc = code; msg = racc_hip_last_error(); }
    }
    int collect(std::string& text) { std::lock_guard<std::mutex> g(m); const int r = rc; if (r != RACC_HIP_OK) text = msg; rc = RACC_HIP_OK; msg.clear(); return r; }
};
}  // namespace

struct racc_hip_group {
    std::vector<racc_hip_ctx*> ctx;
    std::vector<std::unique_ptr<GroupWorker>> worker;
};
struct racc_hip_group_scene { std::vector<racc_hip_scene*> scene; };
struct racc_hip_group_env { std::vector<racc_hip_env*> env; };

int racc_hip_group_create(const int* devices, uint32_t n, const racc_hip_options* opts, racc_hip_group** out) {
    if (!devices || !n || n > 64u || !out) return fail(RACC_HIP_ERR_INVALID, "group_create: bad argument");
    *out = nullptr;
    racc_hip_group* g = new (std::nothrow) racc_hip_group();
    if (!g) return fail(RACC_HIP_ERR_NOMEM, "out of host memory");
    for (uint32_t i = 0; i < n; ++i) {
        racc_hip_ctx* c = nullptr;
        if (int rc = racc_hip_create(devices[i], opts, &c)) { racc_hip_group_destroy(g); return rc; }
        g->ctx.push_back(c);
        g->worker.emplace_back(new GroupWorker());
        GroupWorker* w = g->worker.back().get();
        w->th = std::thread([w] { w->run(); });
    }
    *out = g;
    return RACC_HIP_OK;
}

int racc_hip_group_destroy(racc_hip_group* g) {
    if (!g) return RACC_HIP_OK;
    for (auto& w : g->worker) {
        { std::lock_guard<std::mutex> lk(w->m); w->stop = true; }
        w->cv.notify_all();
        if (w->th.joinable()) w->th.join();
    }
    for (racc_hip_ctx* c : g->ctx) racc_hip_destroy(c);
    delete g;
    return RACC_HIP_OK;
}

uint32_t racc_hip_group_size(const racc_hip_group* g) { return g ? uint32_t(g->ctx.size()) : 0u; }

racc_hip_ctx* racc_hip_group_ctx(racc_hip_group* g, uint32_t i) { return g && i < g->ctx.size() ? g->ctx[i] : nullptr; }

int racc_hip_group_scene_upload(racc_hip_group* g, const void* nodes64, uint32_t node_count, const void* pairs48, uint32_t pair_count,
                                const uint32_t* remap, uint32_t remap_count, racc_hip_group_scene** out) {
    if (!g || !out) return fail(RACC_HIP_ERR_INVALID, "group/out is NULL");
    *out = nullptr;
    racc_hip_group_scene* s = new (std::nothrow) racc_hip_group_scene();
    if (!s) return fail(RACC_HIP_ERR_NOMEM, "out of host memory");
    for (racc_hip_ctx* c : g->ctx) {
        racc_hip_scene* one = nullptr;
        if (int rc = racc_hip_scene_upload(c, nodes64, node_count, pairs48, pair_count, remap, remap_count, &one)) { racc_hip_group_scene_free(g, s); return rc; }
        s->scene.push_back(one);
    }
    *out = s;
    return RACC_HIP_OK;
}

int racc_hip_group_scene_free(racc_hip_group* g, racc_hip_group_scene* s) {
    if (!s) return RACC_HIP_OK;
    for (size_t i = 0; i < s->scene.size(); ++i) racc_hip_scene_free(g && i < g->ctx.size() ? g->ctx[i] : nullptr, s->scene[i]);
    delete s;
    return RACC_HIP_OK;
}

int racc_hip_group_env_upload(racc_hip_group* g, const float* rgba, uint32_t width, uint32_t height, racc_hip_group_env** out) {
    if (!g || !out) return fail(RACC_HIP_ERR_INVALID, "group/out is NULL");
    *out = nullptr;
    racc_hip_group_env* e = new (std::nothrow) racc_hip_group_env();
    if (!e) return fail(RACC_HIP_ERR_NOMEM, "out of host memory");
    for (racc_hip_ctx* c : g->ctx) {
        racc_hip_env* one = nullptr;
        if (int rc = racc_hip_env_upload(c, rgba, width, height, &one)) { racc_hip_group_env_free(g, e); return rc; }
        e->env.push_back(one);
    }
    *out = e;
    return RACC_HIP_OK;
}

int racc_hip_group_env_free(racc_hip_group* g, racc_hip_group_env* e) {
    if (!e) return RACC_HIP_OK;
    for (size_t i = 0; i < e->env.size(); ++i) racc_hip_env_free(g && i < g->ctx.size() ? g->ctx[i] : nullptr, e->env[i]);
    delete e;
    return RACC_HIP_OK;
}

namespace {
int groupCollect(racc_hip_group* g) {       // after the workers have drained: the first member's failure, if any
    int rc = RACC_HIP_OK; std::string text;
    for (auto& w : g->worker) { std::string t; const int r = w->collect(t); if (r != RACC_HIP_OK && rc == RACC_HIP_OK) { rc = r; text = t; } }
    return rc == RACC_HIP_OK ? RACC_HIP_OK : fail(rc, text.c_str());
}
}  // namespace

int racc_hip_group_intersect(racc_hip_group* g, const racc_hip_group_scene* scene, const racc_hip_group_env* env,
                             const void* rays, void* results, uint32_t count) {
    if (!g || !scene || scene->scene.size() != g->ctx.size() || (env && env->env.size() != g->ctx.size()))
        return fail(RACC_HIP_ERR_INVALID, "group_intersect: the scene/environment does not belong to this group");
    if (!count) return RACC_HIP_OK;
    if (!rays || !results) return fail(RACC_HIP_ERR_INVALID, "rays/results is NULL");
    const uint32_t n = uint32_t(g->ctx.size());
    const uint32_t per = ((count + n - 1u) / n + 63u) / 64u * 64u;        // contiguous shards, whole chunks of 64 rays
    for (uint32_t i = 0; i < n; ++i) {
        const uint64_t b = uint64_t(i) * per;
        if (b >= count) break;
        const uint32_t cnt = uint32_t(uint64_t(count) - b < per ? uint64_t(count) - b : per);
        GroupWorker* w = g->worker[i].get();
        racc_hip_ctx* c = g->ctx[i];
        const racc_hip_scene* sc = scene->scene[i];
        const racc_hip_env* ev = env ? env->env[i] : nullptr;
        w->post([=] {      // the member's thread and PCIe link: the blocking entry copies in, traces, copies out
            w->note(racc_hip_intersect(c, sc, ev, static_cast<const char*>(rays) + b * 32, static_cast<char*>(results) + b * 16, cnt, 0));
        });
    }
    for (auto& w : g->worker) w->drain();
    return groupCollect(g);
}

int racc_hip_group_intersect_device(racc_hip_group* g, const racc_hip_group_scene* scene, const racc_hip_group_env* env,
                                    const void* const* d_rays, void* const* d_results, const uint32_t* counts) {
    if (!g || !scene || scene->scene.size() != g->ctx.size() || (env && env->env.size() != g->ctx.size()))
        return fail(RACC_HIP_ERR_INVALID, "group_intersect_device: the scene/environment does not belong to this group");
    if (!d_rays || !d_results || !counts) return fail(RACC_HIP_ERR_INVALID, "group_intersect_device: NULL array");
    const uint32_t n = uint32_t(g->ctx.size());
    for (uint32_t i = 0; i < n; ++i)
        if (counts[i] && (!d_rays[i] || !d_results[i])) return fail(RACC_HIP_ERR_INVALID, "group_intersect_device: NULL shard");
    for (uint32_t i = 0; i < n; ++i) {
        if (!counts[i]) continue;
        GroupWorker* w = g->worker[i].get();
        racc_hip_ctx* c = g->ctx[i];
        const racc_hip_scene* sc = scene->scene[i];
        const racc_hip_env* ev = env ? env->env[i] : nullptr;
        const void* r = d_rays[i]; void* o = d_results[i]; const uint32_t cnt = counts[i];
        // the member's engine context, its own streams: lanes rotated, launches chained (racc_hip_intersect_device)
        w->post([=] { w->note(racc_hip_intersect_device(c, sc, ev, r, o, cnt, RACC_HIP_LANE_AUTO, nullptr)); });
    }
    return RACC_HIP_OK;
}

int racc_hip_group_wait(racc_hip_group* g) {
    if (!g) return fail(RACC_HIP_ERR_INVALID, "group is NULL");
    for (size_t i = 0; i < g->ctx.size(); ++i) {
        GroupWorker* w = g->worker[i].get();
        racc_hip_ctx* c = g->ctx[i];
        w->post([=] { w->note(racc_hip_wait(c, RACC_HIP_LANE_AUTO)); });      // (behind the member's issued batches: the worker runs its jobs in order)
    }
    for (auto& w : g->worker) w->drain();
    return groupCollect(g);
}


int racc_hip_synchronize(racc_hip_ctx* ctx) {
    if (!ctx) return fail(RACC_HIP_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    HIP_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
    return checkWatchdog(ctx);
}

}  // extern "C"
