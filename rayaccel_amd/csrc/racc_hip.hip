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
//   * the per-ray traversal stack lives in LDS as [level][thread] (bank = thread % 32 at every level, so pushes/pops never
//     conflict), sized from the real tree height at upload time, with a global-memory spill instantiation for tall trees
//     — unlike the reference's unchecked stack[64] it cannot overflow;
//   * hit epilogues (remap gather + barycentric rotation) are batched into the refill step; miss radiance is evaluated by
//     a second, streaming kernel (envShadeKernel).
// traverseKernelV2 below is the shipped kernel (thin-wave drain policy, lazy epilogues, deferred miss shading, static
// first chunk, while-while inner repeats).  Three other generations — V1 (the first correct kernel, optional LDS cache of
// the top of the tree), V3 (workgroup-wide regrouping through LDS) and V4 (two rays per lane) — live in
// racc_kernels_experimental.inc and stay selectable through racc_hip_options::kernel_variant.
// Arithmetic is IEEE binary32 with explicit fmaf only (built with -ffp-contract=off, no fast-math), the same evaluation
// order as oracle/racc_oracle.c, so primId/t/u/v are bit-identical to the CPU restatement for every finite ray.  The
// traversal ORDER is the reference's (nearer child first, far child pushed only if both hit, pairs of a leaf in order),
// which is what makes ties resolve identically.

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <queue>
#include <utility>
#include <vector>

#include "racc_hip.h"

extern "C" void racc_hip_set_error_(const char* msg);

namespace {

constexpr int kCacheMax = 4096;    // nodes re-ordered to the front of the device array (upper bound of any variant's LDS cache)
constexpr uint32_t kInvalidTriangle = 0xFFFFFFFFu;
constexpr uint32_t kLeafBase = 0x1000000u;   // node refs below this carry no work (done / empty lane)

struct TraverseArgs {
    const float4* rays;
    float4* results;
    uint32_t count;
    const float4* nodes;      // 64 B device records (see slabPair); the first kCacheMax are the largest-area
                              // top of the tree (see reorderNodes)
    uint32_t cacheCount;      // nodes [0, cacheCount) are also resident in LDS
    uint32_t nodeBytes, pairBytes;   // extents for the buffer descriptors (V2 BUF ablation)
    const float4* nodesSoa;   // SOA ablation only: the same records transposed into 4 planes of nodeCount float4 each
    uint32_t nodeCount;
    const float4* pairs;      // 3 x float4 per pair (Scene.cpp:83-87 order)
    const uint32_t* remap;
    const float4* env;        // RGBA32F probe image or nullptr
    uint32_t envW, envH;
    uint32_t* cursor;         // [0] ray cursor, [1] finished-block counter
    uint32_t* trips;          // watchdog trips: one word of host-mapped memory per context, so the host sees it without a copy
    uint32_t* spill;          // [spillLevels][gridThreads]
    uint32_t spillStride;     // gridThreads
    uint32_t chunk;           // rays per cursor dequeue
    uint32_t refillMin;       // idle lanes that trigger a refill
    uint32_t leafMin;         // leaf lanes that trigger a leaf step
    uint32_t maxIters;        // watchdog: a wave gives up after this many scheduling iterations
    uint32_t regroup;         // V3: scheduling iterations between workgroup-wide regroupings
    uint32_t tailActive;      // V2: waves with at most this many live rays run inner AND leaf bodies every iteration
    uint32_t thinReps;        // V2: inner steps per scheduling iteration in such waves
    uint32_t innerReps;       // V2: inner steps per scheduling iteration in all other waves
    unsigned long long* stats;   // STATS builds only: [0] inner iters [1] inner lanes [2] leaf iters [3] leaf lanes
                                 //                    [4] refill iters [5] rays loaded [6] dequeues [7] waves
};

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz) {
    return __builtin_fmaf(az, bz, __builtin_fmaf(ay, by, ax * bx));
}

// Device node record (64 B; written by reorderNodes): words 0-1 = child refs (first, last), 2-3 unused,
//   float4 #1 = (Lmin.x, Lmax.x, Lmin.y, Lmax.y)   #2 = (Lmin.z, Lmax.z, Rmin.x, Rmax.x)   #3 = (Rmin.y, Rmax.y, Rmin.z, Rmax.z)
// i.e. each min/max plane pair is one aligned register pair, so the six "a = fma(min, inv, ood); b = fma(max, inv, ood)"
// of the two slab tests (Kernels.h:122-123) are six v_pk_fma_f32 instead of twelve v_fma_f32 — same IEEE results.
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void slabPair(const float4 q1, const float4 q2, const float4 q3,
                                         float ix, float iy, float iz, float ex, float ey, float ez,
                                         float tNear, float tFar, float& tFirst, float& tLast) {
    const f32x2 vix = {ix, ix}, viy = {iy, iy}, viz = {iz, iz}, vex = {ex, ex}, vey = {ey, ey}, vez = {ez, ez};
    const f32x2 lx = __builtin_elementwise_fma((f32x2){q1.x, q1.y}, vix, vex);
    const f32x2 ly = __builtin_elementwise_fma((f32x2){q1.z, q1.w}, viy, vey);
    const f32x2 lz = __builtin_elementwise_fma((f32x2){q2.x, q2.y}, viz, vez);
    const f32x2 rx = __builtin_elementwise_fma((f32x2){q2.z, q2.w}, vix, vex);
    const f32x2 ry = __builtin_elementwise_fma((f32x2){q3.x, q3.y}, viy, vey);
    const f32x2 rz = __builtin_elementwise_fma((f32x2){q3.z, q3.w}, viz, vez);
    const float l0 = fmaxf(fmaxf(tNear, fminf(lx.x, lx.y)), fmaxf(fminf(ly.x, ly.y), fminf(lz.x, lz.y)));
    const float l1 = fminf(fminf(tFar, fmaxf(lx.x, lx.y)), fminf(fmaxf(ly.x, ly.y), fmaxf(lz.x, lz.y)));
    const float r0 = fmaxf(fmaxf(tNear, fminf(rx.x, rx.y)), fmaxf(fminf(ry.x, ry.y), fminf(rz.x, rz.y)));
    const float r1 = fminf(fminf(tFar, fmaxf(rx.x, rx.y)), fminf(fmaxf(ry.x, ry.y), fmaxf(rz.x, rz.y)));
    tFirst = (l0 > l1) ? tFar : l0;      // Kernels.h:131-134: tFar doubles as the "missed" sentinel
    tLast = (r0 > r1) ? tFar : r0;
}

struct LaneRay {
    float ox, oy, oz, dx, dy, dz;       // origin, (clamped) direction
    float ix, iy, iz, ex, ey, ez;       // 1/dir, -origin/dir
    float tNear, tFar;
    int hitIndex;                       // pair*2 + which, or -1
    float hitU, hitV;
};

// Triangle-pair test, Kernels.h:36-115.  Updates the lane's hit and returns the new tFar.
__device__ __forceinline__ float pairIntersectData(const float4 t0, const float4 t1, const float4 t2, uint32_t index, LaneRay& r) {
    const float tNear = r.tNear, tMax = r.tFar;

    // n1 = e1 x e2, n2 = e3 x e1 (mad_cross, Kernels.h:23-25)
    const float n1x = __builtin_fmaf(t0.y, t1.z, -(t0.z * t1.y));
    const float n1y = __builtin_fmaf(t0.z, t1.x, -(t0.x * t1.z));
    const float n1z = __builtin_fmaf(t0.x, t1.y, -(t0.y * t1.x));
    const float n2x = __builtin_fmaf(t1.w, t0.z, -(t2.w * t0.y));
    const float n2y = __builtin_fmaf(t2.w, t0.x, -(t0.w * t0.z));
    const float n2z = __builtin_fmaf(t0.w, t0.y, -(t1.w * t0.x));
    const float cx = t2.x - r.ox, cy = t2.y - r.oy, cz = t2.z - r.oz;
    const float rx = __builtin_fmaf(r.dy, cz, -(r.dz * cy));
    const float ry = __builtin_fmaf(r.dz, cx, -(r.dx * cz));
    const float rz = __builtin_fmaf(r.dx, cy, -(r.dy * cx));

    const float det1 = dot3(n1x, n1y, n1z, r.dx, r.dy, r.dz);
    const float det2 = dot3(n2x, n2y, n2z, r.dx, r.dy, r.dz);
    const uint32_t sgn1 = __float_as_uint(det1) & 0x80000000u;
    const uint32_t sgn2 = __float_as_uint(det2) & 0x80000000u;

    const float re1 = dot3(rx, ry, rz, t0.x, t0.y, t0.z);
    const uint32_t iU1 = __float_as_uint(dot3(rx, ry, rz, t1.x, t1.y, t1.z)) ^ sgn1;
    const uint32_t iV1 = __float_as_uint(re1) ^ sgn1;
    const uint32_t iU2 = __float_as_uint(-re1) ^ sgn2;
    const uint32_t iV2 = __float_as_uint(-dot3(rx, ry, rz, t0.w, t1.w, t2.w)) ^ sgn2;

    bool outside1 = int(iU1 | iV1) < 0;
    bool outside2 = int(iU2 | iV2) < 0;

    float U1 = __uint_as_float(iU1), V1 = __uint_as_float(iV1);
    const float U2 = __uint_as_float(iU2), V2 = __uint_as_float(iV2);
    float absDet1 = fabsf(det1);
    const float absDet2 = fabsf(det2);
    const float W1 = absDet1 - U1 - V1;
    const float W2 = absDet2 - U2 - V2;
    float T1 = __uint_as_float(__float_as_uint(dot3(n1x, n1y, n1z, cx, cy, cz)) ^ sgn1);
    const float T2 = __uint_as_float(__float_as_uint(dot3(n2x, n2y, n2z, cx, cy, cz)) ^ sgn2);

    outside1 = outside1 || (W1 < 0.0f || T1 <= absDet1 * tNear || T1 > absDet1 * tMax);
    outside2 = outside2 || (W2 < 0.0f || T2 <= absDet2 * tNear || T2 > absDet2 * tMax);
    if (outside1 && outside2) return tMax;

    uint32_t which = 0;
    if ((!outside2 && outside1) || (!outside1 && !outside2 && T1 * absDet2 > T2 * absDet1)) {
        absDet1 = absDet2; T1 = T2; U1 = U2; V1 = V2;
        which = 1;
    }
    const float rcp = 1.0f / absDet1;      // correctly rounded (reference: native_recip, Kernels.h:107)
    const float t = T1 * rcp;
    r.hitIndex = int(index * 2 + which);
    r.hitU = U1 * rcp;
    r.hitV = V1 * rcp;
    return t;
}

__device__ __forceinline__ float pairIntersect(const float4* __restrict__ pairs, uint32_t index, LaneRay& r) {
    return pairIntersectData(pairs[index * 3 + 0], pairs[index * 3 + 1], pairs[index * 3 + 2], index, r);
}

// Miss colour, Kernels.h:213-222 with OpenCL CLAMP_TO_EDGE | FILTER_LINEAR on normalized coordinates.
__device__ __forceinline__ float4 envSample(const float4* __restrict__ env, uint32_t envW, uint32_t envH,
                                             float dx, float dy, float dz) {
    float4 out = make_float4(__uint_as_float(kInvalidTriangle), 0.0f, 0.0f, 0.0f);
    if (!env) return out;
    const float rlen = 1.0f / sqrtf(__builtin_fmaf(dz, dz, dy * dy));
    float r = (rlen > 1e+6f) ? 0.0f : acosf(-dx) * (1.0f / (2.0f * 3.141593f)) * rlen;
    if (!isfinite(r)) r = 0.0f;
    const float u = 0.5f - r * dz, v = 0.5f - r * dy;
    const float fx = u * float(envW) - 0.5f, fy = v * float(envH) - 0.5f;
    const float flx = floorf(fx), fly = floorf(fy);
    const float a = fx - flx, b = fy - fly;
    const int w1 = int(envW) - 1, h1 = int(envH) - 1;
    const int x0 = min(max(int(flx), 0), w1), x1 = min(max(int(flx) + 1, 0), w1);
    const int y0 = min(max(int(fly), 0), h1), y1 = min(max(int(fly) + 1, 0), h1);
    const float4 t00 = env[size_t(y0) * envW + x0], t10 = env[size_t(y0) * envW + x1];
    const float4 t01 = env[size_t(y1) * envW + x0], t11 = env[size_t(y1) * envW + x1];
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    out.y = w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x;
    out.z = w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y;
    out.w = w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z;
    return out;
}

__device__ __forceinline__ uint32_t laneRank(uint64_t mask) {   // # set bits below this lane: prefix sum of the ballot
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

// ================================================================================================ V2
// Same algorithm and arithmetic as traverseKernel, with the per-iteration dependent chain shortened:
//   * node / pair fetches are buffer loads (SGPR descriptor + 32-bit byte offset = ref << 6 / first * 48):
//     no 64-bit address arithmetic on the critical path;
//   * the stack keeps a register copy of its top entry: a pop takes the register and immediately issues the
//     ds_read of the NEXT entry, whose latency overlaps the following iteration (LDS holds every entry, so a push
//     never has to wait for that read);
//   * SPILL = false instantiations (tree height <= LDS_LEVELS) have no spill branches at all;
//   * finished rays wait for their epilogue as node == kDone; once the batch is exhausted the (expensive, divergent)
//     epilogue runs only when >= refillMin rays are pending or the wave has nothing else to do.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
constexpr uint32_t kEmpty = 0u, kDone = 1u;
constexpr uint32_t kXcdCursorWord = 40;   // cursor block: words 0-2 control, 8-39 statistics, 40-47 per-XCD ray cursors

__device__ __forceinline__ float4 asFloat4(u32x4 v) {
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

template <int BLOCK, int LDS_LEVELS, bool SPILL, bool STATS, bool BUF = true, bool TOS = true, bool XQ = false, bool SOA = false, bool PF = false>
__global__ void __launch_bounds__(BLOCK) traverseKernelV2(const TraverseArgs a) {
    __shared__ uint32_t lds[LDS_LEVELS * BLOCK];
    __shared__ uint32_t pfSink[PF ? 2 * BLOCK : 1];     // PF: where the touch loads land (never read)
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63u;
    uint32_t* const myLds = lds + tid;
    uint32_t* const mySpill = a.spill + (blockIdx.x * BLOCK + tid);
    const __amdgpu_buffer_rsrc_t nodeRsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.nodes), 0, a.nodeBytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t pairRsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(a.pairs), 0, a.pairBytes, 0x00020000);

    LaneRay r;
    r.ox = r.oy = r.oz = r.dx = r.dy = r.dz = 0.0f;
    r.ix = r.iy = r.iz = r.ex = r.ey = r.ez = 0.0f;
    r.tNear = r.tFar = 0.0f; r.hitIndex = -1; r.hitU = r.hitV = 0.0f;
    uint32_t rayIdx = 0;
    uint32_t node = kEmpty;     // bit31: inner ref | >= kLeafBase: leaf, pairs pending | kDone: awaiting epilogue | kEmpty
    uint32_t sp = 0, top = 0;   // stack height; register copy of entry sp-1
    // A wave's first chunk is static — its own index in the grid — and the shared cursor hands out everything after those:
    // 5,120 waves hitting one address at launch queue up behind each other (same-address atomics retire at about one per
    // 10 ns on this part, i.e. 50 us until the last wave had work).
    // (readfirstlane: the wave index is uniform, but only this tells the compiler — otherwise wBeg/wEnd/exhausted and the
    // whole loop control turn into exec-masked vector code: measured +7 % per ray)
    uint32_t wBeg = XQ ? 0u : min((blockIdx.x * uint32_t(BLOCK / 64) + uint32_t(__builtin_amdgcn_readfirstlane(int(tid >> 6)))) * a.chunk, a.count);
    uint32_t wEnd = XQ ? 0u : min(wBeg + a.chunk, a.count);
    bool exhausted = false;
    uint32_t xqTried = 0;       // XQ: how many of the 8 per-XCD queues this wave has found empty
    uint32_t stInner = 0, stInnerLanes = 0, stLeaf = 0, stLeafLanes = 0, stRefill = 0, stLoaded = 0, stDeq = 0;
    unsigned long long cyInner = 0, cyLeaf = 0, cyRefill = 0, cyStart = 0;
    if (STATS) cyStart = __builtin_readcyclecounter();

#define RACC_PUSH(x)                                                                              \
    do {                                                                                          \
        const uint32_t v_ = (x);                                                                  \
        if (!SPILL || sp < uint32_t(LDS_LEVELS)) myLds[sp * BLOCK] = v_;                          \
        else mySpill[size_t(sp - LDS_LEVELS) * a.spillStride] = v_;                               \
        if (TOS) top = v_;                                                                        \
        ++sp;                                                                                     \
    } while (0)
#define RACC_POP_OR_DONE()                                                                        \
    do {                                                                                          \
        if (sp == 0u) { node = kDone; }                                                           \
        else if (!TOS) {                                                                          \
            --sp;                                                                                 \
            node = myLds[(SPILL ? min(sp, uint32_t(LDS_LEVELS - 1)) : sp) * BLOCK];               \
            if (SPILL && sp >= uint32_t(LDS_LEVELS)) node = mySpill[size_t(sp - LDS_LEVELS) * a.spillStride]; \
        } else {                                                                                  \
            node = top;                                                                           \
            --sp;                                                                                 \
            if (sp != 0u) {                                                                       \
                const uint32_t k_ = sp - 1u;                                                      \
                top = myLds[(SPILL ? min(k_, uint32_t(LDS_LEVELS - 1)) : k_) * BLOCK];            \
                if (SPILL && k_ >= uint32_t(LDS_LEVELS)) top = mySpill[size_t(k_ - LDS_LEVELS) * a.spillStride]; \
            }                                                                                     \
        }                                                                                         \
    } while (0)

    for (uint32_t iter = 0;; ++iter) {
        if (iter >= a.maxIters) {
            if (lane == 0) __hip_atomic_fetch_add(a.trips, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        unsigned long long cyTop = 0;
        if (STATS) cyTop = __builtin_readcyclecounter();
        const uint32_t nInner = __popcll(__ballot(int(node) < 0));
        const uint32_t nLeaf = __popcll(__ballot(int(node) >= int(kLeafBase)));
        const bool noWork = (nInner | nLeaf) == 0u;
        bool refill = noWork;
        if (!noWork) {
            if (!exhausted) refill = (64u - nInner - nLeaf) >= a.refillMin;
            else refill = uint32_t(__popcll(__ballot(node == kDone))) >= a.refillMin;
        }

        if (refill) {
            // ---------------- batched epilogue (Kernels.h:213-241) ----------------
            if (node == kDone) {
                float4 out;
                if (r.hitIndex < 0) {
                    // Miss: park the (clamped) direction in the rgb slots; envShadeKernel turns it into radiance right after
                    // this kernel on the same stream.  Keeps acosf + bilinear (~130 divergent instructions) out of the hot loop.
                    out = a.env ? make_float4(__uint_as_float(kInvalidTriangle), r.dx, r.dy, r.dz)
                                : make_float4(__uint_as_float(kInvalidTriangle), 0.0f, 0.0f, 0.0f);
                } else {
                    uint32_t m = a.remap[r.hitIndex];
                    const uint32_t edge = m >> 30;
                    m &= 0x3FFFFFFFu;
                    const float bx = r.hitU, by = r.hitV, bz = 1.0f - r.hitU - r.hitV;
                    float u = bx, v = by;
                    if (edge == 1u) { u = bz; v = bx; } else if (edge == 2u) { u = by; v = bz; }
                    out = make_float4(__uint_as_float(m), r.tFar, u, v);
                }
                a.results[rayIdx] = out;
                node = kEmpty;
            }
            // ---------------- refill: ballot + mbcnt prefix sum over the empty lanes ----------------
            const uint64_t emptyMask = __ballot(node == kEmpty);
            const uint32_t need = __popcll(emptyMask);
            if (STATS) ++stRefill;
            if (wBeg == wEnd && !exhausted) {
                if (STATS) ++stDeq;
                if (!XQ) {
                    // (a wave's FIRST chunk is static, see the initialisation of wBeg/wEnd: the cursor hands out the rest)
                    uint32_t b = 0;
                    if (lane == 0) b = atomicAdd(a.cursor, a.chunk);
                    const uint32_t r = __builtin_amdgcn_readfirstlane(b);
                    b = r + gridDim.x * uint32_t(BLOCK / 64) * a.chunk;
                    if (b < r) b = 0xFFFFFFFFu;      // 32-bit wrap: past any batch
                    exhausted = (b >= a.count) || (b + a.chunk < b);
                    wBeg = exhausted ? a.count : b;
                    wEnd = exhausted ? a.count : min(b + a.chunk, a.count);
                } else {
                    // XCD-partitioned queue: workgroup b runs on XCD b % 8 (round-robin dispatch), and each XCD has its own L2.
                    // The batch is cut into 8 contiguous eighths with a cursor each; a wave drains its own XCD's eighth first
                    // (neighbouring rays -> the same subtrees stay in that L2) and then steals from the others.
                    const uint32_t per = ((a.count + 7u) / 8u + a.chunk - 1u) / a.chunk * a.chunk;
                    for (;;) {
                        const uint32_t q = (blockIdx.x + xqTried) & 7u;
                        uint32_t b = 0;
                        if (lane == 0) b = atomicAdd(a.cursor + kXcdCursorWord + q, a.chunk);
                        b = __builtin_amdgcn_readfirstlane(b);
                        const uint64_t lo = uint64_t(q) * per + b;
                        const uint64_t hi = min(uint64_t(q + 1u) * per, uint64_t(a.count));
                        if (b < per && lo < hi) {
                            wBeg = uint32_t(lo);
                            wEnd = uint32_t(min(lo + a.chunk, hi));
                            break;
                        }
                        if (++xqTried == 8u) { exhausted = true; break; }
                    }
                }
            }
            const uint32_t take = min(need, wEnd - wBeg);
            const uint32_t rank = laneRank(emptyMask);
            if (node == kEmpty && rank < take) {
                const uint32_t idx = wBeg + rank;
                const float4 q0 = a.rays[size_t(idx) * 2 + 0];
                const float4 q1 = a.rays[size_t(idx) * 2 + 1];
                const bool valid = isfinite(q0.x) && isfinite(q0.y) && isfinite(q0.z) && isfinite(q0.w) &&
                                   isfinite(q1.x) && isfinite(q1.y) && isfinite(q1.z) && !isnan(q1.w);
                if (!valid) {   // NaN in the first slot tells envShadeKernel to leave rgb = 0
                    a.results[idx] = make_float4(__uint_as_float(kInvalidTriangle), a.env ? __uint_as_float(0x7FC00000u) : 0.0f, 0.0f, 0.0f);
                } else {
                    const float eps = 1e-10f;   // Kernels.h:149-157
                    r.ox = q0.x; r.oy = q0.y; r.oz = q0.z; r.tNear = q0.w;
                    r.dx = (fabsf(q1.x) < eps) ? copysignf(eps, q1.x) : q1.x;
                    r.dy = (fabsf(q1.y) < eps) ? copysignf(eps, q1.y) : q1.y;
                    r.dz = (fabsf(q1.z) < eps) ? copysignf(eps, q1.z) : q1.z;
                    r.tFar = q1.w;
                    r.ix = 1.0f / r.dx; r.iy = 1.0f / r.dy; r.iz = 1.0f / r.dz;    // Kernels.h:159-160
                    r.ex = -r.ox * r.ix; r.ey = -r.oy * r.iy; r.ez = -r.oz * r.iz;
                    r.hitIndex = -1; r.hitU = 0.0f; r.hitV = 0.0f;
                    rayIdx = idx;
                    node = 0x80000000u;     // Kernels.h:164
                    sp = 0;
                }
            }
            wBeg += take;
            if (STATS) { stLoaded += take; cyRefill += __builtin_readcyclecounter() - cyTop; }
            if (exhausted && wBeg == wEnd && __ballot(node != kEmpty) == 0ull) break;
            continue;
        }

        // Vote: a leaf step when enough lanes wait at a leaf — `leafMin` of them in a full wave, a quarter of the active
        // lanes in a thin one (the launch tail, where a lone long ray must not idle behind the vote) — or when no lane
        // holds an inner node.  Thin waves (<= tailActive rays) run both bodies per iteration: they are latency-bound.
        const uint32_t nActive = nInner + nLeaf;
        const bool doLeaf = nLeaf >= a.leafMin || nInner == 0u || nLeaf * 4u >= nActive;
        const bool doInner = nInner != 0u && (!doLeaf || nActive <= a.tailActive);
        if (doLeaf) {
            // ---------------- leaf step (Kernels.h:200-205 + 36-115) ----------------
            if (STATS) { ++stLeaf; stLeafLanes += nLeaf; }
            if (int(node) >= int(kLeafBase)) {
                const uint32_t cur = node & 0xFFFFFFu;
                const uint32_t cnt = node >> 24;
                const uint32_t off = cur * 48u;
                float4 t0, t1, t2;
                if (BUF) {
                    t0 = asFloat4(__builtin_amdgcn_raw_buffer_load_b128(pairRsrc, off, 0, 0));
                    t1 = asFloat4(__builtin_amdgcn_raw_buffer_load_b128(pairRsrc, off + 16u, 0, 0));
                    t2 = asFloat4(__builtin_amdgcn_raw_buffer_load_b128(pairRsrc, off + 32u, 0, 0));
                } else {
                    t0 = a.pairs[cur * 3u]; t1 = a.pairs[cur * 3u + 1u]; t2 = a.pairs[cur * 3u + 2u];
                }
                r.tFar = pairIntersectData(t0, t1, t2, cur, r);
                if (cnt > 1u) node = ((cnt - 1u) << 24) | (cur + 1u);
                else RACC_POP_OR_DONE();
            }
            if (STATS) { cyLeaf += __builtin_readcyclecounter() - cyTop; cyTop = __builtin_readcyclecounter(); }
        }
        if (doInner) {
            // ---------------- inner step (Kernels.h:170-199 + 117-135) ----------------
            // Up to `reps` inner steps per scheduling iteration, skipping the vote/refill header in between (the classic
            // while-while inner loop; lanes that reach a leaf wait for the next iteration): 3 in ordinary waves (steady state
            // -4 %), 8 in thin waves, which are bound by the dependent instruction chain of an iteration (fixed cost -10 %).
            const uint32_t reps = nActive <= a.tailActive ? a.thinReps : a.innerReps;
            for (uint32_t rep = 0;; ++rep) {
                if (STATS) { ++stInner; stInnerLanes += uint32_t(__popcll(__ballot(int(node) < 0))); }
            if (int(node) < 0) {
                const uint32_t off = node << 6;             // bit 31 falls off: byte offset of the 64 B record
                u32x2 kids;
                float4 d1, d2, d3;
                if (BUF) {
                    kids = __builtin_amdgcn_raw_buffer_load_b64(nodeRsrc, off, 0, 0);
                    d1 = asFloat4(__builtin_amdgcn_raw_buffer_load_b128(nodeRsrc, off + 16u, 0, 0));
                    d2 = asFloat4(__builtin_amdgcn_raw_buffer_load_b128(nodeRsrc, off + 32u, 0, 0));
                    d3 = asFloat4(__builtin_amdgcn_raw_buffer_load_b128(nodeRsrc, off + 48u, 0, 0));
                } else if (SOA) {
                    // Ablation of the "SoA in HBM" layout: plane p of node i lives at nodesSoa[p * nodeCount + i].  Every lane
                    // is at a different node, so the four 16 B reads of one visit land in four cache lines instead of one.
                    const float4* np = a.nodesSoa + size_t(node & 0x7FFFFFFFu);
                    const uint2 k2 = *reinterpret_cast<const uint2*>(np);
                    d1 = np[size_t(a.nodeCount)]; d2 = np[size_t(a.nodeCount) * 2]; d3 = np[size_t(a.nodeCount) * 3];
                    asm volatile("" :: "v"(k2.x), "v"(k2.y));
                    kids.x = k2.x; kids.y = k2.y;
                } else {
                    const float4* np = a.nodes + size_t(node & 0x7FFFFFFFu) * 4;
                    const uint2 k2 = *reinterpret_cast<const uint2*>(np);
                    d1 = np[1]; d2 = np[2]; d3 = np[3];
                    asm volatile("" :: "v"(k2.x), "v"(k2.y));   // keep the child-ref load up here, in flight with the boxes (the
                    kids.x = k2.x; kids.y = k2.y;               // compiler otherwise sinks it behind the slab tests: +1 round trip)
                    // The whole record must have arrived before the touches go out, on every path: otherwise the compiler parks
                    // its vmcnt(0) for d1..d3 behind the branch, where it would wait for the touches as well.
                    if (PF) asm volatile("" :: "v"(d1.x), "v"(d2.x), "v"(d3.w));
                    if (PF && nActive <= a.tailActive) {
                        // Thin wave = latency-bound: pull both children's lines towards this CU while the slab tests run.
                        // LDS-DMA loads have no register destination, so nothing has to stay reserved while they are in flight
                        // and the compiler does not wait for them before the next record's own wait (where they are older).
                        const char* pa = int(kids.x) < 0 ? reinterpret_cast<const char*>(a.nodes) + (size_t(kids.x & 0x7FFFFFFFu) << 6)
                                                         : reinterpret_cast<const char*>(a.pairs) + size_t(kids.x & 0xFFFFFFu) * 48u;
                        const char* pb = int(kids.y) < 0 ? reinterpret_cast<const char*>(a.nodes) + (size_t(kids.y & 0x7FFFFFFFu) << 6)
                                                         : reinterpret_cast<const char*>(a.pairs) + size_t(kids.y & 0xFFFFFFu) * 48u;
                        typedef const __attribute__((address_space(1))) void* gptr_t;
                        typedef __attribute__((address_space(3))) void* lptr_t;
                        __builtin_amdgcn_global_load_lds((gptr_t)pa, (lptr_t)(pfSink + (tid & ~63u)), 4, 0, 0);
                        __builtin_amdgcn_global_load_lds((gptr_t)pb, (lptr_t)(pfSink + BLOCK + (tid & ~63u)), 4, 0, 0);
                    }
                }
                const float tRay = r.tFar;
                float tFirst, tLast;
                slabPair(d1, d2, d3, r.ix, r.iy, r.iz, r.ex, r.ey, r.ez, r.tNear, tRay, tFirst, tLast);
                const float firstDiff = tRay - tFirst, lastDiff = tRay - tLast;
                if (firstDiff + lastDiff != 0.0f) {
                    const bool lastNearer = tLast < tFirst;      // signbit(tLast - tFirst), Kernels.h:193
                    if (tFirst != tRay && tLast != tRay) RACC_PUSH(lastNearer ? kids.x : kids.y);
                    node = lastNearer ? kids.y : kids.x;
                } else {
                    RACC_POP_OR_DONE();
                }
            }
                if (rep + 1u >= reps || __ballot(int(node) < 0) == 0ull) break;
            }
            if (STATS) cyInner += __builtin_readcyclecounter() - cyTop;
        }
    }
#undef RACC_PUSH
#undef RACC_POP_OR_DONE

    if (STATS && lane == 0) {
        atomicAdd(a.stats + 0, (unsigned long long)stInner); atomicAdd(a.stats + 1, (unsigned long long)stInnerLanes);
        atomicAdd(a.stats + 2, (unsigned long long)stLeaf);  atomicAdd(a.stats + 3, (unsigned long long)stLeafLanes);
        atomicAdd(a.stats + 4, (unsigned long long)stRefill); atomicAdd(a.stats + 5, (unsigned long long)stLoaded);
        atomicAdd(a.stats + 6, (unsigned long long)stDeq);   atomicAdd(a.stats + 7, 1ull);
        atomicAdd(a.stats + 8, cyInner); atomicAdd(a.stats + 10, cyLeaf); atomicAdd(a.stats + 12, cyRefill);
        atomicAdd(a.stats + 13, (unsigned long long)(__builtin_readcyclecounter() - cyStart));
    }
    __syncthreads();
    if (tid == 0) {
        const uint32_t prev = atomicInc(a.cursor + 1, gridDim.x - 1);
        if (prev == gridDim.x - 1) {
            atomicExch(a.cursor, 0u);
            if (XQ) for (uint32_t q = 0; q < 8u; ++q) atomicExch(a.cursor + kXcdCursorWord + q, 0u);
        }
    }
}

#include "racc_kernels_experimental.inc"

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
    bool pendingEnv = false;             // last traversal launch parked miss directions (V2): envShade must follow
};

}  // namespace

struct racc_hip_ctx {
    int device = 0;
    int numCUs = 0;
    racc_hip_options opts{};
    Lane lanes[RACC_HIP_MAX_LANES];
    uint32_t* hostTrips = nullptr;       // host-mapped: kernels bump it when a wave hits the iteration limit
    uint32_t* devTrips = nullptr;        // the device alias of the same word
    std::atomic<uint32_t> seenTrips{0};
    uint32_t maxIters = 1u << 24;        // RACC_MAX_ITERS overrides (tests)
};

struct racc_hip_scene {
    float4* nodes = nullptr;
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

struct Variant {
    int block, ldsLevels, cacheNodes;
    void (*kernel)(const TraverseArgs);
    bool noSpill = false;      // kernel has no global spill path: only valid while tree height <= ldsLevels
    bool deferEnv = false;     // kernel parks miss directions; envShadeKernel must follow
    int slots = 1;             // ray slots per lane (V4: 2); ldsLevels counts all of them
    int stackLevels() const { return ldsLevels > 32 ? ldsLevels - (kRecFields + 1) : ldsLevels / slots; }   // V3 rows fold the record words into ldsLevels
};
// kernel_variant n selects kVariants[n-1]; 0 selects kDefaultVariant.  LDS per workgroup = cacheNodes*64 + ldsLevels*block*4.
const Variant kVariants[] = {
    {256, 16, 0, traverseKernel<256, 16, 0>},          // 1: no node cache, 16 KiB/WG, up to 8 WG/CU
    {1024, 12, 1792, traverseKernel<1024, 12, 1792>},  // 2: 160 KiB: 112 KiB cache + 48 KiB stacks, 1 WG/CU (4 waves/SIMD)
    {1024, 16, 1536, traverseKernel<1024, 16, 1536>},  // 3: 160 KiB: 96 + 64
    {512, 16, 2048, traverseKernel<512, 16, 2048>},    // 4: 160 KiB: 128 + 32, 1 WG/CU (2 waves/SIMD)
    {512, 8, 960, traverseKernel<512, 8, 960>},        // 5: 76 KiB: 60 + 16, 2 WG/CU
    {256, 8, 448, traverseKernel<256, 8, 448>},        // 6: 36 KiB: 28 + 8, 4 WG/CU
    {1024, 12, 1024, traverseKernel<1024, 12, 1024>},  // 7: 112 KiB: 64 + 48
    {512, 12, 1024, traverseKernel<512, 12, 1024>},    // 8: 88 KiB: 64 + 24, 1 WG/CU
    {256, 16, 0, traverseKernel<256, 16, 0, true>},    // 9: variant 1 + scheduling statistics (debug)
    {256, 32, 0, traverseKernelV2<256, 32, false, false>, true, true},   // 10: V2, 32 LDS levels, no spill path (height <= 32)
    {256, 16, 0, traverseKernelV2<256, 16, true, false>, false, true},    // 11: V2, 16 LDS levels + global spill (any height)
    {256, 32, 0, traverseKernelV2<256, 32, false, true>, true, true},    // 12: variant 10 + statistics (debug)
    {256, 24, 0, traverseKernelV2<256, 24, true, false>, false, true},    // 13: V2, 24 LDS levels + spill
    {256, 32, 0, traverseKernelV2<256, 32, false, false, false, false>, true, true},   // 14: V2 ablation: global loads, no top-of-stack register
    {256, 32, 0, traverseKernelV2<256, 32, false, false, true, false>, true, true},    // 15: V2 ablation: buffer loads only
    {256, 32, 0, traverseKernelV2<256, 32, false, false, false, true>, true, true},    // 16: V2 ablation: top-of-stack register only
    {256, 16, 0, traverseKernelV2<256, 16, true, false, false, false>, false, true},          // 17: V2 ablation: 16 levels + spill, neither
    {512, 16 + kRecFields + 1, 0, traverseKernelV3<8, 16, true, false>, false, true},      // 18: V3, 8-wave workgroups (2 per CU)
    {1024, 16 + kRecFields + 1, 0, traverseKernelV3<16, 16, true, false>, false, true},    // 19: V3, 16-wave workgroups (1 per CU)
    {256, 16 + kRecFields + 1, 0, traverseKernelV3<4, 16, true, false>, false, true},      // 20: V3, 4-wave workgroups (4 per CU)
    {512, 16 + kRecFields + 1, 0, traverseKernelV3<8, 16, true, true>, false, true},       // 21: variant 18 + statistics (debug)
    {256, 26, 0, traverseKernelV2<256, 26, false, false, false, false>, true, true},       // 22: V2, 26 LDS levels, no spill path (26 KiB: 6 WG/CU)
    {256, 28, 0, traverseKernelV2<256, 28, false, false, false, false>, true, true},       // 23: V2, 28 LDS levels, no spill path (28 KiB: 5 WG/CU)
    {256, 30, 0, traverseKernelV2<256, 30, false, false, false, false>, true, true},       // 24: V2, 30 LDS levels, no spill path
    {256, 26, 0, traverseKernelV2<256, 26, false, false, false, false, true>, true, true}, // 25: variant 22 + per-XCD ray queues (measured slower: DESIGN.md §3)
    {256, 26, 0, traverseKernelV2<256, 26, false, true, false, false, false>, true, true}, // 26: variant 22 + statistics (debug)
    {256, 26, 0, traverseKernelV2<256, 26, false, false, false, false, false, true>, true, true}, // 27: variant 22 with the node records transposed to SoA planes (ablation: DESIGN.md §2)
    {256, 26, 0, traverseKernelV2<256, 26, false, false, false, false, false, false, true>, true, true}, // 28: variant 22 + touch loads of both children in thin waves
    {256, 26, 0, traverseKernelV4<256, 13, false>, false, true, 2},   // 29: V4, two ray slots per lane, 13 LDS levels each + spill
    {256, 26, 0, traverseKernelV4<256, 13, true>, false, true, 2},    // 30: variant 29 + statistics (debug)
};
constexpr int kNumVariants = int(sizeof(kVariants) / sizeof(kVariants[0]));
constexpr int kSoaVariant = 27;
constexpr int kSpillFallback = 17;    // V2 with 16 LDS levels + global spill: used when a tree is taller than a variant's LDS stack
constexpr uint32_t kLdsPerCU = 160u * 1024u;

// kernel_variant 0 (default): the V2 kernel with the smallest LDS-only stack that covers the tree height (measured
// best: no spill branches on the hot path), falling back to the 16-level + global-spill instantiation for tall trees.
const Variant& pickVariant(const racc_hip_ctx* ctx, uint32_t treeHeight) {
    const uint32_t v = ctx->opts.kernel_variant;
    if (v >= 1 && v <= uint32_t(kNumVariants)) return kVariants[v - 1];
    if (treeHeight <= 26u) return kVariants[22 - 1];
    if (treeHeight <= 30u) return kVariants[24 - 1];
    if (treeHeight <= 32u) return kVariants[14 - 1];
    return kVariants[kSpillFallback - 1];
}

int launchTraverse(racc_hip_ctx* ctx, Lane& lane, hipStream_t stream, const racc_hip_scene* scene, const racc_hip_env* env,
                   const void* dRays, void* dResults, uint32_t count) {
    if (!count) return RACC_HIP_OK;
    const Variant* vp = &pickVariant(ctx, scene->info.inner_height);
    if (vp->noSpill && scene->info.inner_height > uint32_t(vp->stackLevels())) vp = &kVariants[kSpillFallback - 1];   // tall tree
    const Variant& v = *vp;
    const uint32_t ldsBytes = uint32_t(v.cacheNodes) * 64u + uint32_t(v.ldsLevels) * uint32_t(v.block) * 4u + (v.ldsLevels > 32 ? 272u : 0u);
    const uint32_t wavesPerSimd = optOr(ctx->opts.waves_per_simd, 6u);   // measured best on 1M-ray batches (5: +1.3 %, 4: +8 % per ray); LDS caps it below
    const uint32_t wavesPerBlock = uint32_t(v.block) / 64u;
    uint32_t blocksPerCU = (wavesPerSimd * 4u) / wavesPerBlock;
    if (blocksPerCU < 1u) blocksPerCU = 1u;
    if (blocksPerCU > kLdsPerCU / ldsBytes) blocksPerCU = kLdsPerCU / ldsBytes;
    uint32_t blocks = uint32_t(ctx->numCUs) * blocksPerCU;
    const uint32_t blocksNeeded = (count + uint32_t(v.block) - 1) / uint32_t(v.block);
    if (blocks > blocksNeeded) blocks = blocksNeeded;
    const uint32_t gridThreads = blocks * uint32_t(v.block);
    const uint32_t spillLevels = (scene->info.inner_height > uint32_t(v.stackLevels()) ? scene->info.inner_height - uint32_t(v.stackLevels()) : 0u) * uint32_t(v.slots);
    if (int rc = ensureSpill(ctx, lane, uint32_t(ctx->numCUs) * 2048u, spillLevels)) return rc;

    TraverseArgs a;
    a.rays = static_cast<const float4*>(dRays);
    a.results = static_cast<float4*>(dResults);
    a.count = count;
    a.nodes = scene->nodes; a.pairs = scene->pairs; a.remap = scene->remap;
    a.cacheCount = scene->info.node_count < uint32_t(v.cacheNodes) ? scene->info.node_count : uint32_t(v.cacheNodes);
    a.nodeBytes = scene->info.node_count * 64u;
    a.nodesSoa = scene->nodesSoa; a.nodeCount = scene->info.node_count;
    if (&v == &kVariants[kSoaVariant - 1] && !scene->nodesSoa) return fail(RACC_HIP_ERR_INVALID, "the SoA ablation variant needs a scene uploaded through a context created with that variant");
    a.pairBytes = scene->info.pair_count * 48u;
    a.env = env ? env->pixels : nullptr;
    a.envW = env ? env->width : 0; a.envH = env ? env->height : 0;
    a.cursor = lane.cursor;
    a.spill = lane.spill;
    a.spillStride = gridThreads;
    a.chunk = optOr(ctx->opts.chunk, 64u * uint32_t(v.slots));
    if (a.chunk > 65536u) a.chunk = 65536u;      // grid waves x chunk (the statically assigned first chunks) must stay far below 2^32
    a.refillMin = optOr(ctx->opts.refill_min, 32u);
    a.leafMin = optOr(ctx->opts.leaf_min, 12u);
    a.maxIters = ctx->maxIters;
    a.trips = ctx->devTrips;
    a.tailActive = ctx->opts.tail_active ? (ctx->opts.tail_active > 64u ? 0u : ctx->opts.tail_active) : 32u;   // >64 disables
    a.regroup = optOr(ctx->opts.regroup_period, 8u);
    a.thinReps = optOr(ctx->opts.thin_reps, 8u);
    a.innerReps = optOr(ctx->opts.inner_reps, 3u);
    a.stats = reinterpret_cast<unsigned long long*>(lane.cursor + 8);
    hipLaunchKernelGGL(v.kernel, dim3(blocks), dim3(v.block), 0, stream, a);
    HIP_TRY(hipGetLastError(), "launch traverseKernel");
    lane.pendingEnv = v.deferEnv && env != nullptr;
    lane.info.grid_blocks = blocks;
    lane.info.block_threads = uint32_t(v.block);
    lane.info.lds_bytes_per_block = ldsBytes;
    lane.info.waves_per_simd = blocksPerCU * wavesPerBlock / 4u;
    return RACC_HIP_OK;
}

int launchEnvShade(racc_hip_ctx* ctx, Lane& lane, hipStream_t stream, const racc_hip_env* env, void* dResults, uint32_t count) {
    if (!lane.pendingEnv || !env || !count) return RACC_HIP_OK;
    lane.pendingEnv = false;
    uint32_t blocks = (count + 255u) / 256u;
    const uint32_t cap = uint32_t(ctx->numCUs) * 8u;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(envShadeKernel, dim3(blocks), dim3(256), 0, stream, static_cast<float4*>(dResults), count, env->pixels, env->width, env->height);
    HIP_TRY(hipGetLastError(), "launch envShadeKernel");
    return RACC_HIP_OK;
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

const char* racc_hip_version(void) { return "racc-hip 0.1 (gfx950)"; }

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
    if (!ctx->opts.lanes) ctx->opts.lanes = 4;                       // RayAccelerator.cpp:436
    if (ctx->opts.lanes > RACC_HIP_MAX_LANES) ctx->opts.lanes = RACC_HIP_MAX_LANES;
    if (ctx->opts.waves_per_simd > 8) ctx->opts.waves_per_simd = 8;
    if (ctx->opts.refill_min > 64) ctx->opts.refill_min = 64;
    if (ctx->opts.leaf_min > 64) ctx->opts.leaf_min = 64;
    for (uint32_t i = 0; i < ctx->opts.lanes; ++i) {
        Lane& l = ctx->lanes[i];
        hipError_t e1 = hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking);
        hipError_t e2 = e1 == hipSuccess ? hipMalloc(reinterpret_cast<void**>(&l.cursor), 256) : e1;
        hipError_t e3 = e2 == hipSuccess ? hipMemset(l.cursor, 0, 256) : e2;
        if (e3 != hipSuccess) { racc_hip_destroy(ctx); return fail(RACC_HIP_ERR_DEVICE, "lane setup", e3); }
    }
    {
        hipError_t e1 = hipHostMalloc(reinterpret_cast<void**>(&ctx->hostTrips), 64, hipHostMallocMapped);
        if (e1 == hipSuccess) { *ctx->hostTrips = 0; e1 = hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->devTrips), ctx->hostTrips, 0); }
        if (e1 != hipSuccess) { racc_hip_destroy(ctx); return fail(RACC_HIP_ERR_DEVICE, "watchdog word", e1); }
        if (const char* m = std::getenv("RACC_MAX_ITERS")) { const long long v = std::atoll(m); if (v > 0 && v < (1ll << 31)) ctx->maxIters = uint32_t(v); }
    }
    *out = ctx;
    return RACC_HIP_OK;
}

int racc_hip_destroy(racc_hip_ctx* ctx) {
    if (!ctx) return RACC_HIP_OK;
    hipSetDevice(ctx->device);
    hipDeviceSynchronize();              // launches given a caller's own stream (racc_hip_intersect_device) included
    if (ctx->hostTrips) hipHostFree(ctx->hostTrips);
    for (Lane& l : ctx->lanes) {
        if (l.stream) hipStreamSynchronize(l.stream);
        for (hipEvent_t ev : l.events) hipEventDestroy(ev);
        for (hipEvent_t ev : l.pipeEvents) hipEventDestroy(ev);
        if (l.copyIn) hipStreamDestroy(l.copyIn);
        if (l.copyOut) hipStreamDestroy(l.copyOut);
        if (l.cursor) hipFree(l.cursor);
        if (l.spill) hipFree(l.spill);
        if (l.dRays) hipFree(l.dRays);
        if (l.dResults) hipFree(l.dResults);
        if (l.stream) hipStreamDestroy(l.stream);
    }
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
    if (e == hipSuccess) e = hipMemcpy(s->pairs, pairs48, pb, hipMemcpyHostToDevice);
    if (e == hipSuccess && rb) e = hipMemcpy(s->remap, remap, rb, hipMemcpyHostToDevice);
    if (e != hipSuccess) { racc_hip_scene_free(ctx, s); return fail(RACC_HIP_ERR_DEVICE, "scene upload", e); }
    info.node_count = node_count; info.pair_count = pair_count; info.remap_count = remap_count;
    info.device_bytes = nb + pb + rb;
    {
        const Variant& v = pickVariant(ctx, info.inner_height);
        info.spill_levels = info.inner_height > uint32_t(v.stackLevels()) ? info.inner_height - uint32_t(v.stackLevels()) : 0u;
    }
    s->info = info;
    *out = s;
    return RACC_HIP_OK;
}

int racc_hip_scene_free(racc_hip_ctx* ctx, racc_hip_scene* s) {
    if (!s) return RACC_HIP_OK;
    if (ctx) hipSetDevice(ctx->device);
    if (s->nodes) hipFree(s->nodes);
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
    if (ctx) hipSetDevice(ctx->device);
    if (env->pixels) hipFree(env->pixels);
    delete env;
    return RACC_HIP_OK;
}

int racc_hip_register_host(racc_hip_ctx* ctx, void* ptr, uint64_t bytes) {
    if (!ctx || !ptr || !bytes) return fail(RACC_HIP_ERR_INVALID, "register_host: bad argument");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault), "hipHostRegister");
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
    HIP_TRY(hipHostRegister(rays, size_t(capacity) * 32, hipHostRegisterDefault), "hipHostRegister(rays)");
    hipError_t e = hipHostRegister(results, size_t(capacity) * 16, hipHostRegisterDefault);
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
    HIP_TRY(hipStreamSynchronize(l.stream), "hipStreamSynchronize");     // staging buffers are reused per lane
    if (int rc = ensureStaging(l, count)) return rc;
    HIP_TRY(hipMemcpyAsync(l.dRays, rays, size_t(count) * 32, hipMemcpyHostToDevice, l.stream), "H2D rays");
    if (int rc = launchTraverse(ctx, l, l.stream, scene, env, l.dRays, l.dResults, count)) return rc;
    if (int rc = launchEnvShade(ctx, l, l.stream, env, l.dResults, count)) return rc;
    HIP_TRY(hipMemcpyAsync(results, l.dResults, size_t(count) * 16, hipMemcpyDeviceToHost, l.stream), "D2H results");
    return RACC_HIP_OK;
}

int racc_hip_wait(racc_hip_ctx* ctx, uint32_t lane) {
    if (int rc = checkLane(ctx, lane)) return rc;
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    HIP_TRY(hipStreamSynchronize(ctx->lanes[lane].stream), "hipStreamSynchronize");
    return checkWatchdog(ctx);
}

int racc_hip_intersect(racc_hip_ctx* ctx, const racc_hip_scene* scene, const racc_hip_env* env,
                       const void* rays, void* results, uint32_t count, uint32_t lane) {
    if (count >= 524288u) return racc_hip_intersect_streams(ctx, scene, env, 1, &rays, &results, &count, lane);   // sliced: copies beside kernels
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
    // Large batches: cut into slices so that the PCIe copy of slice k+1 (in) and of slice k-1 (out) run beside the kernel of
    // slice k (PCIe is full duplex; a 1M-ray batch is 32 MiB in, 16 MiB out, 0.9 ms of copies against 0.4 ms of kernel).
    if (!l.copyIn) HIP_TRY(hipStreamCreateWithFlags(&l.copyIn, hipStreamNonBlocking), "hipStreamCreate");
    if (!l.copyOut) HIP_TRY(hipStreamCreateWithFlags(&l.copyOut, hipStreamNonBlocking), "hipStreamCreate");
    while (l.pipeEvents.size() < size_t(slices) * 2) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate");
        l.pipeEvents.push_back(ev);
    }
    const uint64_t per = ((total + slices - 1) / slices + 63) / 64 * 64;
    for (uint32_t k = 0; k < slices; ++k) {
        const uint64_t g0 = uint64_t(k) * per, g1 = g0 + per < total ? g0 + per : total;
        if (g0 >= g1) break;
        HIP_TRY(copyRange(0, g0, g1, l.copyIn), "H2D rays");
        HIP_TRY(hipEventRecord(l.pipeEvents[2 * k], l.copyIn), "hipEventRecord");
        HIP_TRY(hipStreamWaitEvent(l.stream, l.pipeEvents[2 * k], 0), "hipStreamWaitEvent");
        if (int rc = launchTraverse(ctx, l, l.stream, scene, env, static_cast<char*>(l.dRays) + g0 * 32, static_cast<char*>(l.dResults) + g0 * 16, uint32_t(g1 - g0))) return rc;
        if (int rc = launchEnvShade(ctx, l, l.stream, env, static_cast<char*>(l.dResults) + g0 * 16, uint32_t(g1 - g0))) return rc;
        HIP_TRY(hipEventRecord(l.pipeEvents[2 * k + 1], l.stream), "hipEventRecord");
        HIP_TRY(hipStreamWaitEvent(l.copyOut, l.pipeEvents[2 * k + 1], 0), "hipStreamWaitEvent");
        HIP_TRY(copyRange(1, g0, g1, l.copyOut), "D2H results");
    }
    HIP_TRY(hipStreamSynchronize(l.copyOut), "hipStreamSynchronize");
    HIP_TRY(hipStreamSynchronize(l.stream), "hipStreamSynchronize");
    return checkWatchdog(ctx);
}

int racc_hip_intersect_device(racc_hip_ctx* ctx, const racc_hip_scene* scene, const racc_hip_env* env,
                              const void* d_rays, void* d_results, uint32_t count, uint32_t lane, void* stream) {
    if (int rc = checkLane(ctx, lane)) return rc;
    if (!scene) return fail(RACC_HIP_ERR_INVALID, "scene is NULL");
    if (!count) return RACC_HIP_OK;
    if (!d_rays || !d_results) return fail(RACC_HIP_ERR_INVALID, "d_rays/d_results is NULL");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    Lane& l = ctx->lanes[lane];
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : l.stream;
    if (int rc = launchTraverse(ctx, l, st, scene, env, d_rays, d_results, count)) return rc;
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

int racc_hip_synchronize(racc_hip_ctx* ctx) {
    if (!ctx) return fail(RACC_HIP_ERR_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device), "hipSetDevice");
    HIP_TRY(hipDeviceSynchronize(), "hipDeviceSynchronize");
    return checkWatchdog(ctx);
}

}  // extern "C"
