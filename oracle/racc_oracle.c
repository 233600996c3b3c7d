/*
 * racc_oracle.c — CPU restatement of the reference hot path (see racc_oracle.h).
 *
 * TEST INFRASTRUCTURE ONLY — never linked into or loaded by the product.
 * Pinned against the reference's own OpenCL kernel run on the MI355X (oracle/_ref,
 * tests/test_gpu_reference_kernel.py); the Embree CPU path is unavailable.  See racc_oracle.h.
 *
 * Arithmetic contract.  The reference builds its OpenCL kernel with
 * -cl-fast-relaxed-math -cl-mad-enable (RayAccelerator.cpp:489-490), so the
 * rounding of mad()/dot() and the precision of native_recip/native_rsqrt are
 * implementation-defined there.  This restatement fixes ONE IEEE-754 binary32
 * evaluation (round-to-nearest, subnormals kept, no contraction other than
 * the explicit fmaf calls) so that the HIP kernel can be bit-identical to it:
 *   mad(a,b,c)          = fmaf(a,b,c)
 *   dot(a,b)            = fmaf(a.z,b.z, fmaf(a.y,b.y, a.x*b.x))
 *   mad_cross(a,b).x    = fmaf(a.y,b.z, -(a.z*b.y))      (Kernels.h:23-25)
 *   native_recip(x)     = 1.0f/x  (correctly rounded)     (Kernels.h:107)
 *   native_rsqrt(x)     = 1.0f/sqrtf(x)                   (Kernels.h:216)
 *   fmin/fmax           = (a<b?a:b)/(a>b?a:b); equal to IEEE minNum/maxNum for
 *                         the finite, zero-sign-insensitive uses below
 *   signbit(tLast-tFirst) is evaluated as (tLast < tFirst) so that the sign of
 *   a zero produced by min/max can never steer the traversal (Kernels.h:193).
 * Rays with a non-finite origin/direction/minT component or a NaN maxT are
 * undefined behaviour in the reference (the renderer drops them,
 * PathTracingRenderer.cpp:405-408); here they are defined to miss with rgb=0.
 * maxT = +inf is valid and follows the kernel's arithmetic literally.
 *
 * Build: gcc -O2 -ffp-contract=off -mfma (see oracle/Makefile).
 */
#include "racc_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ helpers */

typedef union { float f; uint32_t u; int32_t i; } fbits;

static inline uint32_t f2u(float f) { fbits b; b.f = f; return b.u; }
static inline float u2f(uint32_t u) { fbits b; b.u = u; return b.f; }
static inline float omin(float a, float b) { return a < b ? a : b; }
static inline float omax(float a, float b) { return a > b ? a : b; }

static inline float dot3(const float a[3], const float b[3]) {
    return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
}
/* Kernels.h:23-25 */
static inline void mad_cross(float r[3], const float a[3], const float b[3]) {
    r[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
    r[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
    r[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}

typedef struct {
    float o[3], d[3];
    float tNear, tFar;
} ray_state;

typedef struct { int32_t index; float t, u, v; } hit_state;

/* ------------------------------------------------- Kernels.h:36-115 (pair) */
static inline float pair_intersect(const orc_pair* pairs, int32_t index, const ray_state* ray, hit_state* hit) {
    const orc_pair* p = pairs + index;
    const float tNear = ray->tNear, tMax = ray->tFar;
    const float e1[3] = { p->e1[0], p->e1[1], p->e1[2] };
    const float e2[3] = { p->e2[0], p->e2[1], p->e2[2] };
    const float e3[3] = { p->e3x, p->e3y, p->e3z };
    const float v0[3] = { p->p0[0], p->p0[1], p->p0[2] };

    float n1[3], n2[3], C[3], R[3];
    mad_cross(n1, e1, e2);
    mad_cross(n2, e3, e1);
    C[0] = v0[0] - ray->o[0]; C[1] = v0[1] - ray->o[1]; C[2] = v0[2] - ray->o[2];
    mad_cross(R, ray->d, C);

    const float det1 = dot3(n1, ray->d);
    const float det2 = dot3(n2, ray->d);
    const uint32_t sgnDet1 = f2u(det1) & 0x80000000u;
    const uint32_t sgnDet2 = f2u(det2) & 0x80000000u;

    const uint32_t iU1 = f2u(dot3(R, e2)) ^ sgnDet1;
    const uint32_t iV1 = f2u(dot3(R, e1)) ^ sgnDet1;
    const uint32_t iU2 = f2u(-dot3(R, e1)) ^ sgnDet2;
    const uint32_t iV2 = f2u(-dot3(R, e3)) ^ sgnDet2;

    if ((int32_t)((iU1 | iV1) & (iU2 | iV2)) < 0)
        return tMax;

    int outside1 = (int32_t)(iU1 | iV1) < 0;
    int outside2 = (int32_t)(iU2 | iV2) < 0;

    float U1 = u2f(iU1), V1 = u2f(iV1);
    const float U2 = u2f(iU2), V2 = u2f(iV2);
    float absDet1 = fabsf(det1);
    const float absDet2 = fabsf(det2);
    const float W1 = absDet1 - U1 - V1;
    const float W2 = absDet2 - U2 - V2;
    float T1 = u2f(f2u(dot3(n1, C)) ^ sgnDet1);
    const float T2 = u2f(f2u(dot3(n2, C)) ^ sgnDet2);

    /* open at tNear, closed at tMax (Kernels.h:88-89) */
    outside1 = outside1 || (W1 < 0.0f || T1 <= absDet1 * tNear || T1 > absDet1 * tMax);
    outside2 = outside2 || (W2 < 0.0f || T2 <= absDet2 * tNear || T2 > absDet2 * tMax);
    if (outside1 && outside2)
        return tMax;

    index = index * 2;
    if ((!outside2 && outside1) || (!outside1 && !outside2 && T1 * absDet2 > T2 * absDet1)) {
        absDet1 = absDet2; T1 = T2; U1 = U2; V1 = V2;
        ++index;
    }
    const float rcp = 1.0f / absDet1;
    const float t = T1 * rcp;
    hit->index = index;
    hit->t = t;
    hit->u = U1 * rcp;
    hit->v = V1 * rcp;
    return t;
}

/* ------------------------------------------------ Kernels.h:117-135 (slab) */
static inline float aabb_intersect(const float mn[3], const float mx[3], const ray_state* ray,
                                   const float invDir[3], const float OoD[3]) {
    float t0 = ray->tNear, t1 = ray->tFar;
    float tMin[3], tMax[3];
    for (int k = 0; k < 3; ++k) {
        const float a = fmaf(mn[k], invDir[k], OoD[k]);
        const float b = fmaf(mx[k], invDir[k], OoD[k]);
        tMin[k] = omin(a, b);
        tMax[k] = omax(a, b);
    }
    t0 = omax(omax(t0, tMin[0]), omax(tMin[1], tMin[2]));
    t1 = omin(omin(t1, tMax[0]), omin(tMax[1], tMax[2]));
    if (t0 > t1)
        return ray->tFar;
    return t0;
}

/* --------------------------------------------- Kernels.h:213-222 (miss rgb) */
void orc_env_sample(const float* env, uint32_t envW, uint32_t envH, const float d[3], float rgb[3]) {
    rgb[0] = rgb[1] = rgb[2] = 0.0f;
    if (!env || !envW || !envH)
        return;
    const float rlen = 1.0f / sqrtf(fmaf(d[2], d[2], d[1] * d[1]));
    float r = (rlen > 1e+6f) ? 0.0f : acosf(-d[0]) * (1.0f / (2.0f * 3.141593f)) * rlen;
    if (!isfinite(r))
        r = 0.0f; /* Environment.h:42-43 guard; the GPU kernel leaves this case undefined */
    const float u = 0.5f - r * d[2];
    const float v = 0.5f - r * d[1];
    /* OpenCL 1.2 §8.2: normalized coords, CLAMP_TO_EDGE, FILTER_LINEAR */
    const float fx = u * (float)envW - 0.5f, fy = v * (float)envH - 0.5f;
    const float flx = floorf(fx), fly = floorf(fy);
    const float a = fx - flx, b = fy - fly;
    int x0 = (int)flx, y0 = (int)fly, x1 = x0 + 1, y1 = y0 + 1;
    const int w1 = (int)envW - 1, h1 = (int)envH - 1;
    x0 = x0 < 0 ? 0 : (x0 > w1 ? w1 : x0); x1 = x1 < 0 ? 0 : (x1 > w1 ? w1 : x1);
    y0 = y0 < 0 ? 0 : (y0 > h1 ? h1 : y0); y1 = y1 < 0 ? 0 : (y1 > h1 ? h1 : y1);
    const float* t00 = env + ((size_t)y0 * envW + x0) * 4;
    const float* t10 = env + ((size_t)y0 * envW + x1) * 4;
    const float* t01 = env + ((size_t)y1 * envW + x0) * 4;
    const float* t11 = env + ((size_t)y1 * envW + x1) * 4;
    for (int c = 0; c < 3; ++c)
        rgb[c] = (1.0f - a) * (1.0f - b) * t00[c] + a * (1.0f - b) * t10[c]
               + (1.0f - a) * b * t01[c] + a * b * t11[c];
}

/* Optional per-node visit histogram (analysis aid for device-layout decisions). */
static uint32_t* g_visit_hist = 0;
void orc_set_visit_histogram(uint32_t* hist) { g_visit_hist = hist; }

/* ------------------------------------------ Kernels.h:139-242 (`traversal`) */
static void traverse_one(const orc_gpu_node* nodes, const orc_pair* pairs, const uint32_t* remap,
                         const float* env, uint32_t envW, uint32_t envH,
                         const orc_ray* in, orc_result* out,
                         uint32_t* nvOut, uint32_t* npOut, uint32_t* depthOut) {
    ray_state ray;
    uint32_t nv = 0, np = 0, maxDepth = 0;
    for (int k = 0; k < 3; ++k) { ray.o[k] = in->origin[k]; ray.d[k] = in->dir[k]; }
    ray.tNear = in->minT; ray.tFar = in->maxT;

    int finite = isfinite(ray.tNear) && !isnan(ray.tFar); /* maxT = +inf is a legitimate "no limit" */
    for (int k = 0; k < 3; ++k) finite = finite && isfinite(ray.o[k]) && isfinite(ray.d[k]);
    if (!finite) {
        out->triangle = 0xFFFFFFFFu; out->t = out->u = out->v = 0.0f;
        if (nvOut) *nvOut = 0;
        if (npOut) *npOut = 0;
        if (depthOut) *depthOut = 0;
        return;
    }

    const float epsilon = 1e-10f; /* Kernels.h:149-157 */
    for (int k = 0; k < 3; ++k)
        if (fabsf(ray.d[k]) < epsilon) ray.d[k] = copysignf(epsilon, ray.d[k]);

    float invDir[3], OoD[3];
    for (int k = 0; k < 3; ++k) { invDir[k] = 1.0f / ray.d[k]; OoD[k] = -ray.o[k] * invDir[k]; }

    hit_state hit = { -1, ray.tFar, 0.0f, 0.0f };
    uint32_t node = 0x80000000u;
    uint32_t stack[256]; /* reference: int stack[64] with no overflow check (Kernels.h:166) */
    uint32_t head = 0;

    for (;;) {
        if (node & 0x80000000u) {
            const orc_gpu_node* n = nodes + (node & 0x7FFFFFFFu);
            ++nv;
            if (g_visit_hist) __atomic_fetch_add(&g_visit_hist[node & 0x7FFFFFFFu], 1u, __ATOMIC_RELAXED);
            const float tRay = ray.tFar;
            const float tFirst = aabb_intersect(n->leftMin, n->leftMax, &ray, invDir, OoD);
            const float tLast = aabb_intersect(n->rightMin, n->rightMax, &ray, invDir, OoD);
            const float firstDiff = tRay - tFirst, lastDiff = tRay - tLast;
            if (firstDiff + lastDiff != 0.0f) {
                const int sgn = tLast < tFirst; /* signbit(tLast - tFirst), see header note */
                if (omax(tFirst, tLast) != tRay) {
                    if (head < 256) stack[head++] = sgn ? n->first : n->last;
                    if (head > maxDepth) maxDepth = head;
                }
                node = sgn ? n->last : n->first;
                continue;
            }
        } else {
            const int32_t first = (int32_t)(node & 0xFFFFFFu);
            const int32_t last = first + (int32_t)(node >> 24);
            for (int32_t i = first; i < last; ++i) {
                ray.tFar = pair_intersect(pairs, i, &ray, &hit);
                ++np;
            }
        }
        if (!head) break;
        node = stack[--head];
    }

    if (hit.index == -1) { /* Kernels.h:213-222 */
        float rgb[3];
        orc_env_sample(env, envW, envH, ray.d, rgb);
        out->triangle = 0xFFFFFFFFu; out->t = rgb[0]; out->u = rgb[1]; out->v = rgb[2];
    } else { /* Kernels.h:223-239 */
        uint32_t index = remap[hit.index];
        const uint32_t edge = index >> 30;
        index &= 0x3FFFFFFFu;
        const float bx = hit.u, by = hit.v, bz = 1.0f - hit.u - hit.v;
        float u = bx, v = by;
        if (edge == 1) { u = bz; v = bx; }       /* barys.zxy */
        else if (edge == 2) { u = by; v = bz; }  /* barys.yzx */
        out->triangle = index; out->t = hit.t; out->u = u; out->v = v;
    }
    if (nvOut) *nvOut = nv;
    if (npOut) *npOut = np;
    if (depthOut) *depthOut = maxDepth;
}

void orc_traverse(const orc_gpu_node* nodes, const orc_pair* pairs, const uint32_t* remap,
                  const float* env, uint32_t envW, uint32_t envH,
                  const orc_ray* rays, orc_result* results, uint32_t start, uint32_t end,
                  uint32_t* nv, uint32_t* np, uint32_t* depth) {
    for (uint32_t i = start; i < end; ++i)
        traverse_one(nodes, pairs, remap, env, envW, envH, rays + i, results + i,
                     nv ? nv + i : 0, np ? np + i : 0, depth ? depth + i : 0);
}

typedef struct {
    const orc_gpu_node* nodes; const orc_pair* pairs; const uint32_t* remap;
    const float* env; uint32_t envW, envH;
    const orc_ray* rays; orc_result* results; uint32_t count, slice;
    uint64_t* cursor;      /* counts slices over `repeat` passes of the batch */
    uint32_t repeat;
} mt_job;

static void* mt_worker(void* arg) {
    mt_job* j = (mt_job*)arg;
    const uint64_t perPass = ((uint64_t)j->count + j->slice - 1) / j->slice;
    for (;;) {
        const uint64_t k = __atomic_fetch_add(j->cursor, 1, __ATOMIC_RELAXED);
        if (k >= perPass * j->repeat) break;
        const uint32_t s = (uint32_t)((k % perPass) * j->slice);
        const uint32_t e = s + j->slice < j->count ? s + j->slice : j->count;
        orc_traverse(j->nodes, j->pairs, j->remap, j->env, j->envW, j->envH, j->rays, j->results, s, e, 0, 0, 0);
    }
    return 0;
}

void orc_traverse_mt(const orc_gpu_node* nodes, const orc_pair* pairs, const uint32_t* remap,
                     const float* env, uint32_t envW, uint32_t envH,
                     const orc_ray* rays, orc_result* results, uint32_t count,
                     uint32_t slice, uint32_t threads, uint32_t repeat) {
    if (!slice) slice = 1024;
    if (!threads) threads = 1;
    if (!repeat) repeat = 1;
    uint64_t cursor = 0;
    mt_job job = { nodes, pairs, remap, env, envW, envH, rays, results, count, slice, &cursor, repeat };
    pthread_t* tid = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    for (uint32_t t = 1; t < threads; ++t) pthread_create(&tid[t], 0, mt_worker, &job);
    mt_worker(&job);
    for (uint32_t t = 1; t < threads; ++t) pthread_join(tid[t], 0);
    free(tid);
}

/* ============================================================ BVH2 builder */
/* Restates Bvh2.cpp:257-535 (build) and :537-753,772-907 (bounds, keys, sort,
 * root).  Deviations, all schedule/ISA artefacts of the reference:
 *  - node numbering is the single-thread order (children allocated when their
 *    parent is split, left subtree before right; Bvh2.cpp:489-534 with no task
 *    spawn) — the reference's numbering depends on thread timing;
 *  - every surface area uses the vector path's expression
 *    fma(dx,dy, fma(dx,dz, dy*dz)) (Bvh2.cpp:339,405); the reference mixes it
 *    with (dx*dy + dy*dz) + dx*dz (Bvh2.cpp:76-80) depending on 8-alignment;
 *  - 1/psa is exact instead of _mm_rcp_ss (Bvh2.cpp:465), whose value differs
 *    between CPU vendors;
 *  - the early-outs of the two sweeps (Bvh2.cpp:346-351,418,431-432) are not
 *    taken: they cannot change the chosen split, only skip work.
 * Triangle bounds are (−min, max) 8-float records as in Bvh2.cpp:587-621. */

typedef struct {
    uint32_t* sorted[3];
    uint32_t* temp;
    float* accSah;
    uint8_t* left;
    float* tb;          /* [T][8]: -minx,-miny,-minz,-minw, maxx,maxy,maxz,maxw */
    orc_bvh2_node* nodes;
    uint32_t counter;
} build_state;

static inline float surface_area8(const float b[8]) {
    const float dx = b[4] + b[0], dy = b[5] + b[1], dz = b[6] + b[2];
    return fmaf(dx, dy, fmaf(dx, dz, dy * dz));
}
static inline void max8(float d[8], const float s[8]) {
    for (int k = 0; k < 8; ++k) d[k] = omax(d[k], s[k]);
}
static void range_bounds(const float* tb, const uint32_t* idx, uint32_t first, uint32_t last, float out[8]) {
    memcpy(out, tb + (size_t)idx[first] * 8, 32);
    for (uint32_t i = first + 1; i < last; ++i) max8(out, tb + (size_t)idx[i] * 8);
}

/* Bvh2.cpp:217-253 */
static void partition_ref(build_state* s, unsigned axis, uint32_t first, uint32_t last) {
    uint32_t* idx = s->sorted[axis];
    uint32_t l = first, r = 0;
    for (uint32_t i = first; i < last; ++i) {
        const uint32_t v = idx[i];
        if (s->left[v]) idx[l++] = v; else s->temp[r++] = v;
    }
    memcpy(idx + l, s->temp, sizeof(uint32_t) * r);
}

static void build_node(build_state* s, uint32_t nodeIndex) {
    orc_bvh2_node* node = &s->nodes[nodeIndex];
    const uint32_t first = node->first, last = node->last;
    float bounds[8];

    if (nodeIndex != 0) { /* Bvh2.cpp:265-270 */
        range_bounds(s->tb, s->sorted[0], first, last, bounds);
        node->bbMin[0] = -bounds[0]; node->bbMin[1] = -bounds[1]; node->bbMin[2] = -bounds[2];
        node->dummy0 = f2u(-bounds[3]);
        node->bbMax[0] = bounds[4]; node->bbMax[1] = bounds[5]; node->bbMax[2] = bounds[6];
        node->dummy1 = f2u(bounds[7]);
    } else {
        bounds[0] = -node->bbMin[0]; bounds[1] = -node->bbMin[1]; bounds[2] = -node->bbMin[2]; bounds[3] = 0;
        bounds[4] = node->bbMax[0]; bounds[5] = node->bbMax[1]; bounds[6] = node->bbMax[2]; bounds[7] = 0;
    }
    if (last - first <= 2) /* Bvh2.cpp:272 */
        return;

    const float psa = surface_area8(bounds);
    uint32_t bestDim = 0xFFFFFFFFu, pivot = 0xFFFFFFFFu;

    if (psa > 0.0f) {
        float bestSah = INFINITY;
        for (unsigned dim = 0; dim < 3; ++dim) {
            const uint32_t* idx = s->sorted[dim];
            float b[8];
            /* left sweep, Bvh2.cpp:292-357: accSah[i] = SA([first..i]) * (i-first+1) */
            memcpy(b, s->tb + (size_t)idx[first] * 8, 32);
            for (uint32_t i = first; i < last - 1; ++i) {
                max8(b, s->tb + (size_t)idx[i] * 8);
                s->accSah[i] = surface_area8(b) * (float)(int)(i - first + 1);
            }
            /* right sweep, Bvh2.cpp:359-453: pivot i splits [first,i) | [i,last) */
            uint32_t bestPivot = 0xFFFFFFFFu;
            memcpy(b, s->tb + (size_t)idx[last - 1] * 8, 32);
            for (uint32_t i = last - 1; i > first; --i) {
                max8(b, s->tb + (size_t)idx[i] * 8);
                const float sah = s->accSah[i - 1] + surface_area8(b) * (float)(int)(last - i);
                if (sah < bestSah) { bestSah = sah; bestPivot = i; }
            }
            if (bestPivot != 0xFFFFFFFFu) { pivot = bestPivot; bestDim = dim; }
        }
        /* Bvh2.cpp:462-475 */
        const float traversalCost = 2.0f, intersectionCost = 1.0f;
        const float cost = traversalCost + intersectionCost * (1.0f / psa) * bestSah;
        if (cost > (float)(int)(last - first) * intersectionCost) {
            if (last - first >= 127) { bestDim = 0; pivot = (first + last) >> 1; }
            else return;
        }
    } else { /* Bvh2.cpp:477-485 */
        if (last - first >= 127) { bestDim = 0; pivot = (first + last) >> 1; }
        else return;
    }

    /* Bvh2.cpp:242-253 */
    {
        const uint32_t* ref = s->sorted[bestDim];
        for (uint32_t i = first; i < pivot; ++i) s->left[ref[i]] = 1;
        for (uint32_t i = pivot; i < last; ++i) s->left[ref[i]] = 0;
        partition_ref(s, (bestDim + 1) % 3, first, last);
        partition_ref(s, (bestDim + 2) % 3, first, last);
    }

    /* Bvh2.cpp:489-509 */
    const uint32_t counter = (++s->counter) * 2 + 1;
    const uint32_t left = counter - 2, right = counter - 1;
    node->kind = bestDim + 1; node->first = left; node->last = right;
    orc_bvh2_node* ln = &s->nodes[left];
    orc_bvh2_node* rn = &s->nodes[right];
    memset(ln, 0, sizeof(*ln)); memset(rn, 0, sizeof(*rn));
    ln->kind = 0; ln->parent = nodeIndex; ln->first = first; ln->last = pivot;
    rn->kind = 0; rn->parent = nodeIndex; rn->first = pivot; rn->last = last;
    build_node(s, left);
    build_node(s, right);
}

/* Bvh2.cpp:128-184: stable LSD radix on the high 32 bits of (key<<32 | index) */
static void radix_sort_high32(uint64_t* data, uint64_t* tmp, uint32_t count) {
    for (int pass = 0; pass < 4; ++pass) {
        uint32_t hist[257] = { 0 };
        const int shift = 32 + pass * 8;
        for (uint32_t i = 0; i < count; ++i) ++hist[((data[i] >> shift) & 0xFF) + 1];
        for (int i = 1; i < 257; ++i) hist[i] += hist[i - 1];
        for (uint32_t i = 0; i < count; ++i) tmp[hist[(data[i] >> shift) & 0xFF]++] = data[i];
        uint64_t* sw = data; data = tmp; tmp = sw;
    }
}

int orc_bvh2_build(const float* vertices, uint32_t vertexCount,
                   const uint32_t* indices, uint32_t T,
                   orc_bvh2_node* nodes, uint32_t* triangles, uint32_t* nodeCount) {
    (void)vertexCount;
    if (!T) return -1;
    build_state s;
    memset(&s, 0, sizeof(s));
    s.nodes = nodes;
    s.sorted[0] = triangles;
    s.sorted[1] = (uint32_t*)malloc(sizeof(uint32_t) * T);
    s.sorted[2] = (uint32_t*)malloc(sizeof(uint32_t) * T);
    s.temp = (uint32_t*)malloc(sizeof(uint32_t) * T);
    s.accSah = (float*)malloc(sizeof(float) * T);
    s.left = (uint8_t*)malloc(T);
    s.tb = (float*)malloc(sizeof(float) * 8 * (size_t)T);
    uint64_t* keys = (uint64_t*)malloc(sizeof(uint64_t) * 2 * (size_t)T);

    float scene[8];
    for (int k = 0; k < 8; ++k) scene[k] = -INFINITY;
    /* Bvh2.cpp:701-750 (scalar form of :565-628) */
    for (uint32_t i = 0; i < T; ++i) {
        const float* p0 = vertices + (size_t)indices[i * 3 + 0] * 4;
        const float* p1 = vertices + (size_t)indices[i * 3 + 1] * 4;
        const float* p2 = vertices + (size_t)indices[i * 3 + 2] * 4;
        float* b = s.tb + (size_t)i * 8;
        for (int k = 0; k < 4; ++k) {
            const float mn = omin(omin(p0[k], p1[k]), p2[k]);
            const float mx = omax(omax(p0[k], p1[k]), p2[k]);
            b[k] = -mn; b[4 + k] = mx;
        }
        max8(scene, b);
    }
    for (unsigned dim = 0; dim < 3; ++dim) {
        for (uint32_t i = 0; i < T; ++i) {
            const float* b = s.tb + (size_t)i * 8;
            const float mid = (-b[dim] + b[4 + dim]) * 0.5f;
            uint32_t enc = f2u(mid);
            enc ^= ((int32_t)enc < 0) ? 0xFFFFFFFFu : 0x80000000u; /* Bvh2.cpp:743-745 */
            keys[i] = ((uint64_t)enc << 32) | i;
        }
        radix_sort_high32(keys, keys + T, T); /* 4 passes: result back in keys */
        for (uint32_t i = 0; i < T; ++i) s.sorted[dim][i] = (uint32_t)keys[i];
    }
    free(keys);

    /* Bvh2.cpp:882-891 */
    memset(&nodes[0], 0, sizeof(nodes[0]));
    nodes[0].kind = 0; nodes[0].parent = 0xFFFFFFFFu; nodes[0].first = 0; nodes[0].last = T;
    nodes[0].bbMin[0] = -scene[0]; nodes[0].bbMin[1] = -scene[1]; nodes[0].bbMin[2] = -scene[2];
    nodes[0].dummy0 = f2u(-scene[3]);
    nodes[0].bbMax[0] = scene[4]; nodes[0].bbMax[1] = scene[5]; nodes[0].bbMax[2] = scene[6];
    nodes[0].dummy1 = f2u(scene[7]);

    s.counter = 0;
    build_node(&s, 0);
    *nodeCount = s.counter * 2 + 1; /* Bvh2.cpp:901 */

    free(s.sorted[1]); free(s.sorted[2]); free(s.temp); free(s.accSah); free(s.left); free(s.tb);
    return 0;
}

/* ===================================================== scene pack / flatten */

/* Scene.cpp:109-120 */
static int find_shared_edge(const uint32_t* tri0, const uint32_t* tri1, unsigned* e0, unsigned* e1) {
    for (unsigned a = 0; a < 3; ++a)
        for (unsigned b = 0; b < 3; ++b)
            if (tri0[a] == tri1[(b + 1) % 3] && tri0[(a + 1) % 3] == tri1[b]) { *e0 = a; *e1 = b; return 1; }
    return 0;
}

static void make_pair(orc_pair* out, const float* p0, const float* p1, const float* p2, const float* p3) {
    /* Scene.cpp:149-153 / 174-178 */
    for (int k = 0; k < 3; ++k) { out->e1[k] = p0[k] - p1[k]; out->e2[k] = p2[k] - p0[k]; out->p0[k] = p0[k]; }
    out->e3x = p3[0] - p0[0]; out->e3y = p3[1] - p0[1]; out->e3z = p3[2] - p0[2];
}

int orc_scene_pack(orc_bvh2_node* nodes, uint32_t nodeCount, const uint32_t* triangles,
                   const float* vertices, const uint32_t* indices, uint32_t T,
                   orc_gpu_node* gpuNodes, uint32_t* gpuNodeCount,
                   orc_pair* pairs, uint32_t* pairCount, uint32_t* pairCountPadded,
                   uint32_t* remap) {
    if (T >= (1u << 30)) return -4;
    if (!nodeCount || !nodes[0].kind) return -1; /* root must be inner (Kernels.h:164) */
    uint32_t nPairs = 0;
    uint32_t cand[128];
    for (uint32_t i = 0; i < 2 * T; ++i) remap[i] = 0; /* Scene.cpp:264-271: unset entries become 0 */

    /* Scene.cpp:237-261 */
    for (uint32_t i = 0; i < nodeCount; ++i) {
        if (nodes[i].kind) continue;
        uint32_t nc = 0;
        if (nodes[i].last - nodes[i].first > 127) return -2;
        for (uint32_t j = nodes[i].first; j < nodes[i].last; ++j) cand[nc++] = triangles[j];
        nodes[i].first = nPairs;
        uint32_t head = 0;
        while (head < nc) {
            const uint32_t firstIdx = cand[head++];
            const uint32_t* ft = indices + (size_t)firstIdx * 3;
            int merged = 0;
            /* Scene.cpp:127-157 */
            for (uint32_t c = head; c < nc; ++c) {
                const uint32_t* st = indices + (size_t)cand[c] * 3;
                unsigned e0, e1;
                if (find_shared_edge(ft, st, &e0, &e1)) {
                    remap[nPairs * 2] = firstIdx | (e0 << 30);
                    remap[nPairs * 2 + 1] = cand[c] | ((e1 + 1) << 30);
                    const float* p0 = vertices + (size_t)ft[e0] * 4;
                    const float* p1 = vertices + (size_t)ft[(e0 + 1) % 3] * 4;
                    const float* p2 = vertices + (size_t)ft[(e0 + 2) % 3] * 4;
                    const float* p3 = vertices + (size_t)st[(e1 + 2) % 3] * 4;
                    make_pair(&pairs[nPairs], p0, p1, p2, p3);
                    for (uint32_t k = c; k + 1 < nc; ++k) cand[k] = cand[k + 1];
                    --nc;
                    merged = 1;
                    break;
                }
            }
            if (!merged) { /* Scene.cpp:160-180: degenerate second triangle, p3 = p1 */
                remap[nPairs * 2] = firstIdx;
                const float* p0 = vertices + (size_t)ft[0] * 4;
                const float* p1 = vertices + (size_t)ft[1] * 4;
                const float* p2 = vertices + (size_t)ft[2] * 4;
                make_pair(&pairs[nPairs], p0, p1, p2, p1);
            }
            ++nPairs;
        }
        nodes[i].last = nPairs;
    }
    if (nPairs >= (1u << 24)) return -3;

    /* Scene.cpp:275-332 */
    uint32_t* indexRemap = (uint32_t*)malloc(sizeof(uint32_t) * nodeCount);
    uint32_t nInner = 0;
    for (uint32_t i = 0; i < nodeCount; ++i) {
        const orc_bvh2_node* n = &nodes[i];
        if (!n->kind) continue;
        const orc_bvh2_node* l = &nodes[n->first];
        const orc_bvh2_node* r = &nodes[n->last];
        indexRemap[i] = nInner;
        orc_gpu_node* g = &gpuNodes[nInner++];
        g->kind = n->kind; g->parent = n->parent;
        g->first = l->kind ? (n->first | 0x80000000u) : (((l->last - l->first) << 24) | l->first);
        g->last = r->kind ? (n->last | 0x80000000u) : (((r->last - r->first) << 24) | r->first);
        for (int k = 0; k < 3; ++k) {
            g->leftMin[k] = l->bbMin[k]; g->leftMax[k] = l->bbMax[k];
            g->rightMin[k] = r->bbMin[k]; g->rightMax[k] = r->bbMax[k];
        }
    }
    for (uint32_t i = 0; i < nInner; ++i) {
        if (gpuNodes[i].first & 0x80000000u) gpuNodes[i].first = 0x80000000u | indexRemap[gpuNodes[i].first & 0x7FFFFFFFu];
        if (gpuNodes[i].last & 0x80000000u) gpuNodes[i].last = 0x80000000u | indexRemap[gpuNodes[i].last & 0x7FFFFFFFu];
    }
    free(indexRemap);

    /* Scene.cpp:334-338: pad with copies of pair 0 until 3*N % 32 == 0 (at least one) */
    uint32_t padded = nPairs;
    do { pairs[padded++] = pairs[0]; } while ((padded * 3) % 32 != 0);

    *gpuNodeCount = nInner; *pairCount = nPairs; *pairCountPadded = padded;
    return 0;
}

/* ========================================================= brute-force arbiter */
/* Independent of everything above: textbook Moller-Trumbore in double on the
 * ORIGINAL vertices (not the edge-vector pairs), no direction clamp. */
static int mt_double(const float* a, const float* b, const float* c, const orc_ray* r,
                     double* t, double* u, double* v) {
    const double e1[3] = { (double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2] };
    const double e2[3] = { (double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2] };
    const double d[3] = { r->dir[0], r->dir[1], r->dir[2] };
    const double p[3] = { d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0] };
    const double det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
    if (det == 0.0) return 0;
    const double inv = 1.0 / det;
    const double s[3] = { (double)r->origin[0] - a[0], (double)r->origin[1] - a[1], (double)r->origin[2] - a[2] };
    const double uu = (s[0] * p[0] + s[1] * p[1] + s[2] * p[2]) * inv;
    if (uu < 0.0 || uu > 1.0) return 0;
    const double q[3] = { s[1] * e1[2] - s[2] * e1[1], s[2] * e1[0] - s[0] * e1[2], s[0] * e1[1] - s[1] * e1[0] };
    const double vv = (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]) * inv;
    if (vv < 0.0 || uu + vv > 1.0) return 0;
    const double tt = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) * inv;
    if (!(tt > (double)r->minT && tt <= (double)r->maxT)) return 0;
    *t = tt; *u = uu; *v = vv;
    return 1;
}

int orc_brute_one(const float* vertices, const uint32_t* indices, uint32_t triangle,
                  const orc_ray* ray, double* t, double* u, double* v) {
    const uint32_t* tri = indices + (size_t)triangle * 3;
    return mt_double(vertices + (size_t)tri[0] * 4, vertices + (size_t)tri[1] * 4, vertices + (size_t)tri[2] * 4, ray, t, u, v);
}

void orc_brute_closest(const float* vertices, const uint32_t* indices, uint32_t T,
                       const orc_ray* rays, uint32_t count,
                       uint32_t* tri, double* t, double* u, double* v, double* t2) {
    for (uint32_t i = 0; i < count; ++i) {
        double bt = INFINITY, bu = 0, bv = 0, st = INFINITY;
        uint32_t best = 0xFFFFFFFFu;
        for (uint32_t k = 0; k < T; ++k) {
            double tt, uu, vv;
            if (!orc_brute_one(vertices, indices, k, rays + i, &tt, &uu, &vv)) continue;
            if (tt < bt) { st = bt; bt = tt; bu = uu; bv = vv; best = k; }
            else if (tt < st) st = tt;
        }
        tri[i] = best; t[i] = bt; u[i] = bu; v[i] = bv;
        if (t2) t2[i] = st;
    }
}

/* ======================================================================================================================
 * Path-tracing consumer: scalar restatement of the reference's material sampling, one lane of
 * ReflectiveDiffuseMaterial::sample8 (Renderer/Materials.cpp:39-151) with its own sine / cosine approximations
 * (Materials.cpp:11-29), operation by operation: _mm256_fmadd_ps -> fmaf, _mm256_fmsub_ps(a,b,c) -> fmaf(a,b,-c),
 * blendv(a,b,m) -> sign bit of m picks b.  The two hardware approximations the reference uses, _mm256_rcp_ps and
 * _mm256_rsqrt_ps (12-bit, implementation defined), are evaluated exactly here (1/x, 1/sqrt(x)).  Test infrastructure:
 * tests/test_host_build.py holds the product's scalar re-derivation (rayaccel_amd/csrc/pt_shade.h sampleMaterial) to this.
 * exact_trig != 0 replaces the reference's parabola sine/cosine (|error| up to 0.056) by sin/cos of 2*pi*r: pt_shade.h
 * deliberately uses exact values there, so the comparison is tight with the flag and loose (the parabola's error) without.
 * ====================================================================================================================== */
static float pt_sin_approx(float x) {                      /* Materials.cpp:11-22: sin(2 pi x), x in [0,1) */
    const float y = fmaf(-16.0f, x, 8.0f);
    const int gt = x >= 0.5f;
    float xy = x * y;
    const float z = gt ? y : 0.0f;
    if (gt) xy = -xy;
    return xy + z;
}
static float pt_cos_approx(float x) {                      /* Materials.cpp:24-28 */
    const float y = x - 0.75f;
    x = signbit(y) ? x + 0.25f : y;                        /* blendv(y, x + 0.25, y): sign of y picks x + 0.25 */
    return pt_sin_approx(x);
}

/* ke = {kd.r, kd.g, kd.b, eta} (Materials.cpp:32-37); rnd, normal, wo: 3 floats each; out: wi[3], colour[3]; returns 1 if the
 * diffuse direction was chosen (the mask of Materials.cpp:128). */
int orc_pt_sample_material(const float* ke, const float* rnd, const float* normal, const float* wo, int exact_trig, float* wi, float* colour) {
    const float nx = normal[0], ny = normal[1], nz = normal[2];
    float cosi = fmaf(nz, wo[2], fmaf(ny, wo[1], nx * wo[0]));                     /* :57 */
    cosi = cosi > 0.0f ? cosi : 0.0f;                                              /* :58 (max_ps(cosi, 0): NaN -> 0) */
    const float two_cosi = 2.0f * cosi;
    const float rx = fmaf(two_cosi, nx, -wo[0]), ry = fmaf(two_cosi, ny, -wo[1]), rz = fmaf(two_cosi, nz, -wo[2]);   /* :60-62 */
    const float eta = ke[3];
    const float cosi2_minus_one = fmaf(cosi, cosi, -1.0f);                         /* :67 */
    const float eta2 = eta * eta;
    const float k = fmaf(eta2, cosi2_minus_one, 1.0f);                             /* :69 */
    const float cost = sqrtf(k);                                                   /* :70 (NaN for k < 0: blended away below) */
    const float rper = fmaf(eta, cosi, -cost) * (1.0f / fmaf(eta, cosi, cost));    /* :74 */
    const float rpar = -(fmaf(eta, cost, -cosi) * (1.0f / fmaf(eta, cost, cosi))); /* :75 */
    float fresnel = 0.5f * fmaf(rpar, rpar, rper * rper);                          /* :77-78 */
    if (signbit(k)) fresnel = 1.0f;                                                /* :79 */
    const int baseMask = !(fabsf(nx) <= 0.1f);                                     /* :82-83 (_CMP_NLE_US) */
    float ux = baseMask ? -nz : 0.0f, uy = baseMask ? 0.0f : -nz, uz = baseMask ? nx : ny;     /* :85-88 */
    const float fb = 1.0f / sqrtf(fmaf(uz, uz, fmaf(uy, uy, ux * ux)));            /* :90 */
    ux *= fb; uy *= fb; uz *= fb;
    const float vx = fmaf(ny, uz, -(nz * uy)), vy = fmaf(nz, ux, -(nx * uz)), vz = fmaf(nx, uy, -(ny * ux));          /* :96-98 */
    float sinX, cosX;
    if (exact_trig) { sinX = (float)sin(6.283185307179586 * (double)rnd[0]); cosX = (float)cos(6.283185307179586 * (double)rnd[0]); }
    else { sinX = pt_sin_approx(rnd[0]); cosX = pt_cos_approx(rnd[0]); }           /* :100-101 */
    const float r2 = rnd[1], r2s = sqrtf(r2), sq1 = sqrtf(1.0f - r2);              /* :103-106 */
    float dx = fmaf(nx, sq1, fmaf(ux, cosX, vx * sinX) * r2s);                     /* :108-110 */
    float dy = fmaf(ny, sq1, fmaf(uy, cosX, vy * sinX) * r2s);
    float dz = fmaf(nz, sq1, fmaf(uz, cosX, vz * sinX) * r2s);
    const float fd = 1.0f / sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));            /* :112 */
    dx *= fd; dy *= fd; dz *= fd;
    float r = ke[0], g = ke[1], b = ke[2];
    const float s0 = fresnel * 3.0f, s1 = b + (r + g), sum = s0 + s1;              /* :123-125 */
    const int mask = rnd[2] * sum >= s0;                                           /* :127-128 (_CMP_GE_OQ) */
    wi[0] = mask ? dx : rx; wi[1] = mask ? dy : ry; wi[2] = mask ? dz : rz;        /* :130-132 */
    if (!mask) r = g = b = fresnel;                                                /* :134-136 */
    const float scale = sum * (1.0f / (b + (r + g)));                              /* :138 */
    colour[0] = r * scale; colour[1] = g * scale; colour[2] = b * scale;
    return mask;
}
