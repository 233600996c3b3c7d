/*
 * racc_oracle_simd.c — the oracle's traversal (racc_oracle.c: Kernels.h:139-242 restated) EIGHT RAYS AT A TIME in AVX2.
 *
 * TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg, kind "simd-port"; tests/test_oracle.py holds it to the scalar port bit for bit).
 *
 * Why: the reference's CPU leg hands its intersector eight rays at a time (Scene.cpp:386-428: AoS -> SoA, rtcIntersect8 with an all-ones
 * mask, isa=avx2: RayAccelerator.cpp:422); a scalar port understates what the host cores do with the same algorithm.  This is NOT Embree
 * (no bvh8, no triangle4 leaves, no packet/frustum culling): it is the scalar port's algorithm with eight independent rays in the eight
 * lanes of a ymm register — every lane has its own node, its own stack and its own tFar and does, instruction for instruction, what
 * traverse_one() does for its ray:
 *   fmaf -> vfmadd (one rounding), a*b -> vmulps, -(a*b) -> vmulps + sign flip (never vfnmadd: that would round once instead of twice),
 *   omin/omax (a<b?a:b / a>b?a:b) -> vminps/vmaxps (same operand order: the second operand comes back on NaN or equality),
 *   1.0f/x -> vdivps (correctly rounded), comparisons ordered/unordered exactly as C's <, <=, >, != are.
 * Results are therefore BIT-IDENTICAL to orc_traverse for every ray (miss colours included: the epilogue is the scalar one).
 * Scheduling is the GPU kernel's in miniature: lanes that finish are refilled from the slice at once; one iteration runs an inner step
 * for the lanes at an inner node and a pair test for the lanes in a leaf.  The 64-byte node / 48-byte pair of each lane is loaded with
 * full-width loads and transposed in registers (no vgatherdps).
 */
#include "racc_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

int orc_simd_available(void) {
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
}

#define LANES 8
#define STACK 256

typedef struct {
    float ox[LANES], oy[LANES], oz[LANES], dx[LANES], dy[LANES], dz[LANES];
    float ix[LANES], iy[LANES], iz[LANES], ex[LANES], ey[LANES], ez[LANES];      /* invDir, OoD (Kernels.h:159-160) */
    float tNear[LANES], tFar[LANES], hu[LANES], hv[LANES];
    int32_t hidx[LANES];
    uint32_t node[LANES];      /* bit 31: inner ref; else (pairs left << 24 | current pair); 0 = lane idle */
    uint32_t head[LANES], ray[LANES];
    uint32_t stack[LANES][STACK];
} __attribute__((aligned(32))) lanes_t;

/* 8 rows of 8 floats (row k at p[k]) -> 8 columns.  The 128-bit halves of rows k and k + 4 are combined by the loads themselves (vinsertf128
 * from memory: a load-port operation), which leaves two 4x4 transposes inside the 128-bit lanes: 16 shuffles instead of 24 — the shuffle port is
 * what this routine is bound by. */
static inline void load_transpose8(const float* const p[8], int off, __m256 c[8]) {
#define ROW2(a, b, h) _mm256_insertf128_ps(_mm256_castps128_ps256(_mm_loadu_ps(p[a] + off + 4 * (h))), _mm_loadu_ps(p[b] + off + 4 * (h)), 1)
    const __m256 r0 = ROW2(0, 4, 0), r1 = ROW2(1, 5, 0), r2 = ROW2(2, 6, 0), r3 = ROW2(3, 7, 0);      /* columns 0-3 of rows 0-3 | 4-7 */
    const __m256 q0 = ROW2(0, 4, 1), q1 = ROW2(1, 5, 1), q2 = ROW2(2, 6, 1), q3 = ROW2(3, 7, 1);      /* columns 4-7 */
#undef ROW2
    __m256 t0 = _mm256_unpacklo_ps(r0, r1), t1 = _mm256_unpackhi_ps(r0, r1), t2 = _mm256_unpacklo_ps(r2, r3), t3 = _mm256_unpackhi_ps(r2, r3);
    c[0] = _mm256_shuffle_ps(t0, t2, 0x44); c[1] = _mm256_shuffle_ps(t0, t2, 0xEE); c[2] = _mm256_shuffle_ps(t1, t3, 0x44); c[3] = _mm256_shuffle_ps(t1, t3, 0xEE);
    t0 = _mm256_unpacklo_ps(q0, q1); t1 = _mm256_unpackhi_ps(q0, q1); t2 = _mm256_unpacklo_ps(q2, q3); t3 = _mm256_unpackhi_ps(q2, q3);
    c[4] = _mm256_shuffle_ps(t0, t2, 0x44); c[5] = _mm256_shuffle_ps(t0, t2, 0xEE); c[6] = _mm256_shuffle_ps(t1, t3, 0x44); c[7] = _mm256_shuffle_ps(t1, t3, 0xEE);
}

/* 8 rows of 4 floats (one __m128 per lane) -> 4 columns of 8 */
static inline void transpose8x4(const __m128 in[8], __m256 out[4]) {
    const __m256 a = _mm256_insertf128_ps(_mm256_castps128_ps256(in[0]), in[4], 1);      /* lanes 0 | 4 */
    const __m256 b = _mm256_insertf128_ps(_mm256_castps128_ps256(in[1]), in[5], 1);
    const __m256 c = _mm256_insertf128_ps(_mm256_castps128_ps256(in[2]), in[6], 1);
    const __m256 d = _mm256_insertf128_ps(_mm256_castps128_ps256(in[3]), in[7], 1);
    const __m256 t0 = _mm256_unpacklo_ps(a, b), t1 = _mm256_unpackhi_ps(a, b), t2 = _mm256_unpacklo_ps(c, d), t3 = _mm256_unpackhi_ps(c, d);
    out[0] = _mm256_shuffle_ps(t0, t2, 0x44); out[1] = _mm256_shuffle_ps(t0, t2, 0xEE);
    out[2] = _mm256_shuffle_ps(t1, t3, 0x44); out[3] = _mm256_shuffle_ps(t1, t3, 0xEE);
}

static inline __m256 neg(__m256 x) { return _mm256_xor_ps(x, _mm256_castsi256_ps(_mm256_set1_epi32((int)0x80000000u))); }
static inline __m256 vabs(__m256 x) { return _mm256_and_ps(x, _mm256_castsi256_ps(_mm256_set1_epi32(0x7FFFFFFF))); }
/* dot3(a, b) = fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)) */
static inline __m256 dot3v(__m256 ax, __m256 ay, __m256 az, __m256 bx, __m256 by, __m256 bz) {
    return _mm256_fmadd_ps(az, bz, _mm256_fmadd_ps(ay, by, _mm256_mul_ps(ax, bx)));
}
/* mad_cross: r.x = fmaf(a.y, b.z, -(a.z * b.y)), ... (Kernels.h:23-25) */
#define CROSS(rx, ry, rz, ax, ay, az, bx, by, bz)                      \
    do {                                                               \
        rx = _mm256_fmadd_ps(ay, bz, neg(_mm256_mul_ps(az, by)));      \
        ry = _mm256_fmadd_ps(az, bx, neg(_mm256_mul_ps(ax, bz)));      \
        rz = _mm256_fmadd_ps(ax, by, neg(_mm256_mul_ps(ay, bx)));      \
    } while (0)

/* one box of every lane: aabb_intersect() of racc_oracle.c */
static inline __m256 slab(__m256 mnx, __m256 mny, __m256 mnz, __m256 mxx, __m256 mxy, __m256 mxz,
                          __m256 ix, __m256 iy, __m256 iz, __m256 ex, __m256 ey, __m256 ez, __m256 tNear, __m256 tFar) {
    const __m256 ax = _mm256_fmadd_ps(mnx, ix, ex), bx = _mm256_fmadd_ps(mxx, ix, ex);
    const __m256 ay = _mm256_fmadd_ps(mny, iy, ey), by = _mm256_fmadd_ps(mxy, iy, ey);
    const __m256 az = _mm256_fmadd_ps(mnz, iz, ez), bz = _mm256_fmadd_ps(mxz, iz, ez);
    const __m256 nx = _mm256_min_ps(ax, bx), fx = _mm256_max_ps(ax, bx);
    const __m256 ny = _mm256_min_ps(ay, by), fy = _mm256_max_ps(ay, by);
    const __m256 nz = _mm256_min_ps(az, bz), fz = _mm256_max_ps(az, bz);
    const __m256 t0 = _mm256_max_ps(_mm256_max_ps(tNear, nx), _mm256_max_ps(ny, nz));
    const __m256 t1 = _mm256_min_ps(_mm256_min_ps(tFar, fx), _mm256_min_ps(fy, fz));
    return _mm256_blendv_ps(t0, tFar, _mm256_cmp_ps(t0, t1, _CMP_GT_OQ));      /* if (t0 > t1) return tFar */
}

/* the scalar prologue of traverse_one(): returns 0 when the ray is invalid (result written) */
static int load_ray(lanes_t* L, int k, const orc_ray* in, orc_result* out, uint32_t rayIndex) {
    float o[3] = { in->origin[0], in->origin[1], in->origin[2] }, d[3] = { in->dir[0], in->dir[1], in->dir[2] };
    const float tNear = in->minT, tFar = in->maxT;
    int finite = isfinite(tNear) && !isnan(tFar);
    for (int a = 0; a < 3; ++a) finite = finite && isfinite(o[a]) && isfinite(d[a]);
    if (!finite) { out->triangle = 0xFFFFFFFFu; out->t = out->u = out->v = 0.0f; return 0; }
    const float epsilon = 1e-10f;
    for (int a = 0; a < 3; ++a)
        if (fabsf(d[a]) < epsilon) d[a] = copysignf(epsilon, d[a]);
    float inv[3], ood[3];
    for (int a = 0; a < 3; ++a) { inv[a] = 1.0f / d[a]; ood[a] = -o[a] * inv[a]; }
    L->ox[k] = o[0]; L->oy[k] = o[1]; L->oz[k] = o[2]; L->dx[k] = d[0]; L->dy[k] = d[1]; L->dz[k] = d[2];
    L->ix[k] = inv[0]; L->iy[k] = inv[1]; L->iz[k] = inv[2]; L->ex[k] = ood[0]; L->ey[k] = ood[1]; L->ez[k] = ood[2];
    L->tNear[k] = tNear; L->tFar[k] = tFar; L->hu[k] = 0.0f; L->hv[k] = 0.0f; L->hidx[k] = -1;
    L->node[k] = 0x80000000u; L->head[k] = 0; L->ray[k] = rayIndex;
    return 1;
}

/* the scalar epilogue of traverse_one() (Kernels.h:213-239) */
static void store_result(const lanes_t* L, int k, const uint32_t* remap, const float* env, uint32_t envW, uint32_t envH, orc_result* out) {
    if (L->hidx[k] == -1) {
        const float d[3] = { L->dx[k], L->dy[k], L->dz[k] };
        float rgb[3];
        orc_env_sample(env, envW, envH, d, rgb);
        out->triangle = 0xFFFFFFFFu; out->t = rgb[0]; out->u = rgb[1]; out->v = rgb[2];
    } else {
        uint32_t index = remap[L->hidx[k]];
        const uint32_t edge = index >> 30;
        index &= 0x3FFFFFFFu;
        const float bx = L->hu[k], by = L->hv[k], bz = 1.0f - L->hu[k] - L->hv[k];
        float u = bx, v = by;
        if (edge == 1) { u = bz; v = bx; }
        else if (edge == 2) { u = by; v = bz; }
        out->triangle = index; out->t = L->tFar[k]; out->u = u; out->v = v;      /* (hit.t == ray.tFar after a hit) */
    }
}

#ifndef ORC_SIMD_LEAF_MIN
#define ORC_SIMD_LEAF_MIN 3      /* a pair test runs once this many lanes wait in a leaf (or no lane is at an inner node): the GPU kernel's vote, in miniature */
#endif

/* One scheduling iteration of one group of eight lanes; returns 0 when the group holds no ray and the slice has none left. */
static inline __attribute__((always_inline)) int simd_step(lanes_t* Lp, int* idleMaskP, uint32_t* nextP, uint32_t end,
                                                           const orc_gpu_node* nodes, const orc_pair* pairs, const uint32_t* remap,
                                                           const float* env, uint32_t envW, uint32_t envH, const orc_ray* rays, orc_result* results) {
#define L (*Lp)
    const __m256i signBit = _mm256_set1_epi32((int)0x80000000u);
    const __m256i laneBase = _mm256_setr_epi32(0, STACK, 2 * STACK, 3 * STACK, 4 * STACK, 5 * STACK, 6 * STACK, 7 * STACK);
    int idleMask = *idleMaskP;
    uint32_t next = *nextP;
    /* Everything per lane is branch-free (a lane's near/far/hit decisions are coin flips: as branches they cost more than the arithmetic). */
    {
        /* ---- refill idle lanes (rare: once per ray) */
        if (idleMask) {
            for (int m = idleMask; m; m &= m - 1) {
                const int k = __builtin_ctz(m);
                while (L.node[k] == 0u && next < end) {
                    const uint32_t r = next++;
                    load_ray(&L, k, rays + r, results + r, r);      /* invalid rays are answered there; the lane stays idle and takes the next one */
                }
            }
        }
        __m256i node = _mm256_load_si256((const __m256i*)L.node);
        const int innerMask = _mm256_movemask_ps(_mm256_castsi256_ps(node));
        int leafMask = _mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpgt_epi32(node, _mm256_setzero_si256())));      /* 0 < node < 2^31 */
        *nextP = next;
        if (!(innerMask | leafMask)) return 0;
        if (innerMask && __builtin_popcount(leafMask) < ORC_SIMD_LEAF_MIN) leafMask = 0;      /* postponed: those lanes sit this iteration out */
        const __m256 tNear = _mm256_load_ps(L.tNear);
        __m256 tFar = _mm256_load_ps(L.tFar);
        __m256i head = _mm256_load_si256((const __m256i*)L.head);
        __m256i popV = _mm256_setzero_si256();      /* lanes that take their next node from the stack (or finish) */

        if (innerMask) {      /* ---- Kernels.h:170-199 for the lanes at an inner node */
            const __m256i innerV = _mm256_srai_epi32(node, 31);
            uint32_t idx[LANES];
            _mm256_storeu_si256((__m256i*)idx, _mm256_and_si256(_mm256_and_si256(node, innerV), _mm256_set1_epi32(0x7FFFFFFF)));      /* other lanes: node 0 (exists) */
            __m256 A[8], B[8];
            const float* rows[8];
            for (int k = 0; k < LANES; ++k) rows[k] = (const float*)(nodes + idx[k]);
            load_transpose8(rows, 0, A); load_transpose8(rows, 8, B);      /* A: kind parent first last lminx lminy lminz lmaxx | B: lmaxy lmaxz rminx rminy rminz rmaxx rmaxy rmaxz */
            const __m256 ix = _mm256_load_ps(L.ix), iy = _mm256_load_ps(L.iy), iz = _mm256_load_ps(L.iz);
            const __m256 ex = _mm256_load_ps(L.ex), ey = _mm256_load_ps(L.ey), ez = _mm256_load_ps(L.ez);
            const __m256 tFirst = slab(A[4], A[5], A[6], A[7], B[0], B[1], ix, iy, iz, ex, ey, ez, tNear, tFar);
            const __m256 tLast = slab(B[2], B[3], B[4], B[5], B[6], B[7], ix, iy, iz, ex, ey, ez, tNear, tFar);
            const __m256 firstDiff = _mm256_sub_ps(tFar, tFirst), lastDiff = _mm256_sub_ps(tFar, tLast);
            const __m256i any = _mm256_and_si256(innerV, _mm256_castps_si256(_mm256_cmp_ps(_mm256_add_ps(firstDiff, lastDiff), _mm256_setzero_ps(), _CMP_NEQ_UQ)));      /* Kernels.h:192 */
            const __m256 sgn = _mm256_cmp_ps(tLast, tFirst, _CMP_LT_OQ);                                                 /* signbit(tLast - tFirst), :193 */
            const __m256i both = _mm256_and_si256(any, _mm256_castps_si256(_mm256_cmp_ps(_mm256_max_ps(tFirst, tLast), tFar, _CMP_NEQ_UQ)));      /* fmax(tFirst, tLast) != tRay, :194 */
            const __m256 nearKid = _mm256_blendv_ps(A[2], A[3], sgn), farKid = _mm256_blendv_ps(A[3], A[2], sgn);
            /* push the far child where both are hit: written above every lane's top unconditionally, counted by the head (stack[64] in the reference; 256 here, clamped like the scalar port) */
            uint32_t fk[LANES], hd[LANES];
            _mm256_storeu_ps((float*)fk, farKid);
            _mm256_storeu_si256((__m256i*)hd, _mm256_min_epu32(head, _mm256_set1_epi32(STACK - 1)));
            for (int k = 0; k < LANES; ++k) L.stack[k][hd[k]] = fk[k];
            head = _mm256_sub_epi32(head, _mm256_and_si256(both, _mm256_cmpgt_epi32(_mm256_set1_epi32(STACK), head)));      /* += 1 where both && head < STACK */
            node = _mm256_blendv_epi8(node, _mm256_castps_si256(nearKid), any);
            popV = _mm256_andnot_si256(any, innerV);
        }

        if (leafMask) {       /* ---- Kernels.h:200-205 + 36-115: one pair of every lane that is in a leaf */
            const __m256i leafV = _mm256_cmpgt_epi32(_mm256_load_si256((const __m256i*)L.node), _mm256_setzero_si256());
            const __m256 ox = _mm256_load_ps(L.ox), oy = _mm256_load_ps(L.oy), oz = _mm256_load_ps(L.oz);
            const __m256 dx = _mm256_load_ps(L.dx), dy = _mm256_load_ps(L.dy), dz = _mm256_load_ps(L.dz);
            const __m256i curV = _mm256_and_si256(_mm256_and_si256(node, leafV), _mm256_set1_epi32(0xFFFFFF));
            uint32_t cur[LANES];
            _mm256_storeu_si256((__m256i*)cur, curV);
            __m128 c0[8], c1[8], c2[8];
            for (int k = 0; k < LANES; ++k) {
                const float* p = (const float*)(pairs + cur[k]);
                c0[k] = _mm_loadu_ps(p); c1[k] = _mm_loadu_ps(p + 4); c2[k] = _mm_loadu_ps(p + 8);
            }
            __m256 E1[4], E2[4], P0[4];      /* e1.xyz e3.x | e2.xyz e3.y | p0.xyz e3.z */
            transpose8x4(c0, E1); transpose8x4(c1, E2); transpose8x4(c2, P0);
            const __m256 e1x = E1[0], e1y = E1[1], e1z = E1[2], e3x = E1[3], e2x = E2[0], e2y = E2[1], e2z = E2[2], e3y = E2[3], e3z = P0[3];
            __m256 n1x, n1y, n1z, n2x, n2y, n2z, Rx, Ry, Rz;
            CROSS(n1x, n1y, n1z, e1x, e1y, e1z, e2x, e2y, e2z);
            CROSS(n2x, n2y, n2z, e3x, e3y, e3z, e1x, e1y, e1z);
            const __m256 Cx = _mm256_sub_ps(P0[0], ox), Cy = _mm256_sub_ps(P0[1], oy), Cz = _mm256_sub_ps(P0[2], oz);
            CROSS(Rx, Ry, Rz, dx, dy, dz, Cx, Cy, Cz);
            const __m256 det1 = dot3v(n1x, n1y, n1z, dx, dy, dz), det2 = dot3v(n2x, n2y, n2z, dx, dy, dz);
            const __m256 s1 = _mm256_and_ps(det1, _mm256_castsi256_ps(signBit)), s2 = _mm256_and_ps(det2, _mm256_castsi256_ps(signBit));
            const __m256 re1 = dot3v(Rx, Ry, Rz, e1x, e1y, e1z);
            const __m256 U1 = _mm256_xor_ps(dot3v(Rx, Ry, Rz, e2x, e2y, e2z), s1), V1 = _mm256_xor_ps(re1, s1);
            const __m256 U2 = _mm256_xor_ps(neg(re1), s2), V2 = _mm256_xor_ps(neg(dot3v(Rx, Ry, Rz, e3x, e3y, e3z)), s2);
            __m256 out1 = _mm256_castsi256_ps(_mm256_srai_epi32(_mm256_castps_si256(_mm256_or_ps(U1, V1)), 31));      /* (int)(iU1 | iV1) < 0 */
            __m256 out2 = _mm256_castsi256_ps(_mm256_srai_epi32(_mm256_castps_si256(_mm256_or_ps(U2, V2)), 31));
            const __m256 a1 = vabs(det1), a2 = vabs(det2);
            const __m256 W1 = _mm256_sub_ps(_mm256_sub_ps(a1, U1), V1), W2 = _mm256_sub_ps(_mm256_sub_ps(a2, U2), V2);
            const __m256 T1 = _mm256_xor_ps(dot3v(n1x, n1y, n1z, Cx, Cy, Cz), s1), T2 = _mm256_xor_ps(dot3v(n2x, n2y, n2z, Cx, Cy, Cz), s2);
            const __m256 zero = _mm256_setzero_ps();
            out1 = _mm256_or_ps(out1, _mm256_or_ps(_mm256_cmp_ps(W1, zero, _CMP_LT_OQ),
                                _mm256_or_ps(_mm256_cmp_ps(T1, _mm256_mul_ps(a1, tNear), _CMP_LE_OQ), _mm256_cmp_ps(T1, _mm256_mul_ps(a1, tFar), _CMP_GT_OQ))));
            out2 = _mm256_or_ps(out2, _mm256_or_ps(_mm256_cmp_ps(W2, zero, _CMP_LT_OQ),
                                _mm256_or_ps(_mm256_cmp_ps(T2, _mm256_mul_ps(a2, tNear), _CMP_LE_OQ), _mm256_cmp_ps(T2, _mm256_mul_ps(a2, tFar), _CMP_GT_OQ))));
            /* second triangle wins: (!out2 && out1) || (!out1 && !out2 && T1 * absDet2 > T2 * absDet1)   (Kernels.h:97) */
            const __m256 nearer2 = _mm256_cmp_ps(_mm256_mul_ps(T1, a2), _mm256_mul_ps(T2, a1), _CMP_GT_OQ);
            const __m256 second = _mm256_andnot_ps(out2, _mm256_or_ps(out1, nearer2));      /* = !out2 && (out1 || nearer2) */
            const __m256 hitM = _mm256_andnot_ps(_mm256_and_ps(out1, out2), _mm256_castsi256_ps(leafV));      /* leaf lane && !(out1 && out2) */
            const __m256 ad = _mm256_blendv_ps(a1, a2, second), Ts = _mm256_blendv_ps(T1, T2, second);
            const __m256 Us = _mm256_blendv_ps(U1, U2, second), Vs = _mm256_blendv_ps(V1, V2, second);
            const __m256 rcp = _mm256_div_ps(_mm256_set1_ps(1.0f), ad);
            tFar = _mm256_blendv_ps(tFar, _mm256_mul_ps(Ts, rcp), hitM);
            _mm256_store_ps(L.tFar, tFar);
            _mm256_store_ps(L.hu, _mm256_blendv_ps(_mm256_load_ps(L.hu), _mm256_mul_ps(Us, rcp), hitM));
            _mm256_store_ps(L.hv, _mm256_blendv_ps(_mm256_load_ps(L.hv), _mm256_mul_ps(Vs, rcp), hitM));
            const __m256i which = _mm256_sub_epi32(_mm256_add_epi32(curV, curV), _mm256_castps_si256(second));      /* pair * 2 + (second ? 1 : 0)  (a set mask is -1) */
            _mm256_store_si256((__m256i*)L.hidx, _mm256_blendv_epi8(_mm256_load_si256((const __m256i*)L.hidx), which, _mm256_castps_si256(hitM)));
            const __m256i more = _mm256_and_si256(leafV, _mm256_cmpgt_epi32(node, _mm256_set1_epi32(0x1FFFFFF)));      /* pairs left in this leaf */
            node = _mm256_sub_epi32(node, _mm256_and_si256(more, _mm256_set1_epi32(0xFFFFFF)));                       /* (count - 1, first + 1) */
            popV = _mm256_or_si256(popV, _mm256_andnot_si256(more, leafV));
        }

        /* ---- pop, or finish (Kernels.h:207-210) */
        const __m256i has = _mm256_and_si256(popV, _mm256_cmpgt_epi32(head, _mm256_setzero_si256()));
        head = _mm256_add_epi32(head, has);      /* -= 1 where popping from a non-empty stack */
        const __m256i popped = _mm256_mask_i32gather_epi32(_mm256_setzero_si256(), (const int*)&L.stack[0][0], _mm256_add_epi32(laneBase, head), has, 4);
        node = _mm256_blendv_epi8(node, popped, has);
        const __m256i done = _mm256_andnot_si256(has, popV);
        node = _mm256_andnot_si256(done, node);      /* finished lanes: idle */
        _mm256_store_si256((__m256i*)L.node, node);
        _mm256_store_si256((__m256i*)L.head, head);
        idleMask = _mm256_movemask_ps(_mm256_castsi256_ps(_mm256_cmpeq_epi32(node, _mm256_setzero_si256())));
        for (int m = _mm256_movemask_ps(_mm256_castsi256_ps(done)); m; m &= m - 1) {
            const int k = __builtin_ctz(m);
            store_result(&L, k, remap, env, envW, envH, results + L.ray[k]);
        }
        *idleMaskP = idleMask;
    }
    return 1;
#undef L
}

/* ORC_SIMD_GROUPS groups of eight rays in turn: their dependent chains (node fetch -> slab tests -> next node) are independent, so the core
 * overlaps one group's loads with the others' arithmetic (2 measured best; a software prefetch of every lane's next record measured 20 % slower). */
#ifndef ORC_SIMD_GROUPS
#define ORC_SIMD_GROUPS 2
#endif
void orc_traverse_simd(const orc_gpu_node* nodes, const orc_pair* pairs, const uint32_t* remap,
                       const float* env, uint32_t envW, uint32_t envH,
                       const orc_ray* rays, orc_result* results, uint32_t start, uint32_t end) {
    static __thread lanes_t G[ORC_SIMD_GROUPS];
    memset(G, 0, sizeof(G));
    uint32_t next = start;
    int idle[ORC_SIMD_GROUPS], alive[ORC_SIMD_GROUPS], any = 1;
    for (int g = 0; g < ORC_SIMD_GROUPS; ++g) { idle[g] = 0xFF; alive[g] = 1; }
    while (any) {
        any = 0;
        for (int g = 0; g < ORC_SIMD_GROUPS; ++g)
            if (alive[g]) any |= alive[g] = simd_step(&G[g], &idle[g], &next, end, nodes, pairs, remap, env, envW, envH, rays, results);
    }
}

typedef struct {
    const orc_gpu_node* nodes; const orc_pair* pairs; const uint32_t* remap;
    const float* env; uint32_t envW, envH;
    const orc_ray* rays; orc_result* results; uint32_t count, slice;
    uint64_t* cursor;
    uint32_t repeat;
    int wide;      /* 16 lanes (racc_oracle_simd512.c) instead of 8 */
} simd_job;

static void* simd_worker(void* arg) {
    simd_job* j = (simd_job*)arg;
    const uint64_t perPass = ((uint64_t)j->count + j->slice - 1) / j->slice;
    for (;;) {
        const uint64_t k = __atomic_fetch_add(j->cursor, 1, __ATOMIC_RELAXED);
        if (k >= perPass * j->repeat) break;
        const uint32_t s = (uint32_t)((k % perPass) * j->slice);
        const uint32_t e = s + j->slice < j->count ? s + j->slice : j->count;
        if (j->wide) orc_traverse_simd512(j->nodes, j->pairs, j->remap, j->env, j->envW, j->envH, j->rays, j->results, s, e);
        else orc_traverse_simd(j->nodes, j->pairs, j->remap, j->env, j->envW, j->envH, j->rays, j->results, s, e);
    }
    return 0;
}

/* orc_traverse_mt's contract (slices of `slice` rays — cpuTestBatch = 1024, RayAccelerator.cpp:438 — on `threads` pthreads, `repeat` passes) */
void orc_traverse_simd_mt(const orc_gpu_node* nodes, const orc_pair* pairs, const uint32_t* remap,
                          const float* env, uint32_t envW, uint32_t envH,
                          const orc_ray* rays, orc_result* results, uint32_t count,
                          uint32_t slice, uint32_t threads, uint32_t repeat, uint32_t width) {
    if (!slice) slice = 1024;
    if (!threads) threads = 1;
    if (!repeat) repeat = 1;
    uint64_t cursor = 0;
    simd_job job = { nodes, pairs, remap, env, envW, envH, rays, results, count, slice, &cursor, repeat, width == 16 && orc_simd512_available() };
    pthread_t* tid = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    for (uint32_t t = 1; t < threads; ++t) pthread_create(&tid[t], 0, simd_worker, &job);
    simd_worker(&job);
    for (uint32_t t = 1; t < threads; ++t) pthread_join(tid[t], 0);
    free(tid);
}
