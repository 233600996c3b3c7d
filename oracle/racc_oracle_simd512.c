/*
 * racc_oracle_simd512.c — racc_oracle_simd.c's traversal with SIXTEEN rays in the lanes of a zmm register (AVX-512F/DQ), for hosts that have it
 * (the GPU boxes' EPYC 9575F does).  TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg; tests/test_oracle.py: bit-identical to the scalar port).
 *
 * Same construction as the 8-wide file — every lane has its own node, stack and tFar and does, instruction for instruction, what traverse_one()
 * of racc_oracle.c does for its ray (vfmadd for fmaf, vmulps + sign flip for -(a*b), vminps/vmaxps in the scalar operand order, vdivps, C's
 * ordered / unordered comparisons as mask compares) — with what AVX-512 adds: a 64-byte node record IS one zmm row, so sixteen nodes are sixteen
 * loads and one 16 x 16 register transpose; decisions are mask registers instead of blend vectors.  The reference's own CPU leg is 8-wide
 * (Scene.cpp:386-428, isa=avx2); this is what the same algorithm does on the cores the GPU box actually has.
 */
#include "racc_oracle.h"

#include <immintrin.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

int orc_simd512_available(void) {
    __builtin_cpu_init();
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl");
}

#define LANES 16
#define STACK 256

typedef struct {
    float ox[LANES], oy[LANES], oz[LANES], dx[LANES], dy[LANES], dz[LANES];
    float ix[LANES], iy[LANES], iz[LANES], ex[LANES], ey[LANES], ez[LANES];
    float tNear[LANES], tFar[LANES], hu[LANES], hv[LANES];
    int32_t hidx[LANES];
    uint32_t node[LANES];      /* bit 31: inner ref; else (pairs left << 24 | current pair); 0 = lane idle */
    uint32_t head[LANES], ray[LANES];
    uint32_t stack[LANES][STACK];
} __attribute__((aligned(64))) lanes16_t;

/* 16 rows of 16 floats -> 16 columns (64 shuffles) */
static inline __attribute__((always_inline)) void transpose16(__m512 r[16]) {
    __m512 t[16], u[16], v[16];
    for (int k = 0; k < 8; ++k) { t[2 * k] = _mm512_unpacklo_ps(r[2 * k], r[2 * k + 1]); t[2 * k + 1] = _mm512_unpackhi_ps(r[2 * k], r[2 * k + 1]); }
    for (int k = 0; k < 4; ++k) {
        u[4 * k + 0] = _mm512_shuffle_ps(t[4 * k + 0], t[4 * k + 2], 0x44);
        u[4 * k + 1] = _mm512_shuffle_ps(t[4 * k + 0], t[4 * k + 2], 0xEE);
        u[4 * k + 2] = _mm512_shuffle_ps(t[4 * k + 1], t[4 * k + 3], 0x44);
        u[4 * k + 3] = _mm512_shuffle_ps(t[4 * k + 1], t[4 * k + 3], 0xEE);
    }
    for (int j = 0; j < 4; ++j) {
        v[j] = _mm512_shuffle_f32x4(u[j], u[4 + j], 0x88);
        v[4 + j] = _mm512_shuffle_f32x4(u[j], u[4 + j], 0xDD);
        v[8 + j] = _mm512_shuffle_f32x4(u[8 + j], u[12 + j], 0x88);
        v[12 + j] = _mm512_shuffle_f32x4(u[8 + j], u[12 + j], 0xDD);
    }
    for (int j = 0; j < 4; ++j) {
        r[j] = _mm512_shuffle_f32x4(v[j], v[8 + j], 0x88);
        r[8 + j] = _mm512_shuffle_f32x4(v[j], v[8 + j], 0xDD);
        r[4 + j] = _mm512_shuffle_f32x4(v[4 + j], v[12 + j], 0x88);
        r[12 + j] = _mm512_shuffle_f32x4(v[4 + j], v[12 + j], 0xDD);
    }
}

static inline __m512 neg(__m512 x) { return _mm512_castsi512_ps(_mm512_xor_si512(_mm512_castps_si512(x), _mm512_set1_epi32((int)0x80000000u))); }
static inline __m512 vabs(__m512 x) { return _mm512_castsi512_ps(_mm512_and_si512(_mm512_castps_si512(x), _mm512_set1_epi32(0x7FFFFFFF))); }
static inline __m512 vxor(__m512 a, __m512 b) { return _mm512_castsi512_ps(_mm512_xor_si512(_mm512_castps_si512(a), _mm512_castps_si512(b))); }
static inline __m512 dot3v(__m512 ax, __m512 ay, __m512 az, __m512 bx, __m512 by, __m512 bz) {
    return _mm512_fmadd_ps(az, bz, _mm512_fmadd_ps(ay, by, _mm512_mul_ps(ax, bx)));
}
#define CROSS(rx, ry, rz, ax, ay, az, bx, by, bz)                      \
    do {                                                               \
        rx = _mm512_fmadd_ps(ay, bz, neg(_mm512_mul_ps(az, by)));      \
        ry = _mm512_fmadd_ps(az, bx, neg(_mm512_mul_ps(ax, bz)));      \
        rz = _mm512_fmadd_ps(ax, by, neg(_mm512_mul_ps(ay, bx)));      \
    } while (0)

/* one box of every lane: aabb_intersect() of racc_oracle.c */
static inline __m512 slab(__m512 mnx, __m512 mny, __m512 mnz, __m512 mxx, __m512 mxy, __m512 mxz,
                          __m512 ix, __m512 iy, __m512 iz, __m512 ex, __m512 ey, __m512 ez, __m512 tNear, __m512 tFar) {
    const __m512 ax = _mm512_fmadd_ps(mnx, ix, ex), bx = _mm512_fmadd_ps(mxx, ix, ex);
    const __m512 ay = _mm512_fmadd_ps(mny, iy, ey), by = _mm512_fmadd_ps(mxy, iy, ey);
    const __m512 az = _mm512_fmadd_ps(mnz, iz, ez), bz = _mm512_fmadd_ps(mxz, iz, ez);
    const __m512 nx = _mm512_min_ps(ax, bx), fx = _mm512_max_ps(ax, bx);
    const __m512 ny = _mm512_min_ps(ay, by), fy = _mm512_max_ps(ay, by);
    const __m512 nz = _mm512_min_ps(az, bz), fz = _mm512_max_ps(az, bz);
    const __m512 t0 = _mm512_max_ps(_mm512_max_ps(tNear, nx), _mm512_max_ps(ny, nz));
    const __m512 t1 = _mm512_min_ps(_mm512_min_ps(tFar, fx), _mm512_min_ps(fy, fz));
    return _mm512_mask_blend_ps(_mm512_cmp_ps_mask(t0, t1, _CMP_GT_OQ), t0, tFar);      /* if (t0 > t1) return tFar */
}

/* the scalar prologue of traverse_one(): returns 0 when the ray is invalid (result written) */
static int load_ray(lanes16_t* L, int k, const orc_ray* in, orc_result* out, uint32_t rayIndex) {
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
static void store_result(const lanes16_t* L, int k, const uint32_t* remap, const float* env, uint32_t envW, uint32_t envH, orc_result* out) {
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
        out->triangle = index; out->t = L->tFar[k]; out->u = u; out->v = v;
    }
}

#ifndef ORC_SIMD512_LEAF_MIN
#define ORC_SIMD512_LEAF_MIN 4      /* a pair test runs once this many of the sixteen lanes wait in a leaf (or no lane is at an inner node) */
#endif

static inline __attribute__((always_inline)) int simd_step(lanes16_t* Lp, unsigned* idleMaskP, uint32_t* nextP, uint32_t end,
                                                           const orc_gpu_node* nodes, const orc_pair* pairs, const uint32_t* remap,
                                                           const float* env, uint32_t envW, uint32_t envH, const orc_ray* rays, orc_result* results) {
#define L (*Lp)
    const __m512i zero = _mm512_setzero_si512();
    const __m512i laneBase = _mm512_mullo_epi32(_mm512_setr_epi32(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15), _mm512_set1_epi32(STACK));
    uint32_t next = *nextP;
    for (unsigned m = *idleMaskP; m; m &= m - 1) {      /* ---- refill idle lanes (rare: once per ray) */
        const int k = __builtin_ctz(m);
        while (L.node[k] == 0u && next < end) {
            const uint32_t r = next++;
            load_ray(&L, k, rays + r, results + r, r);
        }
    }
    *nextP = next;
    __m512i node = _mm512_load_si512((const void*)L.node);
    const __mmask16 innerM = _mm512_cmplt_epi32_mask(node, zero);
    __mmask16 leafM = _mm512_cmpgt_epi32_mask(node, zero);      /* 0 < node < 2^31 */
    if (!(innerM | leafM)) return 0;
    if (innerM && __builtin_popcount(leafM) < ORC_SIMD512_LEAF_MIN) leafM = 0;      /* postponed: those lanes sit this iteration out */
    const __m512 tNear = _mm512_load_ps(L.tNear);
    __m512 tFar = _mm512_load_ps(L.tFar);
    __m512i head = _mm512_load_si512((const void*)L.head);
    __mmask16 popM = 0;

    if (innerM) {      /* ---- Kernels.h:170-199 for the lanes at an inner node */
        uint32_t idx[LANES];
        _mm512_storeu_si512((void*)idx, _mm512_maskz_and_epi32(innerM, node, _mm512_set1_epi32(0x7FFFFFFF)));      /* other lanes: node 0 (exists) */
        __m512 N[16];
        for (int k = 0; k < LANES; ++k) N[k] = _mm512_loadu_ps((const float*)(nodes + idx[k]));      /* one 64-byte record = one register */
        transpose16(N);      /* N: kind parent first last | lmin.xyz lmax.xyz | rmin.xyz rmax.xyz */
        const __m512 ix = _mm512_load_ps(L.ix), iy = _mm512_load_ps(L.iy), iz = _mm512_load_ps(L.iz);
        const __m512 ex = _mm512_load_ps(L.ex), ey = _mm512_load_ps(L.ey), ez = _mm512_load_ps(L.ez);
        const __m512 tFirst = slab(N[4], N[5], N[6], N[7], N[8], N[9], ix, iy, iz, ex, ey, ez, tNear, tFar);
        const __m512 tLast = slab(N[10], N[11], N[12], N[13], N[14], N[15], ix, iy, iz, ex, ey, ez, tNear, tFar);
        const __m512 firstDiff = _mm512_sub_ps(tFar, tFirst), lastDiff = _mm512_sub_ps(tFar, tLast);
        const __mmask16 any = innerM & _mm512_cmp_ps_mask(_mm512_add_ps(firstDiff, lastDiff), _mm512_setzero_ps(), _CMP_NEQ_UQ);      /* Kernels.h:192 */
        const __mmask16 sgn = _mm512_cmp_ps_mask(tLast, tFirst, _CMP_LT_OQ);                                                        /* :193 */
        const __mmask16 both = any & _mm512_cmp_ps_mask(_mm512_max_ps(tFirst, tLast), tFar, _CMP_NEQ_UQ);                            /* :194 */
        const __m512i first = _mm512_castps_si512(N[2]), last = _mm512_castps_si512(N[3]);
        const __m512i nearKid = _mm512_mask_blend_epi32(sgn, first, last), farKid = _mm512_mask_blend_epi32(sgn, last, first);
        uint32_t fk[LANES], hd[LANES];
        _mm512_storeu_si512((void*)fk, farKid);
        _mm512_storeu_si512((void*)hd, _mm512_min_epu32(head, _mm512_set1_epi32(STACK - 1)));
        for (int k = 0; k < LANES; ++k) L.stack[k][hd[k]] = fk[k];      /* above every lane's top, unconditionally; the head counts it */
        head = _mm512_mask_add_epi32(head, both & _mm512_cmplt_epu32_mask(head, _mm512_set1_epi32(STACK)), head, _mm512_set1_epi32(1));
        node = _mm512_mask_mov_epi32(node, any, nearKid);
        popM = innerM & (__mmask16)~any;
    }

    if (leafM) {       /* ---- Kernels.h:200-205 + 36-115: one pair of every lane that is in a leaf */
        const __m512 ox = _mm512_load_ps(L.ox), oy = _mm512_load_ps(L.oy), oz = _mm512_load_ps(L.oz);
        const __m512 dx = _mm512_load_ps(L.dx), dy = _mm512_load_ps(L.dy), dz = _mm512_load_ps(L.dz);
        const __m512i curV = _mm512_maskz_and_epi32(leafM, node, _mm512_set1_epi32(0xFFFFFF));
        uint32_t cur[LANES];
        _mm512_storeu_si512((void*)cur, curV);
        __m512 P[16];
        for (int k = 0; k < LANES; ++k) P[k] = _mm512_loadu_ps((const float*)(pairs + cur[k]));      /* 48 bytes of pair + the 16 behind it (the array is padded by at least one record) */
        transpose16(P);      /* P: e1.xyz e3.x | e2.xyz e3.y | p0.xyz e3.z | (next record) */
        const __m512 e1x = P[0], e1y = P[1], e1z = P[2], e3x = P[3], e2x = P[4], e2y = P[5], e2z = P[6], e3y = P[7], e3z = P[11];
        __m512 n1x, n1y, n1z, n2x, n2y, n2z, Rx, Ry, Rz;
        CROSS(n1x, n1y, n1z, e1x, e1y, e1z, e2x, e2y, e2z);
        CROSS(n2x, n2y, n2z, e3x, e3y, e3z, e1x, e1y, e1z);
        const __m512 Cx = _mm512_sub_ps(P[8], ox), Cy = _mm512_sub_ps(P[9], oy), Cz = _mm512_sub_ps(P[10], oz);
        CROSS(Rx, Ry, Rz, dx, dy, dz, Cx, Cy, Cz);
        const __m512 det1 = dot3v(n1x, n1y, n1z, dx, dy, dz), det2 = dot3v(n2x, n2y, n2z, dx, dy, dz);
        const __m512i signBit = _mm512_set1_epi32((int)0x80000000u);
        const __m512 s1 = _mm512_castsi512_ps(_mm512_and_si512(_mm512_castps_si512(det1), signBit)), s2 = _mm512_castsi512_ps(_mm512_and_si512(_mm512_castps_si512(det2), signBit));
        const __m512 re1 = dot3v(Rx, Ry, Rz, e1x, e1y, e1z);
        const __m512 U1 = vxor(dot3v(Rx, Ry, Rz, e2x, e2y, e2z), s1), V1 = vxor(re1, s1);
        const __m512 U2 = vxor(neg(re1), s2), V2 = vxor(neg(dot3v(Rx, Ry, Rz, e3x, e3y, e3z)), s2);
        __mmask16 out1 = _mm512_cmplt_epi32_mask(_mm512_or_si512(_mm512_castps_si512(U1), _mm512_castps_si512(V1)), zero);      /* (int)(iU1 | iV1) < 0 */
        __mmask16 out2 = _mm512_cmplt_epi32_mask(_mm512_or_si512(_mm512_castps_si512(U2), _mm512_castps_si512(V2)), zero);
        const __m512 a1 = vabs(det1), a2 = vabs(det2);
        const __m512 W1 = _mm512_sub_ps(_mm512_sub_ps(a1, U1), V1), W2 = _mm512_sub_ps(_mm512_sub_ps(a2, U2), V2);
        const __m512 T1 = vxor(dot3v(n1x, n1y, n1z, Cx, Cy, Cz), s1), T2 = vxor(dot3v(n2x, n2y, n2z, Cx, Cy, Cz), s2);
        const __m512 fzero = _mm512_setzero_ps();
        out1 |= _mm512_cmp_ps_mask(W1, fzero, _CMP_LT_OQ) | _mm512_cmp_ps_mask(T1, _mm512_mul_ps(a1, tNear), _CMP_LE_OQ) | _mm512_cmp_ps_mask(T1, _mm512_mul_ps(a1, tFar), _CMP_GT_OQ);
        out2 |= _mm512_cmp_ps_mask(W2, fzero, _CMP_LT_OQ) | _mm512_cmp_ps_mask(T2, _mm512_mul_ps(a2, tNear), _CMP_LE_OQ) | _mm512_cmp_ps_mask(T2, _mm512_mul_ps(a2, tFar), _CMP_GT_OQ);
        /* second triangle wins: (!out2 && out1) || (!out1 && !out2 && T1 * absDet2 > T2 * absDet1)   (Kernels.h:97) */
        const __mmask16 nearer2 = _mm512_cmp_ps_mask(_mm512_mul_ps(T1, a2), _mm512_mul_ps(T2, a1), _CMP_GT_OQ);
        const __mmask16 second = (__mmask16)~out2 & (out1 | nearer2);
        const __mmask16 hitM = leafM & (__mmask16)~(out1 & out2);
        const __m512 ad = _mm512_mask_blend_ps(second, a1, a2), Ts = _mm512_mask_blend_ps(second, T1, T2);
        const __m512 Us = _mm512_mask_blend_ps(second, U1, U2), Vs = _mm512_mask_blend_ps(second, V1, V2);
        const __m512 rcp = _mm512_div_ps(_mm512_set1_ps(1.0f), ad);
        tFar = _mm512_mask_mov_ps(tFar, hitM, _mm512_mul_ps(Ts, rcp));
        _mm512_store_ps(L.tFar, tFar);
        _mm512_store_ps(L.hu, _mm512_mask_mov_ps(_mm512_load_ps(L.hu), hitM, _mm512_mul_ps(Us, rcp)));
        _mm512_store_ps(L.hv, _mm512_mask_mov_ps(_mm512_load_ps(L.hv), hitM, _mm512_mul_ps(Vs, rcp)));
        __m512i which = _mm512_add_epi32(curV, curV);
        which = _mm512_mask_add_epi32(which, second, which, _mm512_set1_epi32(1));      /* pair * 2 + (second ? 1 : 0) */
        _mm512_store_si512((void*)L.hidx, _mm512_mask_mov_epi32(_mm512_load_si512((const void*)L.hidx), hitM, which));
        const __mmask16 more = leafM & _mm512_cmpgt_epi32_mask(node, _mm512_set1_epi32(0x1FFFFFF));      /* pairs left in this leaf */
        node = _mm512_mask_sub_epi32(node, more, node, _mm512_set1_epi32(0xFFFFFF));                     /* (count - 1, first + 1) */
        popM |= leafM & (__mmask16)~more;
    }

    /* ---- pop, or finish (Kernels.h:207-210) */
    const __mmask16 has = popM & _mm512_cmpgt_epi32_mask(head, zero);
    head = _mm512_mask_sub_epi32(head, has, head, _mm512_set1_epi32(1));
    const __m512i popped = _mm512_mask_i32gather_epi32(zero, has, _mm512_add_epi32(laneBase, head), (const void*)&L.stack[0][0], 4);
    node = _mm512_mask_mov_epi32(node, has, popped);
    const __mmask16 done = popM & (__mmask16)~has;
    node = _mm512_mask_mov_epi32(node, done, zero);      /* finished lanes: idle */
    _mm512_store_si512((void*)L.node, node);
    _mm512_store_si512((void*)L.head, head);
    *idleMaskP = _mm512_cmpeq_epi32_mask(node, zero);
    for (unsigned m = done; m; m &= m - 1) {
        const int k = __builtin_ctz(m);
        store_result(&L, k, remap, env, envW, envH, results + L.ray[k]);
    }
    return 1;
#undef L
}

#ifndef ORC_SIMD512_GROUPS
#define ORC_SIMD512_GROUPS 2      /* groups of sixteen in turn: independent dependent chains */
#endif
void orc_traverse_simd512(const orc_gpu_node* nodes, const orc_pair* pairs, const uint32_t* remap,
                          const float* env, uint32_t envW, uint32_t envH,
                          const orc_ray* rays, orc_result* results, uint32_t start, uint32_t end) {
    static __thread lanes16_t G[ORC_SIMD512_GROUPS];
    memset(G, 0, sizeof(G));
    uint32_t next = start;
    unsigned idle[ORC_SIMD512_GROUPS];
    int alive[ORC_SIMD512_GROUPS], any = 1;
    for (int g = 0; g < ORC_SIMD512_GROUPS; ++g) { idle[g] = 0xFFFF; alive[g] = 1; }
    while (any) {
        any = 0;
        for (int g = 0; g < ORC_SIMD512_GROUPS; ++g)
            if (alive[g]) any |= alive[g] = simd_step(&G[g], &idle[g], &next, end, nodes, pairs, remap, env, envW, envH, rays, results);
    }
}
