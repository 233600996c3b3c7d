/*
 * split_sim.c — CPU prototype of "work splitting in the launch tail" (analysis tool, test infrastructure).
 *
 * The GPU kernel's fixed cost per launch is the drain: after the ray cursor runs dry every wave still holds a few long
 * rays, and a ray is a dependent chain of node visits.  This prototype lets idle lanes of a draining wave take over the
 * OLDEST stack entry (the bottom of the donor's stack: the subtree the reference would visit last) of a running ray and
 * traverse it concurrently, and reconstructs the reference's sequential result exactly:
 *
 *   - fragments of one ray are totally ordered (the donor keeps the front of its work list, the helper gets the back:
 *     key/width interval halving), the first fragment runs with the true tFar, later ones with a LOOSER tFar (the donor's
 *     value at donation time), so they visit a superset of the nodes and test a superset of the pairs;
 *   - when all fragments are done they are folded in order; a fragment's loose run equals its true run (tFar = result of
 *     the earlier fragments) unless that value falls into a "danger interval" recorded at each pair the fragment accepted
 *     ([t(1-eps), max(leaf entry distance, t(1+eps))]: the pair test's scaled compare, Kernels.h:88-89, or the box-entry
 *     sentinel, Kernels.h:131-134, could have decided differently);
 *   - if it does, the ray is simply traversed again, unsplit.
 *
 * The program checks every ray against the plain sequential traversal bit for bit and reports how much shorter the
 * drain gets (scheduling iterations after the cursor is exhausted) and how often the fallback fires.
 *
 * Build: gcc -O2 -ffp-contract=off -mavx2 -mfma -o split_sim split_sim.c -lm -lpthread   (includes racc_oracle.c)
 */
#include "racc_oracle.c"

#include <float.h>
#include <stdio.h>

#define WAVE 64
#define MAXSTACK 128
#define K_EMPTY 0u
#define K_DONE 1u
#define K_WAIT 2u
#define K_LEAFBASE 0x1000000u

typedef struct {
    ray_state ray;
    float invDir[3], OoD[3];
    hit_state hit;
    uint32_t node, sp, base;
    uint32_t stack[MAXSTACK];
    float stackT0[MAXSTACK];
    uint32_t rayIdx;
    /* split state */
    int helper, root, noSplit, replay, flag, nAcc;
    uint32_t key, width;
    float curT0, lastLo, lastHi, oldLo, oldHi;
    uint32_t age;           /* node visits + pair tests this ray (root) or fragment (helper) has done */
    int pending;            /* root only: fragments (including the root's own) still running */
    int nFrag;              /* root only: fragments ever created besides the root's own */
} lane_t;

typedef struct { int leafMin, refillMin, tailActive, chunk, split, minStack, minAge, stash, both, share; } policy_t;

typedef struct {
    unsigned long long mainIters, drainIters, splitSteps, donations, fallbacks, foldedRays, rays, drainVisits, mainVisits, mismatches, monoViol;
    unsigned long long drainMax, drainRefills, drainBodies;
} stats_t;

static const orc_gpu_node* g_nodes;
static const orc_pair* g_pairs;

static void lane_load(lane_t* L, const orc_ray* in, uint32_t idx) {
    for (int k = 0; k < 3; ++k) { L->ray.o[k] = in->origin[k]; L->ray.d[k] = in->dir[k]; }
    L->ray.tNear = in->minT; L->ray.tFar = in->maxT;
    for (int k = 0; k < 3; ++k) if (fabsf(L->ray.d[k]) < 1e-10f) L->ray.d[k] = copysignf(1e-10f, L->ray.d[k]);
    for (int k = 0; k < 3; ++k) { L->invDir[k] = 1.0f / L->ray.d[k]; L->OoD[k] = -L->ray.o[k] * L->invDir[k]; }
    L->hit.index = -1; L->hit.t = L->ray.tFar; L->hit.u = L->hit.v = 0;
    L->node = 0x80000000u; L->sp = 0; L->base = 0; L->rayIdx = idx;
    L->helper = 0; L->root = -1; L->noSplit = 0; L->replay = 0; L->flag = 0; L->nAcc = 0;
    L->key = 0; L->width = 0x80000000u; L->curT0 = -INFINITY; L->pending = 1; L->nFrag = 0; L->age = 0;
}

/* pair test that also reports whether BOTH triangles of the pair passed */
static float pair_ex(int32_t index, const ray_state* ray, hit_state* hit, int* both) {
    hit_state h1 = *hit;
    const float t = pair_intersect(g_pairs, index, ray, &h1);
    *both = 0;
    if (h1.index != hit->index || h1.t != hit->t || t != ray->tFar) {
        /* accepted: probe each triangle alone by re-testing with the range closed just below/above is not possible
           cheaply here; recompute the two pass flags the way the kernel would */
        const orc_pair* p = g_pairs + index;
        const float e1[3] = { p->e1[0], p->e1[1], p->e1[2] }, e2[3] = { p->e2[0], p->e2[1], p->e2[2] }, e3[3] = { p->e3x, p->e3y, p->e3z };
        float n1[3], n2[3], C[3], R[3];
        mad_cross(n1, e1, e2); mad_cross(n2, e3, e1);
        for (int k = 0; k < 3; ++k) C[k] = p->p0[k] - ray->o[k];
        mad_cross(R, ray->d, C);
        const float det1 = dot3(n1, ray->d), det2 = dot3(n2, ray->d);
        const uint32_t s1 = f2u(det1) & 0x80000000u, s2 = f2u(det2) & 0x80000000u;
        const uint32_t iU1 = f2u(dot3(R, e2)) ^ s1, iV1 = f2u(dot3(R, e1)) ^ s1, iU2 = f2u(-dot3(R, e1)) ^ s2, iV2 = f2u(-dot3(R, e3)) ^ s2;
        int o1 = (int32_t)(iU1 | iV1) < 0, o2 = (int32_t)(iU2 | iV2) < 0;
        const float a1 = fabsf(det1), a2 = fabsf(det2);
        const float T1 = u2f(f2u(dot3(n1, C)) ^ s1), T2 = u2f(f2u(dot3(n2, C)) ^ s2);
        o1 = o1 || (a1 - u2f(iU1) - u2f(iV1) < 0.0f || T1 <= a1 * ray->tNear || T1 > a1 * ray->tFar);
        o2 = o2 || (a2 - u2f(iU2) - u2f(iV2) < 0.0f || T2 <= a2 * ray->tNear || T2 > a2 * ray->tFar);
        *both = !o1 && !o2;
    }
    *hit = h1;
    return t;
}

static void lane_pop(lane_t* L) {
    if (L->sp == L->base) { L->node = K_DONE; return; }
    --L->sp;
    L->node = L->stack[L->sp]; L->curT0 = L->stackT0[L->sp];
    /* a helper re-runs the parent's slab test to learn a popped LEAF's entry distance (models the kernel's replay entry) */
    L->replay = (L->helper && (int32_t)L->node >= (int32_t)K_LEAFBASE) ? 1 : 0;
}

static int is_inner(const lane_t* L) { return (int32_t)L->node < 0 || ((int32_t)L->node >= (int32_t)K_LEAFBASE && L->replay); }
static int is_leaf(const lane_t* L) { return (int32_t)L->node >= (int32_t)K_LEAFBASE && !L->replay; }

static void lane_inner(lane_t* L, stats_t* st, int drain) {
    ++L->age;
    if (L->replay) { L->replay = 0; return; }
    const orc_gpu_node* n = g_nodes + (L->node & 0x7FFFFFFFu);
    const float tRay = L->ray.tFar;
    const float tFirst = aabb_intersect(n->leftMin, n->leftMax, &L->ray, L->invDir, L->OoD);
    const float tLast = aabb_intersect(n->rightMin, n->rightMax, &L->ray, L->invDir, L->OoD);
    if (drain) st->drainVisits++; else st->mainVisits++;
    if ((tRay - tFirst) + (tRay - tLast) != 0.0f) {
        const int sgn = tLast < tFirst;
        if (omax(tFirst, tLast) != tRay) {
            L->stack[L->sp] = sgn ? n->first : n->last;
            L->stackT0[L->sp] = sgn ? tFirst : tLast;
            if (L->helper && L->stackT0[L->sp] < L->curT0) st->monoViol++;
            ++L->sp;
        }
        L->node = sgn ? n->last : n->first;
        const float t0 = sgn ? tLast : tFirst;
        if (L->helper && t0 < L->curT0) st->monoViol++;
        L->curT0 = t0;
    } else {
        lane_pop(L);
    }
}

static void lane_leaf(lane_t* L) {
    const uint32_t cur = L->node & 0xFFFFFFu, cnt = L->node >> 24;
    int both = 0;
    ++L->age;
    const float before = L->ray.tFar;
    const int32_t idxBefore = L->hit.index;
    L->ray.tFar = pair_ex((int32_t)cur, &L->ray, &L->hit, &both);
    if (L->helper && (L->hit.index != idxBefore || L->ray.tFar != before)) {
        const float eps = 1.0f / 1048576.0f;
        const float t = L->ray.tFar;
        if (L->nAcc >= 1) {
            if (L->nAcc == 1) { L->oldLo = L->lastLo; L->oldHi = L->lastHi; }
            else { L->oldLo = omin(L->oldLo, L->lastLo); L->oldHi = omax(L->oldHi, L->lastHi); }
        }
        L->lastLo = t * (1.0f - eps) - FLT_MIN;
        L->lastHi = both ? INFINITY : omax(L->curT0, t * (1.0f + eps) + FLT_MIN);
        L->nAcc++;
    }
    if (cnt > 1) L->node = ((cnt - 1) << 24) | (cur + 1);
    else lane_pop(L);
}

/* plain sequential traversal with the same primitives: the reference result */
static void sequential(const orc_ray* in, hit_state* out) {
    static __thread lane_t L;
    stats_t dummy; memset(&dummy, 0, sizeof dummy);
    lane_load(&L, in, 0);
    while (L.node != K_DONE) { if (is_leaf(&L)) lane_leaf(&L); else lane_inner(&L, &dummy, 0); }
    *out = L.hit;
}

static void check(const orc_ray* rays, uint32_t idx, const hit_state* got, stats_t* st) {
    hit_state ref;
    sequential(rays + idx, &ref);
    if (ref.index != got->index || (ref.index >= 0 && (f2u(ref.t) != f2u(got->t) || f2u(ref.u) != f2u(got->u) || f2u(ref.v) != f2u(got->v)))) {
        if (st->mismatches < 5) fprintf(stderr, "MISMATCH ray %u: ref (%d %.9g) got (%d %.9g)\n", idx, ref.index, ref.t, got->index, got->t);
        st->mismatches++;
    }
}

/* finished fragments parked outside the lanes (stash mode) */
typedef struct { int root; uint32_t key; int flag, nAcc; float lastLo, lastHi, oldLo, oldHi, tFar; hit_state hit; int live; } frag_t;
#define MAXFRAG 4096
static __thread frag_t g_frag[MAXFRAG];
static __thread int g_nfrag;

/* fold the fragments of root R in key order; returns 0 when the ray must be traversed again unsplit */
static int fold(lane_t* lanes, int R, hit_state* out) {
    hit_state cur = lanes[R].hit;
    float c = lanes[R].ray.tFar;
    uint32_t lastKey = 0;
    for (;;) {
        frag_t cand; int have = 0;
        for (int l = 0; l < WAVE; ++l)
            if (l != R && lanes[l].node == K_WAIT && lanes[l].helper && lanes[l].root == R && lanes[l].key > lastKey && (!have || lanes[l].key < cand.key)) {
                const lane_t* f = &lanes[l];
                cand = (frag_t){ R, f->key, f->flag, f->nAcc, f->lastLo, f->lastHi, f->oldLo, f->oldHi, f->ray.tFar, f->hit, 1 }; have = 1;
            }
        for (int i = 0; i < g_nfrag; ++i)
            if (g_frag[i].live && g_frag[i].root == R && g_frag[i].key > lastKey && (!have || g_frag[i].key < cand.key)) { cand = g_frag[i]; have = 1; }
        if (!have) break;
        const frag_t* f = &cand;
        lastKey = f->key;
        if (f->flag) return 0;
        if (f->nAcc == 0) continue;
        if (c >= f->lastLo && c <= f->lastHi) return 0;
        if (f->nAcc > 1 && c >= f->oldLo && c <= f->oldHi) return 0;
        if (c > f->lastHi) { cur = f->hit; c = f->hit.t; }
        else if (f->nAcc > 1 && c > f->oldHi) return 0;
    }
    *out = cur;
    return 1;
}

static unsigned long long sim_wave(const orc_ray* rays, uint32_t count, uint32_t firstChunk, uint32_t chunkStride, const policy_t* P, stats_t* st, int verify) {
    lane_t* lanes = (lane_t*)calloc(WAVE, sizeof(lane_t));
    g_nfrag = 0;
    uint32_t chunk = firstChunk, wBeg = 0, wEnd = 0;
    int exhausted = 0;
    unsigned long long drain = 0;
    const unsigned long long cost0 = st->drainIters;
    for (int i = 0; i < WAVE; ++i) lanes[i].node = K_EMPTY;
    for (;;) {
        const int draining = exhausted && wBeg == wEnd;
        /* ---- completion ---- */
        if (draining && P->split) {
            for (int l = 0; l < WAVE; ++l) {
                lane_t* L = &lanes[l];
                if (L->node != K_DONE) continue;
                const int R = L->helper ? L->root : l;
                if (!L->helper && L->nFrag == 0) continue;      /* unsplit ray: ordinary epilogue below */
                L->node = K_WAIT;
                if (P->stash && L->helper) {
                    if (g_nfrag < MAXFRAG) g_frag[g_nfrag++] = (frag_t){ R, L->key, L->flag, L->nAcc, L->lastLo, L->lastHi, L->oldLo, L->oldHi, L->ray.tFar, L->hit, 1 };
                    else L->flag = 1;
                    if (g_nfrag < MAXFRAG || 1) { L->node = K_EMPTY; L->helper = 0; }
                }
                if (--lanes[R].pending == 0) {
                    hit_state res;
                    st->foldedRays++;
                    const int ok = fold(lanes, R, &res);
                    for (int h = 0; h < WAVE; ++h) if (h != R && lanes[h].node == K_WAIT && lanes[h].helper && lanes[h].root == R) { lanes[h].node = K_EMPTY; lanes[h].helper = 0; }
                    for (int i = 0; i < g_nfrag; ++i) if (g_frag[i].root == R) g_frag[i].live = 0;
                    if (ok) {
                        if (verify) check(rays, lanes[R].rayIdx, &res, st);
                        lanes[R].node = K_EMPTY;
                    } else {
                        st->fallbacks++;
                        lane_load(&lanes[R], rays + lanes[R].rayIdx, lanes[R].rayIdx);
                        lanes[R].noSplit = 1;
                    }
                }
            }
        }
        int nInner = 0, nLeaf = 0, nDone = 0, nEmpty = 0;
        for (int l = 0; l < WAVE; ++l) {
            const lane_t* L = &lanes[l];
            if (L->node == K_DONE) ++nDone; else if (L->node == K_EMPTY) ++nEmpty;
            else if (L->node == K_WAIT) {} else if (is_inner(L)) ++nInner; else ++nLeaf;
        }
        const int noWork = (nInner == 0 && nLeaf == 0);
        int refill = noWork;
        if (!noWork) refill = exhausted ? (nDone >= P->refillMin) : ((nDone + nEmpty) >= P->refillMin);
        if (draining && P->split && nDone) refill = 1;      /* drain mode: finished rays are written out at once so the lane can help */
        if (refill) {
            if (draining) { ++drain; st->drainRefills++; st->drainIters += 150; } else st->mainIters++;
            for (int i = 0; i < WAVE; ++i) if (lanes[i].node == K_DONE) {
                if (verify) check(rays, lanes[i].rayIdx, &lanes[i].hit, st);
                lanes[i].node = K_EMPTY;
            }
            for (int i = 0; i < WAVE; ++i) {
                if (lanes[i].node != K_EMPTY) continue;
                if (wBeg == wEnd && !exhausted) {
                    const uint64_t b = (uint64_t)chunk * P->chunk;
                    chunk += chunkStride;
                    if (b >= count) { exhausted = 1; } else { wBeg = (uint32_t)b; wEnd = (uint32_t)(b + P->chunk < count ? b + P->chunk : count); }
                }
                if (wBeg == wEnd) break;
                lane_load(&lanes[i], rays + wBeg, wBeg); ++wBeg; st->rays++;
            }
            int any = 0;
            for (int i = 0; i < WAVE; ++i) any |= (lanes[i].node != K_EMPTY);
            if (exhausted && wBeg == wEnd && !any) break;
            continue;
        }
        /* ---- tFar sharing: a helper may prune with the current tFar of its ray's FIRST fragment (always earlier in order) ---- */
        if (draining && P->split && P->share)
            for (int l = 0; l < WAVE; ++l) {
                lane_t* H = &lanes[l];
                if (H->node > K_WAIT && H->helper && lanes[H->root].ray.tFar < H->ray.tFar) H->ray.tFar = lanes[H->root].ray.tFar;
            }
        /* ---- split step ---- */
        if (draining && P->split && nEmpty) {
            int did = 0;
            int idle = 0;
            char used[WAVE]; memset(used, 0, sizeof used);
            for (;;) {
                while (idle < WAVE && lanes[idle].node != K_EMPTY) ++idle;
                if (idle == WAVE) break;
                int d = -1;
                for (int k = 0; k < WAVE; ++k) {
                    lane_t* D = &lanes[k];
                    if (used[k] || D->node <= K_WAIT || D->noSplit || D->width < 2u || (int)(D->sp - D->base) < P->minStack || !isfinite(D->ray.tFar) || (int)D->age < P->minAge) continue;
                    if (d < 0 || D->age > lanes[d].age) d = k;
                }
                if (d < 0) break;
                used[d] = 1;
                lane_t* D = &lanes[d];
                lane_t* H = &lanes[idle];
                const int R = D->helper ? D->root : d;
                memcpy(&H->ray, &D->ray, sizeof H->ray); memcpy(H->invDir, D->invDir, sizeof H->invDir); memcpy(H->OoD, D->OoD, sizeof H->OoD);
                H->hit.index = -1; H->hit.t = D->ray.tFar; H->hit.u = H->hit.v = 0;
                H->node = D->stack[D->base]; H->curT0 = -INFINITY; ++D->base;
                H->sp = H->base = 0; H->rayIdx = D->rayIdx;
                H->helper = 1; H->root = R; H->noSplit = 0; H->replay = 0; H->flag = 0; H->nAcc = 0; H->age = D->age / 2;
                D->width >>= 1; H->width = D->width; H->key = D->key + D->width;
                lanes[R].pending++; lanes[R].nFrag++;
                st->donations++; did = 1;
            }
            if (did) { st->splitSteps++; ++drain; }
        }
        const int nActive = nInner + nLeaf;
        const int thin = nActive <= P->tailActive;
        int doLeaf = nLeaf >= P->leafMin || nInner == 0 || (nLeaf * 4 >= nActive);
        int doInner = nInner != 0 && (!doLeaf || thin);
        if (draining && P->both == 1) { doLeaf = nLeaf != 0; doInner = 1; }
        if (draining && P->both >= 2 && thin) {
            /* critical-ray-first: the ray that has already done the most steps decides which body runs */
            int o = -1;
            for (int l = 0; l < WAVE; ++l) if (lanes[l].node > K_WAIT && (o < 0 || lanes[l].age > lanes[o].age)) o = l;
            const int oLeaf = is_leaf(&lanes[o]);
            if (P->both == 2) { doLeaf = oLeaf; doInner = !oLeaf; }
            else { doLeaf = oLeaf || (nLeaf * 2 >= nActive); doInner = nInner != 0 && (!oLeaf || nInner * 2 >= nActive); }
        }
        if (draining) st->drainIters += 40 + (doLeaf ? 140 : 0) + (doInner ? 65 : 0);
        if (draining) { drain += (doLeaf ? 1 : 0) + (doInner ? 1 : 0); st->drainBodies += (doLeaf ? 1 : 0) + (doInner ? 1 : 0); } else st->mainIters += (doLeaf ? 1 : 0) + (doInner ? 1 : 0);
        /* snapshot the classes first: a lane does one step per body */
        int wasLeaf[WAVE], wasInner[WAVE];
        for (int l = 0; l < WAVE; ++l) { wasLeaf[l] = lanes[l].node > K_WAIT && is_leaf(&lanes[l]); wasInner[l] = lanes[l].node > K_WAIT && is_inner(&lanes[l]); }
        if (doLeaf) for (int l = 0; l < WAVE; ++l) if (wasLeaf[l]) lane_leaf(&lanes[l]);
        if (doInner) for (int l = 0; l < WAVE; ++l) if (lanes[l].node > K_WAIT && is_inner(&lanes[l]) && (wasInner[l] || doLeaf)) lane_inner(&lanes[l], st, draining);
    }
    free(lanes);
    { const unsigned long long c = st->drainIters - cost0; if (c > st->drainMax) st->drainMax = c; }
    return drain;
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: split_sim nodes.bin pairs.bin rays.bin nwaves [split minStack minAge stash both]\n"); return 2; }
    FILE* f; size_t n;
    f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); n = ftell(f); rewind(f); orc_gpu_node* nodes = malloc(n); if (fread(nodes, 1, n, f) != n) return 3; fclose(f);
    f = fopen(argv[2], "rb"); fseek(f, 0, SEEK_END); n = ftell(f); rewind(f); orc_pair* pairs = malloc(n); if (fread(pairs, 1, n, f) != n) return 3; fclose(f);
    f = fopen(argv[3], "rb"); fseek(f, 0, SEEK_END); n = ftell(f); rewind(f); orc_ray* rays = malloc(n); if (fread(rays, 1, n, f) != n) return 3; fclose(f);
    const uint32_t count = (uint32_t)(n / 32);
    g_nodes = nodes; g_pairs = pairs;
    const int nw = atoi(argv[4]);
    policy_t P = { 12, 32, 16, 64, 1, 1, 0, 0, 0, 0 };
    if (argc > 5) P.split = atoi(argv[5]);
    if (argc > 6) P.minStack = atoi(argv[6]);
    if (argc > 7) P.minAge = atoi(argv[7]);
    if (argc > 8) P.stash = atoi(argv[8]);
    if (argc > 9) P.both = atoi(argv[9]);
    if (argc > 10) P.share = atoi(argv[10]);
    const uint32_t totalWaves = 5120u;
    const int verify = 1;
    stats_t st; memset(&st, 0, sizeof(st));
    unsigned long long sum = 0, mx = 0;
    for (int w = 0; w < nw; ++w) {
        const unsigned long long d = sim_wave(rays, count, (uint32_t)w * (totalWaves / nw), totalWaves, &P, &st, verify);
        sum += d; if (d > mx) mx = d;
    }
    printf("{\"split\": %d, \"minStack\": %d, \"minAge\": %d, \"stash\": %d, \"both\": %d, \"waves\": %d, \"rays\": %llu, \"main_iters_per_wave\": %.1f, \"drain_iters_mean\": %.1f, \"drain_iters_max\": %llu, "
           "\"donations\": %llu, \"split_steps\": %llu, \"folded_rays\": %llu, \"fallbacks\": %llu, \"drain_visits\": %llu, \"main_visits\": %llu, \"mono_violations\": %llu, \"mismatches\": %llu, \"drain_refills_per_wave\": %.1f, \"drain_bodies_per_wave\": %.1f, \"split_steps_per_wave\": %.1f, \"drain_cost_mean\": %.0f, \"drain_cost_max\": %llu}\n",
           P.split, P.minStack, P.minAge, P.stash, P.both, nw, st.rays, (double)st.mainIters / nw, (double)sum / nw, mx, st.donations, st.splitSteps, st.foldedRays, st.fallbacks,
           st.drainVisits, st.mainVisits, st.monoViol, st.mismatches, (double)st.drainRefills / nw, (double)st.drainBodies / nw, (double)st.splitSteps / nw, (double)st.drainIters / nw, st.drainMax);
    return st.mismatches ? 1 : 0;
}
