/*
 * wave_sim.c — CPU model of the GPU kernel's wave-level scheduling (analysis tool, test infrastructure).
 *
 * Replays real rays through the oracle's own slab / pair tests one STEP at a time under the scheduling policy of
 * traverseKernelV2 (vote between an inner step and a leaf step, refill thresholds, thin-wave rule) and counts
 * scheduling iterations and live lanes.  Used to evaluate policy changes (speculative traversal, two rays per lane,
 * regrouping periods) before spending GPU time on them; validated against the GPU's own counters
 * (kernel_variant 12: 77 inner + 22 leaf + 2 refill iterations per 64 rays in steady state).
 *
 * Build: gcc -O2 -ffp-contract=off -mavx2 -mfma -o wave_sim wave_sim.c -lm -lpthread   (includes racc_oracle.c)
 */
#include "racc_oracle.c"

#include <stdio.h>

#define WAVE 64
#define MAXSTACK 128
#define K_EMPTY 0u
#define K_DONE 1u
#define K_LEAFBASE 0x1000000u

typedef struct {
    ray_state ray;
    float invDir[3], OoD[3];
    hit_state hit;
    uint32_t node, sp;
    uint32_t stack[MAXSTACK];
    float stackT0[MAXSTACK];
    uint8_t stackSpec[MAXSTACK];
    /* speculative traversal state */
    uint32_t pending;       /* postponed leaf ref (0 = none) */
    uint32_t specBase;      /* stack height when the leaf was postponed */
    float curT0;            /* entry distance of the current node if it was reached speculatively */
    int curSpec;
    int blocked;            /* holds a postponed leaf and needs a push it may not make: waits for the leaf step */
    uint32_t rayIdx;
} lane_t;

typedef struct {
    int leafMin, refillMin, tailActive, chunk;
    int speculate;          /* 0 = exact V2; 1 = postpone one leaf, continue without pushes; 2 = continue with pushes */
    int raysPerLane;        /* 1 or 2 */
} policy_t;

typedef struct {
    unsigned long long innerIters, innerLanes, leafIters, leafLanes, refillIters, rays, extraVisits, nodeVisits;
} stats_t;

static const orc_gpu_node* g_nodes;
static const orc_pair* g_pairs;

static void lane_load(lane_t* L, const orc_ray* in, uint32_t idx) {
    for (int k = 0; k < 3; ++k) { L->ray.o[k] = in->origin[k]; L->ray.d[k] = in->dir[k]; }
    L->ray.tNear = in->minT; L->ray.tFar = in->maxT;
    for (int k = 0; k < 3; ++k) if (fabsf(L->ray.d[k]) < 1e-10f) L->ray.d[k] = copysignf(1e-10f, L->ray.d[k]);
    for (int k = 0; k < 3; ++k) { L->invDir[k] = 1.0f / L->ray.d[k]; L->OoD[k] = -L->ray.o[k] * L->invDir[k]; }
    L->hit.index = -1; L->hit.t = L->ray.tFar; L->hit.u = L->hit.v = 0;
    L->node = 0x80000000u; L->sp = 0; L->pending = 0; L->specBase = 0; L->curSpec = 0; L->curT0 = 0; L->blocked = 0; L->rayIdx = idx;
}

static void lane_pop(lane_t* L) {
    /* speculative entries are validated lazily here: equivalent to filtering them when the postponed leaf commits,
       because tFar cannot change between that commit and this pop without another commit doing the same filtering */
    while (L->sp) {
        --L->sp;
        if (L->stackSpec[L->sp] && !(L->stackT0[L->sp] < L->ray.tFar)) continue;
        L->node = L->stack[L->sp]; L->curSpec = L->stackSpec[L->sp]; L->curT0 = L->stackT0[L->sp];
        return;
    }
    L->node = K_DONE;
}

/* one inner step; `spec` = the lane currently holds a postponed leaf */
static int lane_inner(lane_t* L, int spec, int allowPush, stats_t* st) {
    const orc_gpu_node* n = g_nodes + (L->node & 0x7FFFFFFFu);
    const float tRay = L->ray.tFar;
    const float tFirst = aabb_intersect(n->leftMin, n->leftMax, &L->ray, L->invDir, L->OoD);
    const float tLast = aabb_intersect(n->rightMin, n->rightMax, &L->ray, L->invDir, L->OoD);
    st->nodeVisits++;
    if ((tRay - tFirst) + (tRay - tLast) != 0.0f) {
        const int sgn = tLast < tFirst;
        const int both = omax(tFirst, tLast) != tRay;
        if (both) {
            if (spec && !allowPush) { L->blocked = 1; return 0; }   /* cannot proceed without a push: wait for the commit */
            L->stack[L->sp] = sgn ? n->first : n->last;
            L->stackT0[L->sp] = sgn ? tFirst : tLast;
            L->stackSpec[L->sp] = (uint8_t)spec;
            ++L->sp;
        }
        L->node = sgn ? n->last : n->first;
        L->curT0 = sgn ? tLast : tFirst;
        L->curSpec = spec;
    } else {
        lane_pop(L);
    }
    return 1;
}

static void lane_leaf_one(lane_t* L, uint32_t* leafRef) {
    const uint32_t cur = *leafRef & 0xFFFFFFu, cnt = *leafRef >> 24;
    L->ray.tFar = pair_intersect(g_pairs, (int32_t)cur, &L->ray, &L->hit);
    *leafRef = cnt > 1 ? (((cnt - 1) << 24) | (cur + 1)) : 0u;
}

static int is_inner(uint32_t n) { return (int32_t)n < 0; }
static int is_leaf(uint32_t n) { return (int32_t)n >= (int32_t)K_LEAFBASE; }

/* Simulates one wave over the chunk sequence first, first+stride, ... */
static void sim_wave(const orc_ray* rays, uint32_t count, uint32_t firstChunk, uint32_t chunkStride, const policy_t* P, stats_t* st) {
    lane_t* lanes = (lane_t*)calloc(WAVE * P->raysPerLane, sizeof(lane_t));
    const int NL = WAVE * P->raysPerLane;      /* ray slots; with raysPerLane = 2 a lane is live if either slot matches */
    uint32_t chunk = firstChunk, wBeg = 0, wEnd = 0;
    int exhausted = 0;
    for (int i = 0; i < NL; ++i) lanes[i].node = K_EMPTY;
    for (;;) {
        int nDone = 0, nEmpty = 0;
        for (int l = 0; l < WAVE; ++l) {
            int dn = 0, em = 0;
            for (int s = 0; s < P->raysPerLane; ++s) {
                const lane_t* L = &lanes[l * P->raysPerLane + s];
                if (L->node == K_DONE && !L->pending) dn = 1;
                if (L->node == K_EMPTY) em = 1;
            }
            nDone += dn; nEmpty += em;
        }
        /* lanes that can make progress with an inner step: inner node and (no pending leaf or speculation allowed) */
        int canInner = 0, mustLeaf = 0;
        for (int l = 0; l < WAVE; ++l) {
            int ci = 0, ml = 0;
            for (int s = 0; s < P->raysPerLane; ++s) {
                lane_t* L = &lanes[l * P->raysPerLane + s];
                if (is_inner(L->node) && (!L->pending || (P->speculate && !L->blocked))) ci = 1;
                if (is_leaf(L->node) || L->pending) ml = 1;
            }
            canInner += ci; mustLeaf += ml;
        }
        const int noWork = (canInner == 0 && mustLeaf == 0);
        int refill = noWork;
        if (!noWork) refill = exhausted ? (nDone >= P->refillMin) : ((nDone + nEmpty) >= P->refillMin);
        if (refill) {
            st->refillIters++;
            for (int i = 0; i < NL; ++i) if (lanes[i].node == K_DONE && !lanes[i].pending) lanes[i].node = K_EMPTY;
            for (int i = 0; i < NL; ++i) {
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
            for (int i = 0; i < NL; ++i) any |= (lanes[i].node != K_EMPTY) || lanes[i].pending;
            if (exhausted && wBeg == wEnd && !any) break;
            continue;
        }
        const int nActive = canInner + mustLeaf;
        const int thin = nActive <= P->tailActive;
        const int doLeaf = mustLeaf >= P->leafMin || canInner == 0 || (thin && mustLeaf * 4 >= nActive);
        const int doInner = canInner != 0 && (!doLeaf || thin);
        if (doLeaf) {
            st->leafIters++; st->leafLanes += mustLeaf;
            for (int l = 0; l < WAVE; ++l) {
                for (int s = 0; s < P->raysPerLane; ++s) {       /* one pair per LANE per step: first slot that has leaf work */
                    lane_t* L = &lanes[l * P->raysPerLane + s];
                    if (L->pending) {
                        lane_leaf_one(L, &L->pending);
                        if (!L->pending) {   /* commit: validate what was reached speculatively */
                            L->blocked = 0;
                            if (L->curSpec && (is_inner(L->node) || is_leaf(L->node)) && !(L->curT0 < L->ray.tFar)) lane_pop(L);
                            L->curSpec = 0;
                            for (uint32_t k = L->specBase; k < L->sp; ++k) if (L->stackSpec[k] && L->stackT0[k] < L->ray.tFar) L->stackSpec[k] = 0;
                        }
                        break;
                    } else if (is_leaf(L->node)) {
                        uint32_t ref = L->node;
                        lane_leaf_one(L, &ref);
                        if (ref) L->node = ref; else lane_pop(L);
                        break;
                    }
                }
            }
        }
        if (doInner) {
            st->innerIters++;
            int live = 0;
            for (int l = 0; l < WAVE; ++l) {
                for (int s = 0; s < P->raysPerLane; ++s) {
                    lane_t* L = &lanes[l * P->raysPerLane + s];
                    if (!is_inner(L->node)) {
                        /* a lane that sits at a leaf with nothing postponed may postpone it and keep going */
                        if (P->speculate && is_leaf(L->node) && !L->pending && L->sp > 0 && !doLeaf) {
                            L->pending = L->node; L->specBase = L->sp;
                            lane_pop(L);   /* pre-existing entry: the reference pops it next as well */
                            L->curSpec = 0;
                            if (!is_inner(L->node)) continue;
                        } else continue;
                    }
                    if (L->pending && (!P->speculate || L->blocked)) continue;
                    if (lane_inner(L, L->pending != 0, P->speculate == 2, st)) { ++live; break; }
                }
            }
            st->innerLanes += live;
        }
    }
    free(lanes);
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: wave_sim nodes.bin pairs.bin rays.bin nwaves [leafMin refillMin tail speculate raysPerLane totalWaves]\n"); return 2; }
    FILE* f; size_t n;
    f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); n = ftell(f); rewind(f); orc_gpu_node* nodes = malloc(n); if (fread(nodes, 1, n, f) != n) return 3; fclose(f);
    f = fopen(argv[2], "rb"); fseek(f, 0, SEEK_END); n = ftell(f); rewind(f); orc_pair* pairs = malloc(n); if (fread(pairs, 1, n, f) != n) return 3; fclose(f);
    f = fopen(argv[3], "rb"); fseek(f, 0, SEEK_END); n = ftell(f); rewind(f); orc_ray* rays = malloc(n); if (fread(rays, 1, n, f) != n) return 3; fclose(f);
    const uint32_t count = (uint32_t)(n / 32);
    g_nodes = nodes; g_pairs = pairs;
    const int nw = atoi(argv[4]);
    policy_t P = { 12, 32, 16, 64, 0, 1 };
    if (argc > 5) P.leafMin = atoi(argv[5]);
    if (argc > 6) P.refillMin = atoi(argv[6]);
    if (argc > 7) P.tailActive = atoi(argv[7]);
    if (argc > 8) P.speculate = atoi(argv[8]);
    if (argc > 9) P.raysPerLane = atoi(argv[9]);
    const uint32_t totalWaves = argc > 10 ? (uint32_t)atoi(argv[10]) : 5120u;
    stats_t st; memset(&st, 0, sizeof(st));
    for (int w = 0; w < nw; ++w) sim_wave(rays, count, (uint32_t)w * (totalWaves / nw), totalWaves, &P, &st);
    const double per64 = 64.0 / (double)st.rays;
    printf("{\"rays\": %llu, \"inner_per64\": %.2f, \"leaf_per64\": %.2f, \"refill_per64\": %.2f, \"inner_util\": %.3f, \"leaf_util\": %.3f, \"visits_per_ray\": %.2f, "
           "\"cost_per_ray\": %.1f, \"policy\": [%d,%d,%d,%d,%d]}\n",
           st.rays, st.innerIters * per64, st.leafIters * per64, st.refillIters * per64,
           (double)st.innerLanes / (st.innerIters ? st.innerIters : 1) / 64.0, (double)st.leafLanes / (st.leafIters ? st.leafIters : 1) / 64.0,
           (double)st.nodeVisits / st.rays,
           (st.innerIters * 65.0 + st.leafIters * 140.0 + st.refillIters * 150.0) / (double)st.rays,
           P.leafMin, P.refillMin, P.tailActive, P.speculate, P.raysPerLane);
    return 0;
}
