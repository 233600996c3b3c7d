/*
 * wide_sim.c — CPU model of the engine's 4-wide traversal (analysis tool + checker, test infrastructure).
 *
 * The engine's device format may differ from the reference's as long as the results do not: at upload the
 * reference-format BVH2 (Scene.cpp:73-78) can be collapsed into 4-wide nodes, one 128-byte vector-L1 line each.  This
 * program restates that collapse and the 4-wide traversal on the CPU, runs it next to the reference-order BVH2 traversal
 * (racc_oracle.c, Kernels.h:139-242) on the same buffers and reports
 *   - node visits / pair tests / stack depth per ray for both,
 *   - how many results differ, split into exact-distance ties (same t, other primitive: either answer is a closest hit,
 *     SURVEY.md §8c's arbiter rule) and real differences (must be 0).
 * The per-box test is the reference's (same fma, same "entry distance == tFar counts as missed" rule, Kernels.h:117-135,
 * 192-194); only the ORDER in which the hit children are visited is the wide node's own: nearest entry distance first.
 *
 * Build: gcc -O2 -ffp-contract=off -mavx2 -mfma -o wide_sim wide_sim.c -lm -lpthread   (includes racc_oracle.c)
 * Run:   wide_sim nodes.bin pairs.bin remap.bin rays.bin [order] [maxrays] [quant]
 *        order 0 = full sort (default), 1 = only the nearest child exact
 *        quant 1 = the 64-byte compressed node of racc_kernel_v10.inc: the four child boxes quantised conservatively to 8 bits
 *                  per plane on the box that encloses them (origin + q * scale, decoded with one fmaf exactly as the kernel
 *                  does); every decoded box contains the reference's box, so every box the reference enters is entered.
 */
#include "racc_oracle.c"

#include <stdio.h>

typedef struct { uint32_t ref[4]; float lo[3][4], hi[3][4]; uint32_t pad[4]; } wide_node;   /* 128 B */

static float area6(const float* mn, const float* mx) {
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return dx * dy + dy * dz + dz * dx;
}

typedef struct { uint32_t ref; float mn[3], mx[3]; } cand_t;

/* Collapse: the two children of a BVH2 node are the first candidates; the inner candidate with the largest surface area
 * is replaced (in place: spatial neighbours stay neighbours) by its own two children until there are four. */
static uint32_t collapse(const orc_gpu_node* nodes, uint32_t nodeCount, wide_node* out, uint32_t* remapOut) {
    uint32_t* map = malloc(sizeof(uint32_t) * nodeCount);      /* BVH2 index -> wide index */
    uint32_t* queue = malloc(sizeof(uint32_t) * nodeCount);
    memset(map, 0xFF, sizeof(uint32_t) * nodeCount);
    uint32_t qh = 0, qt = 0, count = 0;
    queue[qt++] = 0; map[0] = count++;
    while (qh < qt) {
        const uint32_t n2 = queue[qh++];
        const orc_gpu_node* n = nodes + n2;
        cand_t c[4]; int k = 2;
        c[0].ref = n->first; memcpy(c[0].mn, n->leftMin, 12); memcpy(c[0].mx, n->leftMax, 12);
        c[1].ref = n->last;  memcpy(c[1].mn, n->rightMin, 12); memcpy(c[1].mx, n->rightMax, 12);
        while (k < 4) {
            int best = -1; float bestA = -1.0f;
            for (int i = 0; i < k; ++i)
                if ((c[i].ref & 0x80000000u) && area6(c[i].mn, c[i].mx) > bestA) { bestA = area6(c[i].mn, c[i].mx); best = i; }
            if (best < 0) break;
            const orc_gpu_node* m = nodes + (c[best].ref & 0x7FFFFFFFu);
            for (int i = k; i > best + 1; --i) c[i] = c[i - 1];
            c[best].ref = m->first; memcpy(c[best].mn, m->leftMin, 12); memcpy(c[best].mx, m->leftMax, 12);
            c[best + 1].ref = m->last; memcpy(c[best + 1].mn, m->rightMin, 12); memcpy(c[best + 1].mx, m->rightMax, 12);
            ++k;
        }
        wide_node* w = out + map[n2];
        memset(w, 0, sizeof(*w));
        for (int i = 0; i < 4; ++i) {
            if (i < k) {
                uint32_t r = c[i].ref;
                if (r & 0x80000000u) {
                    const uint32_t t = r & 0x7FFFFFFFu;
                    if (map[t] == 0xFFFFFFFFu) { map[t] = count++; queue[qt++] = t; }
                    r = 0x80000000u | map[t];
                }
                w->ref[i] = r;
                for (int a = 0; a < 3; ++a) { w->lo[a][i] = omin(c[i].mn[a], c[i].mx[a]); w->hi[a][i] = omax(c[i].mn[a], c[i].mx[a]); }
            } else {
                w->ref[i] = 0;      /* empty slot: never hit (box at +inf entered only by rays with tFar = inf; the model skips it by ref) */
                for (int a = 0; a < 3; ++a) { w->lo[a][i] = INFINITY; w->hi[a][i] = INFINITY; }
            }
        }
    }
    if (remapOut) memcpy(remapOut, map, sizeof(uint32_t) * nodeCount);
    free(map); free(queue);
    return count;
}

/* Conservative 8-bit quantisation of one wide node, in place: lo/hi are replaced by what the kernel decodes.  The frame is the
 * box around the (used) children: origin = its lower corner, scale = extent / 255 rounded up until fmaf(255, scale, origin)
 * reaches the upper corner.  A lower plane takes the largest q whose decoded value does not exceed it, an upper plane the
 * smallest q whose decoded value is not below it — checked with the very fmaf the kernel evaluates. */
static void quantise_node(wide_node* w, double* growth) {
    for (int a = 0; a < 3; ++a) {
        float lo = INFINITY, hi = -INFINITY;
        for (int i = 0; i < 4; ++i) if (w->ref[i]) { lo = omin(lo, w->lo[a][i]); hi = omax(hi, w->hi[a][i]); }
        float scale = (hi - lo) / 255.0f;
        if (!(scale > 0.0f)) scale = 1.17549435e-38f;
        while (fmaf(255.0f, scale, lo) < hi) scale = nextafterf(scale, INFINITY);
        for (int i = 0; i < 4; ++i) {
            if (!w->ref[i]) continue;
            int ql = (int)floorf((w->lo[a][i] - lo) / scale), qh = (int)ceilf((w->hi[a][i] - lo) / scale);
            if (ql < 0) ql = 0; if (ql > 255) ql = 255; if (qh < 0) qh = 0; if (qh > 255) qh = 255;
            while (ql > 0 && fmaf((float)ql, scale, lo) > w->lo[a][i]) --ql;
            while (ql < 255 && fmaf((float)(ql + 1), scale, lo) <= w->lo[a][i]) ++ql;
            while (qh < 255 && fmaf((float)qh, scale, lo) < w->hi[a][i]) ++qh;
            while (qh > 0 && fmaf((float)(qh - 1), scale, lo) >= w->hi[a][i]) --qh;
            const float dl = fmaf((float)ql, scale, lo), dh = fmaf((float)qh, scale, lo);
            if (dl > w->lo[a][i] || dh < w->hi[a][i]) { fprintf(stderr, "quantise: not conservative\n"); exit(4); }
            if (growth) *growth += (double)(w->lo[a][i] - dl) + (double)(dh - w->hi[a][i]);
            w->lo[a][i] = dl; w->hi[a][i] = dh;
        }
    }
}

typedef struct { unsigned long long nv, np, depthSum, depthMax, slots[5], depthHist[64], deepVisits[64]; } wstats;

static void traverse_wide(const wide_node* nodes, const orc_pair* pairs, const uint32_t* remap, const orc_ray* in, orc_result* out, int order, wstats* st) {
    ray_state ray;
    for (int k = 0; k < 3; ++k) { ray.o[k] = in->origin[k]; ray.d[k] = in->dir[k]; }
    ray.tNear = in->minT; ray.tFar = in->maxT;
    const float epsilon = 1e-10f;
    for (int k = 0; k < 3; ++k) if (fabsf(ray.d[k]) < epsilon) ray.d[k] = copysignf(epsilon, ray.d[k]);
    float invDir[3], OoD[3];
    for (int k = 0; k < 3; ++k) { invDir[k] = 1.0f / ray.d[k]; OoD[k] = -ray.o[k] * invDir[k]; }
    hit_state hit = { -1, ray.tFar, 0.0f, 0.0f };
    uint32_t node = 0x80000000u, stack[256], head = 0, maxDepth = 0;
    for (;;) {
        if (node & 0x80000000u) {
            const wide_node* n = nodes + (node & 0x7FFFFFFFu);
            ++st->nv; ++st->deepVisits[head < 63 ? head : 63];      /* stack height at this visit */
            const float tRay = ray.tFar;
            float key[4]; uint32_t ref[4];
            for (int i = 0; i < 4; ++i) {
                float t0 = ray.tNear, t1 = ray.tFar;
                float a[3], b[3];
                for (int k = 0; k < 3; ++k) {      /* the sign of 1/d picks the entry and the exit plane: no min/max needed (monotone fma) */
                    const float lo = fmaf(n->lo[k][i], invDir[k], OoD[k]), hi = fmaf(n->hi[k][i], invDir[k], OoD[k]);
                    a[k] = invDir[k] < 0.0f ? hi : lo; b[k] = invDir[k] < 0.0f ? lo : hi;
                }
                t0 = omax(omax(t0, a[0]), omax(a[1], a[2]));
                t1 = omin(omin(t1, b[0]), omin(b[1], b[2]));
                key[i] = (t0 > t1 || n->ref[i] == 0) ? tRay : t0;
                ref[i] = n->ref[i];
            }
            /* order: nearest last in the array (it is visited next), farthest / missed first */
#define CE(i, j) do { if (key[i] < key[j]) { float tk = key[i]; key[i] = key[j]; key[j] = tk; uint32_t tr = ref[i]; ref[i] = ref[j]; ref[j] = tr; } } while (0)
            if (order == 0) { CE(0, 1); CE(2, 3); CE(0, 2); CE(1, 3); CE(1, 2); }
            else {      /* only the nearest is found exactly (it goes last = visited next); the others keep their slot order */
                int best = 0;
                for (int i = 1; i < 4; ++i) if (key[i] < key[best]) best = i;
                const float tk = key[best]; const uint32_t tr = ref[best];
                for (int i = best; i < 3; ++i) { key[i] = key[i + 1]; ref[i] = ref[i + 1]; }
                key[3] = tk; ref[3] = tr;
            }
#undef CE
            int hits = 0;
            for (int i = 0; i < 4; ++i) if (key[i] != tRay) ++hits;
            ++st->slots[hits];
            if (hits) {
                uint32_t next = 0;
                for (int i = 0; i < 4; ++i) if (key[i] != tRay) { if (next) stack[head++] = next; next = ref[i]; }
                if (head > maxDepth) maxDepth = head;
                node = next;
                continue;
            }
        } else {
            const int32_t first = (int32_t)(node & 0xFFFFFFu);
            const int32_t last = first + (int32_t)(node >> 24);
            for (int32_t i = first; i < last; ++i) { ray.tFar = pair_intersect(pairs, i, &ray, &hit); ++st->np; }
        }
        if (!head) break;
        node = stack[--head];
    }
    st->depthSum += maxDepth; if (maxDepth > st->depthMax) st->depthMax = maxDepth;
    ++st->depthHist[maxDepth < 63 ? maxDepth : 63];
    if (hit.index == -1) { out->triangle = 0xFFFFFFFFu; out->t = out->u = out->v = 0.0f; }
    else {
        uint32_t index = remap[hit.index];
        const uint32_t edge = index >> 30;
        index &= 0x3FFFFFFFu;
        const float bx = hit.u, by = hit.v, bz = 1.0f - hit.u - hit.v;
        float u = bx, v = by;
        if (edge == 1) { u = bz; v = bx; } else if (edge == 2) { u = by; v = bz; }
        out->triangle = index; out->t = hit.t; out->u = u; out->v = v;
    }
}

static void* slurp(const char* path, size_t* n) {
    FILE* f = fopen(path, "rb"); if (!f) { perror(path); exit(3); }
    fseek(f, 0, SEEK_END); *n = (size_t)ftell(f); rewind(f);
    void* p = malloc(*n); if (fread(p, 1, *n, f) != *n) exit(3); fclose(f); return p;
}

int main(int argc, char** argv) {
    if (argc < 5) { fprintf(stderr, "usage: wide_sim nodes.bin pairs.bin remap.bin rays.bin [order] [maxrays]\n"); return 2; }
    size_t nb, pb, rb, mb;
    orc_gpu_node* nodes = slurp(argv[1], &nb);
    orc_pair* pairs = slurp(argv[2], &pb);
    uint32_t* remap = slurp(argv[3], &mb);
    orc_ray* rays = slurp(argv[4], &rb);
    const int order = argc > 5 ? atoi(argv[5]) : 0;
    const uint32_t nodeCount = (uint32_t)(nb / 64);
    uint32_t count = (uint32_t)(rb / 32);
    if (argc > 6 && (uint32_t)atoi(argv[6]) < count) count = (uint32_t)atoi(argv[6]);
    wide_node* wide = aligned_alloc(128, sizeof(wide_node) * nodeCount);
    const uint32_t wideCount = collapse(nodes, nodeCount, wide, 0);
    const int quant = argc > 7 ? atoi(argv[7]) : 0;
    if (quant) { double growth = 0.0; for (uint32_t i = 0; i < wideCount; ++i) quantise_node(wide + i, &growth); fprintf(stderr, "quantised %u nodes, mean plane displacement %.3g\n", wideCount, growth / (24.0 * wideCount)); }
    unsigned long long full = 0;
    for (uint32_t i = 0; i < wideCount; ++i) { int k = 0; for (int j = 0; j < 4; ++j) k += wide[i].ref[j] != 0; full += (unsigned)k; }
    wstats st; memset(&st, 0, sizeof(st));
    unsigned long long nv2 = 0, np2 = 0, d2 = 0, ties = 0, diffs = 0, d2max = 0, closer = 0;
    for (uint32_t i = 0; i < count; ++i) {
        orc_result a, b; uint32_t nv, np, dp;
        traverse_one(nodes, pairs, remap, 0, 0, 0, rays + i, &a, &nv, &np, &dp);
        nv2 += nv; np2 += np; d2 += dp; if (dp > d2max) d2max = dp;
        int finite = isfinite(rays[i].minT) && !isnan(rays[i].maxT);
        for (int k = 0; k < 3; ++k) finite = finite && isfinite(rays[i].origin[k]) && isfinite(rays[i].dir[k]);
        if (!finite) continue;
        traverse_wide(wide, pairs, remap, rays + i, &b, order, &st);
        if (a.triangle != b.triangle || f2u(a.t) != f2u(b.t) || f2u(a.u) != f2u(b.u) || f2u(a.v) != f2u(b.v)) {
            if (a.triangle != 0xFFFFFFFFu && b.triangle != 0xFFFFFFFFu && fabsf(a.t - b.t) <= 1e-6f * fabsf(a.t)) ++ties;
            else if (quant && b.triangle != 0xFFFFFFFFu && (a.triangle == 0xFFFFFFFFu || b.t < a.t)) ++closer;      /* the larger boxes let a hit through that the reference's own box test culled (box vs triangle rounding): a closer hit, not a wrong one */
            else { ++diffs; if (diffs < 5) fprintf(stderr, "diff ray %u: %u %g %g %g | %u %g %g %g\n", i, a.triangle, a.t, a.u, a.v, b.triangle, b.t, b.u, b.v); }
        }
    }
    printf("{\"rays\": %u, \"order\": %d, \"bvh2_nodes\": %u, \"wide_nodes\": %u, \"slots_used\": %.2f, \"nv2\": %.2f, \"np2\": %.2f, \"depth2_mean\": %.2f, \"depth2_max\": %llu, "
           "\"nv4\": %.2f, \"np4\": %.2f, \"depth4_mean\": %.2f, \"depth4_max\": %llu, \"hits_per_visit\": [%.3f, %.3f, %.3f, %.3f, %.3f], \"ties\": %llu, \"closer_than_reference\": %llu, \"differences\": %llu}\n",
           count, order, nodeCount, wideCount, (double)full / wideCount, (double)nv2 / count, (double)np2 / count, (double)d2 / count, d2max,
           (double)st.nv / count, (double)st.np / count, (double)st.depthSum / count, st.depthMax,
           (double)st.slots[0] / st.nv, (double)st.slots[1] / st.nv, (double)st.slots[2] / st.nv, (double)st.slots[3] / st.nv, (double)st.slots[4] / st.nv, ties, closer, diffs);
    if (getenv("WIDE_SIM_HIST")) {
        fprintf(stderr, "rays by deepest stack / visits by stack height at the visit:\n");
        unsigned long long cr = 0, cv = 0;
        for (int d = 0; d < 64; ++d) { cr += st.depthHist[d]; cv += st.deepVisits[d]; if (st.depthHist[d] || st.deepVisits[d]) fprintf(stderr, "  %2d: rays %8llu (cum %.5f)  visits %10llu (cum %.5f)\n", d, st.depthHist[d], (double)cr / count, st.deepVisits[d], (double)cv / st.nv); }
    }
    return diffs ? 1 : 0;
}
