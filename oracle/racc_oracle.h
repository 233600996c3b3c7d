/*
 * racc_oracle.h — CPU restatement of the reference's intersect-batch hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under rayaccel_amd/ (the product) may
 * include, link or load this; only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker.
 *
 * PINNING.  The reference ships no tests, golden vectors or fixtures for this
 * path, and the correctness reference north_star names — its CPU path, three
 * lines of glue around a binary-only third-party library (Intel Embree 2.x,
 * ~2.7.0) — cannot run anywhere here (binaries absent, sources not vendored).
 * What IS available is the reference's own GPU path: oracle/_ref holds its
 * OpenCL `traversal` kernel, compiled from the source where it lies with the
 * reference's own build options (oracle/Makefile, target `ref`), and
 * tests/test_gpu_reference_kernel.py runs it on the MI355X next to this
 * restatement on the same buffers (1M-ray batches included): primId identical
 * up to exact-distance ties, t/u/v within 1e-4 relative (the reference kernel
 * is fast-math, so not bit-comparable).  The restatement is additionally
 * cross-checked against an independent double-precision brute-force arbiter
 * (orc_brute_*).  The reference's C++ translation units are not built: they
 * need headers/libraries this image lacks (<OpenCL/cl.h> Apple path,
 * <libkern/OSAtomic.h>, libembree) and stand-ins for those are not allowed.
 *
 * All file:line citations are relative to /root/reference/.
 */
#ifndef RACC_ORACLE_H
#define RACC_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* RayAccelerator/RayAccelerator.h:59-64 (32 B) */
typedef struct { float origin[3]; float minT; float dir[3]; float maxT; } orc_ray;
/* RayAccelerator/RayAccelerator.h:66-76 (16 B); triangle==0xFFFFFFFF => t,u,v hold miss r,g,b */
typedef struct { uint32_t triangle; float t, u, v; } orc_result;
/* RayAccelerator/Bvh2.h:15-22 (48 B) */
typedef struct {
    uint32_t kind, parent, first, last;
    float bbMin[3]; uint32_t dummy0;
    float bbMax[3]; uint32_t dummy1;
} orc_bvh2_node;
/* RayAccelerator/Scene.cpp:73-78 (64 B): only inner nodes are stored */
typedef struct {
    uint32_t kind, parent, first, last;
    float leftMin[3], leftMax[3], rightMin[3], rightMax[3];
} orc_gpu_node;
/* RayAccelerator/Scene.cpp:83-87 (48 B) */
typedef struct { float e1[3], e3x, e2[3], e3y, p0[3], e3z; } orc_pair;

/* Bvh2.cpp:257-535,772-907 — full-sweep SAH BVH2, deterministic (single-thread
 * node numbering).  nodes: capacity 2*T; triangles: capacity T.  Returns 0 on
 * success.  vertices are xyzw (16 B). */
int orc_bvh2_build(const float* vertices, uint32_t vertexCount,
                   const uint32_t* indices, uint32_t triangleCount,
                   orc_bvh2_node* nodes, uint32_t* triangles, uint32_t* nodeCount);

/* Scene.cpp:109-181,237-339 — leaf pair merge, remap, 64 B node flatten, pair
 * padding.  Capacities: gpuNodes >= nodeCount, pairs >= T + 32, remap >= 2*T.
 * bvh nodes' first/last of leaves are rewritten to pair ranges (as the
 * reference does in place).  Returns 0 on success, <0 if the scene violates a
 * format limit (root not inner, >127 pairs/leaf, >=2^24 pairs, tri id >= 2^30). */
int orc_scene_pack(orc_bvh2_node* nodes, uint32_t nodeCount, const uint32_t* triangles,
                   const float* vertices, const uint32_t* indices, uint32_t triangleCount,
                   orc_gpu_node* gpuNodes, uint32_t* gpuNodeCount,
                   orc_pair* pairs, uint32_t* pairCount, uint32_t* pairCountPadded,
                   uint32_t* remap);

/* Kernels.h:139-242 — scalar restatement of the `traversal` kernel over the
 * reference-format buffers, rays [start,end).  env may be NULL (miss rgb = 0).
 * nv/np/depth (each may be NULL) receive per-ray inner-node visits, pair tests
 * and maximum stack height — the counters SURVEY.md §8(d)'s algorithmic-byte
 * figure is built from. */
void orc_traverse(const orc_gpu_node* nodes, const orc_pair* pairs, const uint32_t* remap,
                  const float* env, uint32_t envW, uint32_t envH,
                  const orc_ray* rays, orc_result* results, uint32_t start, uint32_t end,
                  uint32_t* nv, uint32_t* np, uint32_t* depth);

/* Analysis aid: when set (non-NULL), every inner-node visit increments hist[node]. */
void orc_set_visit_histogram(uint32_t* hist);

/* Same as orc_traverse over [0,count) in slices of `slice` rays
 * (cpuTestBatch = 1024, RayAccelerator.cpp:197-212,438) on `threads` pthreads,
 * `repeat` passes over the batch inside one call (so that thread start-up is
 * amortised when timing).  Used by bench.py's cpu_baseline leg ("port"). */
void orc_traverse_mt(const orc_gpu_node* nodes, const orc_pair* pairs, const uint32_t* remap,
                     const float* env, uint32_t envW, uint32_t envH,
                     const orc_ray* rays, orc_result* results, uint32_t count,
                     uint32_t slice, uint32_t threads, uint32_t repeat);

/* racc_oracle_simd.c — the same traversal eight rays at a time in AVX2 (every lane does what orc_traverse does for its ray: results are
 * bit-identical; the reference hands its CPU leg 8 rays at a time, Scene.cpp:386-428).  bench.py's cpu_baseline kind "simd-port". */
int orc_simd_available(void);
void orc_traverse_simd(const orc_gpu_node* nodes, const orc_pair* pairs, const uint32_t* remap,
                       const float* env, uint32_t envW, uint32_t envH,
                       const orc_ray* rays, orc_result* results, uint32_t start, uint32_t end);
void orc_traverse_simd_mt(const orc_gpu_node* nodes, const orc_pair* pairs, const uint32_t* remap,
                          const float* env, uint32_t envW, uint32_t envH,
                          const orc_ray* rays, orc_result* results, uint32_t count,
                          uint32_t slice, uint32_t threads, uint32_t repeat, uint32_t width /* 8, or 16 where orc_simd512_available() */);
/* racc_oracle_simd512.c — the same with sixteen rays per zmm register (AVX-512F/DQ/VL hosts: the GPU boxes' EPYC 9575F); bit-identical too. */
int orc_simd512_available(void);
void orc_traverse_simd512(const orc_gpu_node* nodes, const orc_pair* pairs, const uint32_t* remap,
                          const float* env, uint32_t envW, uint32_t envH,
                          const orc_ray* rays, orc_result* results, uint32_t start, uint32_t end);

/* Kernels.h:213-222 — miss colour for a (clamped) direction; OpenCL
 * CLK_NORMALIZED_COORDS_TRUE | CLAMP_TO_EDGE | FILTER_LINEAR semantics. */
void orc_env_sample(const float* env, uint32_t envW, uint32_t envH, const float dir[3], float rgb[3]);

/* Independent arbiter: double-precision Moller-Trumbore over ALL triangles (no
 * BVH).  u = weight of the triangle's 2nd index, v = of the 3rd (Embree
 * convention, consumer PathTracingRenderer.cpp:218-227).  tri[i] = 0xFFFFFFFF
 * on miss.  t2[i] (may be NULL) = second-closest distinct-triangle t (inf if
 * none), for tie detection. */
void orc_brute_closest(const float* vertices, const uint32_t* indices, uint32_t triangleCount,
                       const orc_ray* rays, uint32_t count,
                       uint32_t* tri, double* t, double* u, double* v, double* t2);
/* Intersect one named triangle in double precision; returns 1 on a hit inside
 * (minT, maxT] (Kernels.h:88-89 interval), 0 otherwise. */
int orc_brute_one(const float* vertices, const uint32_t* indices, uint32_t triangle,
                  const orc_ray* ray, double* t, double* u, double* v);

#ifdef __cplusplus
}
#endif
#endif
