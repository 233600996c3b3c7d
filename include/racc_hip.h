/*
 * racc_hip.h — C-ABI of the MI355X (gfx950) intersect-batch engine.
 *
 * This is the boundary the reference's host code would bind instead of its
 * OpenCL calls: plain pointers and sizes, `int` status (0 = ok, <0 = error,
 * text via racc_hip_last_error()), never throws, no torch/HIP types in any
 * signature.  Each entry cites the reference interface it replaces
 * (file:line relative to the reference checkout).  The reference-side stubs a
 * maintainer would add are shown in INTEGRATION.md.
 *
 * Record layouts are the reference's, byte for byte:
 *   Ray     32 B  {origin[3], minT, dir[3], maxT}         RayAccelerator.h:59-64
 *   Result  16 B  {u32 triangle; t,u,v | r,g,b}           RayAccelerator.h:66-76
 *                 triangle == 0xFFFFFFFF => miss, floats = environment rgb
 *   Node    64 B  {kind,parent,first,last, leftMin[3],leftMax[3],
 *                  rightMin[3],rightMax[3]}                Scene.cpp:73-78
 *   Pair    48 B  {e1.xyz,e3.x, e2.xyz,e3.y, p0.xyz,e3.z}  Scene.cpp:83-87
 *   remap   u32   tri | edge<<30, two per pair             Scene.cpp:132-133
 */
#ifndef RACC_HIP_H
#define RACC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RACC_HIP_OK                 0
#define RACC_HIP_ERR_INVALID       -1   /* bad argument / malformed scene blob */
#define RACC_HIP_ERR_DEVICE        -2   /* a HIP call failed */
#define RACC_HIP_ERR_NO_DEVICE     -3   /* no gfx950 device / extension unusable */
#define RACC_HIP_ERR_LIMIT         -4   /* scene exceeds a format limit (Scene.cpp:294-312) */
#define RACC_HIP_ERR_NOMEM         -5

#define RACC_HIP_MAX_LANES          8   /* >= gpuSubmissionThreads (RayAccelerator.cpp:436) */
#define RACC_HIP_LANE_AUTO 0xFFFFFFFFu  /* racc_hip_intersect_device: the engine picks the lane, round robin; racc_hip_wait: all lanes */

typedef struct racc_hip_ctx   racc_hip_ctx;    /* one per (process, GPU) */
typedef struct racc_hip_scene racc_hip_scene;  /* device-resident flattened BVH2 + pairs */
typedef struct racc_hip_env   racc_hip_env;    /* device-resident RGBA32F probe image */
typedef struct racc_host_scene racc_host_scene;/* host-side build product (reference-format blobs) */

typedef struct racc_hip_options {
    uint32_t struct_size;      /* = sizeof(racc_hip_options) */
    uint32_t lanes;            /* concurrent submission lanes, each with its own HIP stream (0 => default, racc_hip_lane_count)
                                  (≙ gpuSubmissionThreads queues, RayAccelerator.cpp:711-717); 0 => 4 */
    uint32_t waves_per_simd;   /* persistent-grid occupancy target, 1..8; 0 => engine default */
    uint32_t kernel_variant;   /* 0 => engine default (V8: reference traversal order, results bit-identical to the oracle);
                                  50 => the compressed 4-wide kernel (V10: 64-byte nodes, child boxes quantised conservatively to
                                  8 bits — half the node bytes and vector-memory instructions per ray; 4-8 % faster on incoherent
                                  batches; same closest hit except exact-distance ties and hits the reference's own box test drops
                                  although its pair test accepts them: there the CLOSER hit, confirmed by the double-precision
                                  arbiter, is reported); 45 => the uncompressed 4-wide kernel (V9, 128-byte nodes); 60-63 => the
                                  default kernel with the top of the tree (524 / 768 / 260 / 128 node records) held in LDS, 16
                                  waves per CU as workgroups of 1024 / 1024 / 512 / 256 threads: bit-identical results, measured
                                  slower, kept for the A/B; others: DESIGN.md §3, racc_hip_variant_available */
    uint32_t refill_min;       /* idle lanes that trigger a wave refill; 0 => default */
    uint32_t leaf_min;         /* leaf-holding lanes that trigger a leaf step; 0 => default */
    uint32_t chunk;            /* rays a wave dequeues per cursor atomic; 0 => default */
    uint32_t tail_active;      /* waves with <= this many live rays ("thin") run both step bodies per iteration;
                                  0 => default (32), > 64 => never */
    uint32_t regroup_period;   /* V3 kernels: scheduling iterations between workgroup-wide regroupings; 0 => default */
    uint32_t thin_reps;        /* V2 kernels: inner steps a thin wave runs per scheduling iteration; 0 => default (8) */
    uint32_t inner_reps;       /* V2 kernels: inner steps any other wave runs per scheduling iteration; 0 => default (3) */
    uint32_t coop_same_pct;    /* V7 kernels: inner steps fetch nodes quad-cooperatively through LDS-DMA while fewer than this
                                  percentage of the inner lanes hold the same node as their quad neighbour (divergent waves);
                                  0 => default (20), > 100 => never, 100 => always */
    uint32_t time_kernels;     /* != 0: an event pair around every traversal kernel (racc_hip_read_kernel_times); costs two
                                  hipEventRecord per launch, so off by default */
    uint32_t drain_prefetch;   /* what a thin wave (few live rays: a launch's drain, a small launch) does about the latency of its node
                                  fetches.  0 => default: nothing.  1: touch both children's records as soon as the child refs arrive
                                  (round 2: -3 % on a 64k-ray batch, +4..10 % on 256k..1M rays).  3: every step also requests the record
                                  BEHIND each lane's node — its likeliest next node in the device order, 46 % of the visits — into the
                                  lane's LDS-DMA slots, and a lane that goes on into that record reads it from LDS (round 5, built at
                                  the round-4 verdict's request; measured slower: 64k rays 0.130 vs 0.113-0.116 ms, 1M 0.345 vs 0.320 —
                                  a wave steps in lockstep, one lane's miss costs the whole step).  Same results in all forms */
    uint32_t leaf_step;        /* 0/1 => the leaf step runs inside the kernel's assembly block and takes the inner lanes' step along
                                  (one memory round trip for both); 2 => in C++ through the block's LEAF door; 3 => in the block,
                                  not fused (A/B; same results) */
    uint32_t wide_below;       /* launches of fewer rays than this use the 4-wide kernel (kernel_variant 45) instead of the
                                  context's kernel: a launch of <= ~200k rays is all dependent chain, and the wide tree halves
                                  it (27k-ray launch, the reference's stream size: 0.10 instead of 0.12 ms).  Results: see
                                  kernel_variant 45.  0 => never (default) */
    uint32_t chain_launches;   /* 0/1 => device-resident batches issued on the engine's own streams (racc_hip_intersect_device with
                                  stream = NULL) are chained: waves that run out of rays in one batch go on with the next one
                                  issued, so a sequence of batches runs like one long launch (no drain between them).  A batch's
                                  arrays belong to the engine until racc_hip_wait on its lane has returned and may be re-issued
                                  at once after that (the exact rule: racc_hip_intersect_device below).  2 => off: every launch
                                  stands alone, the lanes' launches merely overlap (same rule).  Round 5: the chain is LAZY — once
                                  a chain has its kernels (one per lane in rotation), a further batch is only published to them (a
                                  descriptor written by a one-thread kernel): no traversal kernel that would find nothing left, no
                                  miss-shading kernel (the chained kernels sample the probe image in their epilogue).  racc_hip_wait /
                                  racc_hip_synchronize establish completion — the chain's kernels have ended and every published
                                  batch's rays were handed out — and trace whatever the chain did not reach (it had ended before the
                                  batch was linked) with a catch-up kernel; hence results are defined, as before, when the wait
                                  returns.  A caller who waits for every batch gains nothing from a chain and would pay for its start and
                                  end (0.46 against 0.35 ms per 1M-ray batch): with the default threshold (chain_min_rays = 0), after
                                  two waits in a row that found a chain of one batch, a batch issued while nothing else of the
                                  context is in flight is launched stand-alone; the first batch issued while another is in flight
                                  starts a chain again (RACC_CHAIN_SOLO=0 switches that off).  3 => round 4's form: every chained
                                  launch brings its own kernel + miss-shading kernel */
    uint32_t chain_min_rays;   /* only batches of at least this many rays are chained; smaller ones are launched stand-alone on their
                                  lane (they overlap like any two lanes' launches).  A chained launch costs the host more stream
                                  operations, and a batch that is worked off before its successor is linked breaks the chain: 27,648-ray
                                  streams back to back run 126 Mrays/s chained, 657 stand-alone; the crossover is between 512k and 1M
                                  rays.  0 => default (786,432); 1 => chain every batch (tests) */
} racc_hip_options;

typedef struct racc_hip_scene_info {
    uint32_t node_count;       /* inner nodes */
    uint32_t pair_count;       /* pairs incl. padding */
    uint32_t remap_count;
    uint32_t inner_height;     /* levels of inner nodes on the longest root path */
    uint32_t max_leaf_pairs;
    uint32_t spill_levels;     /* stack levels beyond the LDS-resident ones */
    uint64_t device_bytes;
} racc_hip_scene_info;

/* Per-launch counters of the last racc_hip_intersect_device* call on a lane (debug/bench). */
typedef struct racc_hip_launch_info {
    uint32_t grid_blocks, block_threads;
    uint32_t lds_bytes_per_block;
    uint32_t waves_per_simd;
    float    last_kernel_ms;   /* hipEvent time of the traversal kernel alone */
} racc_hip_launch_info;

const char* racc_hip_last_error(void);             /* thread-local, never NULL */
const char* racc_hip_version(void);
/* The lanes a context has (options.lanes, or the default: 4; without chained launches 6 when the HIP runtime was given >= 8
 * hardware queues, GPU_MAX_HW_QUEUES) and how many of them RACC_HIP_LANE_AUTO rotates over (3; without chained launches 6 in
 * that case: kernels of two streams that share a hardware queue do not overlap; RACC_AUTO_LANES overrides).  A caller that rotates result buffers needs at least
 * `auto_lanes` of them.  Either pointer may be NULL. */
int racc_hip_lane_count(const racc_hip_ctx* ctx, uint32_t* lanes, uint32_t* auto_lanes);

/* 1 if racc_hip_options::kernel_variant = n selects a kernel of this build (0 = the default always does; the earlier
 * generations and ablations exist only in a `make EXPERIMENTAL=1` build), else 0.  Needs no GPU. */
int racc_hip_variant_available(uint32_t kernel_variant);

/* ≙ clGetDeviceIDs, RayAccelerator.cpp:467-478 (reference picks devices[0]). */
int racc_hip_device_count(int* count);

/* ≙ racc::createContext's OpenCL half: program build + queues, RayAccelerator.cpp:463-514,700-717.
 * Fails with RACC_HIP_ERR_NO_DEVICE if `device` is not a gfx950 GPU: there is no CPU fallback. */
int racc_hip_create(int device, const racc_hip_options* opts, racc_hip_ctx** out);
int racc_hip_destroy(racc_hip_ctx* ctx);           /* ≙ racc::destroy(Context*), RayAccelerator.cpp:761-788 */

/* ≙ the three clCreateBuffer(COPY_HOST_PTR) calls, Scene.cpp:342-346.  Inputs are the
 * reference-format blobs (copied; caller keeps ownership).  The blob is validated
 * (child indices, pair ranges, no cycles) and re-laid-out for the device. */
int racc_hip_scene_upload(racc_hip_ctx* ctx,
                          const void* nodes64, uint32_t node_count,
                          const void* pairs48, uint32_t pair_count,
                          const uint32_t* remap, uint32_t remap_count,
                          racc_hip_scene** out);
int racc_hip_scene_free(racc_hip_ctx* ctx, racc_hip_scene* scene);   /* ≙ Scene.cpp:362-369 */
int racc_hip_scene_get_info(const racc_hip_scene* scene, racc_hip_scene_info* info);

/* ≙ clCreateImage(RGBA, FLOAT), Environment.cpp:36-50.  rgba is width*height*4 floats, copied. */
int racc_hip_env_upload(racc_hip_ctx* ctx, const float* rgba, uint32_t width, uint32_t height, racc_hip_env** out);
int racc_hip_env_free(racc_hip_ctx* ctx, racc_hip_env* env);         /* ≙ Environment.cpp:63-64 */

/* ≙ clCreateBuffer(CL_MEM_USE_HOST_PTR) per ray stream, RayAccelerator.cpp:635-645: page-locks the
 * stream's host arrays so the copies below run at PCIe rate.  Optional. */
int racc_hip_register_stream(racc_hip_ctx* ctx, void* rays, void* results, uint32_t capacity);
int racc_hip_unregister_stream(racc_hip_ctx* ctx, void* rays, void* results);
/* Same for one block that holds many streams (the reference allocates all of them in one block,
 * RayAccelerator.cpp:547-568). */
int racc_hip_register_host(racc_hip_ctx* ctx, void* ptr, uint64_t bytes);
int racc_hip_unregister_host(racc_hip_ctx* ctx, void* ptr);

/* ≙ clSetKernelArg x7 + clEnqueueNDRangeKernel + clFinish on one submission thread's queue,
 * RayAccelerator.cpp:378-404.  Host Ray[count] in, host Result[count] out, in place and in order
 * (RayStream contract, RayAccelerator.h:78-83).  `lane` in [0, lanes): calls on distinct lanes may
 * run concurrently from different threads.  env may be NULL (miss rgb = 0).  Blocking. */
int racc_hip_intersect(racc_hip_ctx* ctx, const racc_hip_scene* scene, const racc_hip_env* env,
                       const void* rays, void* results, uint32_t count, uint32_t lane);
/* Same, split into enqueue and wait (≙ clEnqueueNDRangeKernel / clFinish).  The enqueue returns as soon as the batch's three stages
 * — rays in over PCIe, traversal, results out over PCIe — are queued; it blocks only while the SAME lane's previous host batch is
 * still in flight (a lane has one pair of staging arrays).  With page-locked arrays (racc_hip_register_*) the stages of batches
 * issued on different lanes overlap: all copies into the GPU go through one stream, back to back, all copies out through another,
 * the kernels run on the lanes' streams in between — a caller that rotates 3-4 lanes keeps the link busy in both directions
 * (≙ the reference's gpuSubmissionThreads queues, RayAccelerator.cpp:711-717; 1M-ray batches back to back: DESIGN.md §6).
 * The host arrays belong to the engine until racc_hip_wait on the lane (or RACC_HIP_LANE_AUTO: every lane) has returned. */
int racc_hip_intersect_async(racc_hip_ctx* ctx, const racc_hip_scene* scene, const racc_hip_env* env,
                             const void* rays, void* results, uint32_t count, uint32_t lane);
/* racc_hip_wait(lane): the lane's host batch is copied out / its device-resident launches have ended.  With chained launches (the default
 * for device-resident batches of >= chain_min_rays rays) a batch is complete when its chain is, whichever lane it was issued on: while a
 * chain has batches outstanding, a wait for ONE lane waits for every lane's kernels (a context-wide wait) and holds the chain's mutex
 * meanwhile — threads that each drive a lane of their own and wait often should create the context with chain_launches = 2. */
int racc_hip_wait(racc_hip_ctx* ctx, uint32_t lane);

/* MI355X-sized dispatch: a whole set of ray streams in ONE launch.  The reference launches once per
 * <=27k-ray stream (RayAccelerator.cpp:369-404); 27k rays occupy 5 % of the 524,288 lanes a MI355X keeps
 * resident and every launch pays the latency of its longest ray, so the scheduler (racc::render) hands over
 * everything that is ready.  rays[i]/results[i] are host arrays of counts[i] records; results land in place
 * and in order per stream.  Blocking.  When the arrays are page-locked (racc_hip_register_*) and the launch has
 * >= 512k rays it is cut into 256k-ray slices whose PCIe copies run beside the neighbouring slices' kernels. */
int racc_hip_intersect_streams(racc_hip_ctx* ctx, const racc_hip_scene* scene, const racc_hip_env* env,
                               uint32_t n_streams, const void* const* rays, void* const* results,
                               const uint32_t* counts, uint32_t lane);
/* The same without the wait (≙ racc_hip_intersect_async for a set of streams): the pointer arrays are read during the call, the
 * ray/result arrays belong to the engine until racc_hip_wait(ctx, lane). */
int racc_hip_intersect_streams_async(racc_hip_ctx* ctx, const racc_hip_scene* scene, const racc_hip_env* env,
                                     uint32_t n_streams, const void* const* rays, void* const* results,
                                     const uint32_t* counts, uint32_t lane);

/* Device-resident variant: d_rays/d_results are device pointers (e.g. torch tensors' data_ptr()).
 * `stream` is a hipStream_t passed as void* (NULL => the lane's own stream).  Asynchronous.
 * A lane owns one ray cursor and spill area, so its launches never overlap: a launch that goes to another stream than the
 * lane's previous one first waits (on the device) for that one.  Launches on DIFFERENT lanes do overlap — one's drain hides
 * under the next one's bulk — which is how the reference keeps its gpuSubmissionThreads queues busy
 * (RayAccelerator.cpp:711-717).  lane = RACC_HIP_LANE_AUTO rotates over the lanes, so a single-threaded caller that issues
 * batch after batch (stream = NULL) gets that overlap without managing lanes; racc_hip_wait(ctx, RACC_HIP_LANE_AUTO) or
 * racc_hip_synchronize then waits for all of them.  A launch of <= 2M rays issued while ANOTHER lane has a launch the host has not
 * waited for takes a thin grid (2 waves per SIMD instead of 5), so that up to three launches are co-resident; "in flight" is what the
 * caller has issued and not yet waited for — a property of the call sequence, not of the GPU's progress at that instant: wait for a
 * lane (racc_hip_wait) before issuing a launch that should have the machine to itself.
 * With stream = NULL the batch must be resident and final at the call, and — racc_hip_options::chain_launches, the default, for
 * batches of at least racc_hip_options::chain_min_rays rays (786,432) —
 * launches are CHAINED: waves of the launches issued before may start on this batch at once and carry on through it, so that
 * a sequence of batches runs like one long launch (20 batches of 1M rays back to back: 0.28 instead of 0.33 ms each).  A launch
 * on a caller's stream is never chained.
 *
 * Array reuse under chained launches — the exact rule (≙ the in-place, in-order ray-stream contract, RayAccelerator.h:78-83,
 * RayAccelerator.cpp:369-410: the reference recycles a stream's arrays bounce after bounce):
 *   - a batch's ray and result arrays belong to the engine from the call until racc_hip_wait on ITS lane (or
 *     RACC_HIP_LANE_AUTO / racc_hip_synchronize) has returned; a lane's wait returns when its batch is complete, whoever traced it;
 *   - after that wait the arrays may be rewritten and re-issued AT ONCE, by any means that has completed before the next call
 *     (racc_hip_memcpy_h2d, a copy or kernel on another stream that the caller has synchronised), while the batches of the
 *     other lanes are still in flight and their kernels go on to trace the re-issued one.
 * Why a kernel that outlives kernel boundaries reads the NEW contents: every batch is linked into the chain by a kernel that is
 * dispatched after the call (one thread, on the engine's control stream).  The acquire of that dispatch — ROCm gives every kernel
 * dispatch packet an agent-scope acquire fence — makes the command processor invalidate the vector L1 of every CU and the
 * non-coherent lines of every XCD's L2 BEFORE the kernel runs, i.e. before the link can become visible; no wave reads a batch's
 * rays before it has seen the link (a relaxed agent-scope load, served by the L2), and nothing writes the array in between.  The
 * ray records themselves are loaded with system-scope loads (sc0 sc1: they always miss the CU's L1), which takes the L1 out of the
 * argument altogether.  Results travel the other way through kernel ends: a batch is complete when its own kernel and every kernel
 * issued before it have ended (their end-of-kernel release writes the L2s back), which is what the lane's wait waits for.
 * tests/test_gpu_reuse.py recycles three ray and three result buffers through 2,000 chained batches of 64 ... 1M rays, rewritten
 * by host copies and by a copy kernel on another stream, on a scene that thrashes the L2s and on one that leaves them idle:
 * every batch bit-exact.
 * What this rests on: the dispatch-packet acquire is behaviour of the HIP runtime and the command processor, not of this library —
 * established on ROCm 7.2 (HIP 7.2.26015), gfx950 in SPX partition mode, and checked there by the test above, not derived from a
 * specification.  On any other runtime or partition mode the GUARANTEED rule is the stricter one: re-issue an array only after
 * racc_hip_synchronize (no kernel of the chain is alive then), or create the context with chain_launches = 2 (every launch stands
 * alone; a kernel never outlives the boundary that publishes its inputs). */
int racc_hip_intersect_device(racc_hip_ctx* ctx, const racc_hip_scene* scene, const racc_hip_env* env,
                              const void* d_rays, void* d_results, uint32_t count,
                              uint32_t lane, void* stream);
/* Runs the device variant `iters` times back to back on the lane's stream and writes each launch's
 * traversal-kernel duration in milliseconds (HIP events on that stream) to ms[iters]. Blocking. */
int racc_hip_intersect_device_timed(racc_hip_ctx* ctx, const racc_hip_scene* scene, const racc_hip_env* env,
                                    const void* d_rays, void* d_results, uint32_t count,
                                    uint32_t lane, uint32_t iters, float* ms);
int racc_hip_get_launch_info(racc_hip_ctx* ctx, uint32_t lane, racc_hip_launch_info* info);
/* With racc_hip_options::time_kernels: durations (ms, HIP events on the stream each launch went to) of the lane's traversal
 * kernels since the last call, oldest first, at most `capacity` (the engine keeps the last 256); *n = how many.  Waits for them. */
int racc_hip_read_kernel_times(racc_hip_ctx* ctx, uint32_t lane, float* ms, uint32_t capacity, uint32_t* n);
/* Scheduling statistics accumulated by the statistics build of the default kernel (kernel_variant 42) on a lane:
 * [0] inner steps (wave-level) [1] lanes live in them [2] leaf steps [3] lanes live in them
 * [4] refill iterations [5] rays loaded [6] cursor dequeues [7] waves; shader-clock cycles summed over waves:
 * [8],[10] unused by variant 42 [12] refill iterations [13] wave lifetime;
 * [14] inner steps in which ALL live inner lanes hold the same node (what a wave-uniform scalar step could serve)
 * [15] inner steps fetched quad-cooperatively.  stats16 has 16 entries.  Waits for the lane. */
int racc_hip_read_stats(racc_hip_ctx* ctx, uint32_t lane, uint64_t* stats16, int reset);

/* Device memory helpers for hosts that do not bring their own allocator. */
int racc_hip_malloc(racc_hip_ctx* ctx, uint64_t bytes, void** d_ptr);
int racc_hip_free(racc_hip_ctx* ctx, void* d_ptr);
int racc_hip_memcpy_h2d(racc_hip_ctx* ctx, void* d_dst, const void* src, uint64_t bytes);
int racc_hip_memcpy_d2h(racc_hip_ctx* ctx, void* dst, const void* d_src, uint64_t bytes);
/* Device-to-device copy on `stream` (a hipStream_t as void*, e.g. from racc_hip_stream_create; NULL = the default stream): the
 * runtime's copy kernel.  For hosts that stage ray batches in HBM and hand the engine a rotating set of arrays. */
int racc_hip_memcpy_d2d_async(racc_hip_ctx* ctx, void* d_dst, const void* d_src, uint64_t bytes, void* stream);
int racc_hip_synchronize(racc_hip_ctx* ctx);
/* HIP streams for hosts that do not bring their own (passed back as the `stream` of racc_hip_intersect_device). */
int racc_hip_stream_create(racc_hip_ctx* ctx, void** stream);
int racc_hip_stream_synchronize(racc_hip_ctx* ctx, void* stream);      /* also reports a watchdog trip */
int racc_hip_stream_destroy(racc_hip_ctx* ctx, void* stream);

/* ---- multi-GPU: device groups ----------------------------------------------------------------------
 * ≙ nothing in the reference (its cl_context drives devices[0], RayAccelerator.cpp:467-478).  A group is one engine context per
 * entry of `devices` (entries may repeat an ordinal: rehearsal on one GPU); the scene and the environment are replicated on
 * every member (read-only data, Scene.cpp:342-346); racc_hip_group_intersect cuts a host batch into contiguous shards of whole
 * 64-ray chunks, traces them concurrently (one persistent host thread and one PCIe link per GPU) and lands the results in place, in order.
 * No exchange between GPUs.  racc_hip_group_ctx gives the members for everything else (lanes, device-resident batches). */
typedef struct racc_hip_group racc_hip_group;
typedef struct racc_hip_group_scene racc_hip_group_scene;
typedef struct racc_hip_group_env racc_hip_group_env;
int racc_hip_group_create(const int* devices, uint32_t n, const racc_hip_options* opts, racc_hip_group** out);
int racc_hip_group_destroy(racc_hip_group* group);
uint32_t racc_hip_group_size(const racc_hip_group* group);
racc_hip_ctx* racc_hip_group_ctx(racc_hip_group* group, uint32_t i);
int racc_hip_group_scene_upload(racc_hip_group* group, const void* nodes64, uint32_t node_count, const void* pairs48, uint32_t pair_count,
                                const uint32_t* remap, uint32_t remap_count, racc_hip_group_scene** out);
int racc_hip_group_scene_free(racc_hip_group* group, racc_hip_group_scene* scene);
int racc_hip_group_env_upload(racc_hip_group* group, const float* rgba, uint32_t width, uint32_t height, racc_hip_group_env** out);
int racc_hip_group_env_free(racc_hip_group* group, racc_hip_group_env* env);
int racc_hip_group_intersect(racc_hip_group* group, const racc_hip_group_scene* scene, const racc_hip_group_env* env,
                             const void* rays, void* results, uint32_t count);
/* Device-resident counterpart (≙ racc_hip_intersect_device per member): d_rays[i] / d_results[i] are device pointers on member i's
 * GPU holding that member's shard of counts[i] rays (racc_hip_malloc on racc_hip_group_ctx(group, i), or the host's own allocator);
 * a member with counts[i] == 0 sits the call out.  Asynchronous: the call hands every shard to its member's worker thread, which
 * issues it on the member's own streams — lanes rotated, launches chained, exactly as racc_hip_intersect_device(lane =
 * RACC_HIP_LANE_AUTO, stream = NULL) — so a caller that issues batch after batch keeps all GPUs full without a host thread of its
 * own per GPU.  The arrays belong to the engine until racc_hip_group_wait returns; it waits for everything issued on every
 * member and reports the first failure (a watchdog trip included).  No exchange between the members; a GPU-side consumer that
 * needs every hit everywhere gathers afterwards (racc_hip_allgather_results, one communicator rank per member). */
int racc_hip_group_intersect_device(racc_hip_group* group, const racc_hip_group_scene* scene, const racc_hip_group_env* env,
                                    const void* const* d_rays, void* const* d_results, const uint32_t* counts);
int racc_hip_group_wait(racc_hip_group* group);

/* ---- multi-GPU: hit-record exchange over RCCL/xGMI ------------------------------------------------
 * The path shards without any exchange: rays never interact, the scene is read-only (Scene.cpp:342-346), so every GPU
 * traces its contiguous shard of a batch (one process per GPU, or one racc::Context over several GPUs).  Only a GPU-side
 * consumer that needs EVERY hit on EVERY GPU needs the step below: one ncclAllGather of the 16-byte Result records, a
 * single message per rank.  librccl is bound at run time (dlopen); these entries fail with RACC_HIP_ERR_DEVICE if it is
 * absent.  No counterpart in the reference (it drives one device, RayAccelerator.cpp:467-478).
 *   rank 0: racc_hip_comm_unique_id(&id); the host program ships the 128 bytes to the other ranks (MPI, torch.distributed,
 *   a file); every rank: racc_hip_comm_init_rank(ctx, &id, rank, nranks, &comm);
 *   per batch: racc_hip_allgather_results(comm, d_shard, d_all, rays_per_rank, stream)  — every rank contributes
 *   rays_per_rank records (pad the last shard); d_all holds nranks * rays_per_rank records, rank r's at r * rays_per_rank;
 *   in place if d_shard == d_all + rank * rays_per_rank.  Asynchronous on `stream` (hipStream_t as void*, NULL = default). */
typedef struct racc_hip_comm racc_hip_comm;
typedef struct racc_hip_comm_id { char bytes[128]; } racc_hip_comm_id;      /* = ncclUniqueId */
int racc_hip_comm_unique_id(racc_hip_comm_id* id);
int racc_hip_comm_init_rank(racc_hip_ctx* ctx, const racc_hip_comm_id* id, int rank, int nranks, racc_hip_comm** out);
int racc_hip_allgather_results(racc_hip_comm* comm, const void* d_send, void* d_recv, uint32_t count_per_rank, void* stream);
int racc_hip_comm_destroy(racc_hip_comm* comm);

/* ---- host-side scene build (no GPU needed) ---------------------------------------------------
 * ≙ the GPU branch of racc::createScene, Scene.cpp:216-339: createBvh2 (Bvh2.cpp:772-907) →
 * leaf pair merge (Scene.cpp:237-261) → 64 B node flatten (Scene.cpp:275-332) → pair padding
 * (Scene.cpp:334-338).  vertices: xyzw floats, 16-byte aligned (Scene.cpp:187); index_count % 3 == 0
 * (Scene.cpp:186).  Deterministic (the reference's node numbering is thread-timing dependent).
 * = racc_host_scene_build_ex with options NULL: the library default, RACC_HOST_BUILD_DEFAULT_QUALITY — the quality-1 tree (below; the
 * same reference-format blobs with fewer node visits per ray; it is the tree bench.py's headline figure is measured on, so a drop-in caller
 * of racc::createScene gets what is benchmarked) — unless the environment variable RACC_BUILD_QUALITY (0, 1, 2) says otherwise; 0 there, or
 * racc_host_build_options.quality = 0, gives the reference's own builder (Bvh2.cpp:257-535), byte-identical to the oracle's restatement. */
#define RACC_HOST_BUILD_DEFAULT_QUALITY 1u
int racc_host_scene_build(const float* vertices, uint32_t vertex_count,
                          const uint32_t* indices, uint32_t index_count,
                          racc_host_scene** out);
/* The same with options.  quality 0 = the reference's builder.  quality 1 (= racc_host_scene_build) / 2 (no
 * counterpart in the reference): the finished tree is post-processed — every leaf cut down to ONE triangle pair (the reference
 * leaves up to six in a leaf, and every pair of a visited leaf is tested, Kernels.h:200-205), then subtrees re-inserted where they
 * enlarge the boxes above them least (Bittner et al. 2013; parallel over fixed subtrees, so still deterministic for any thread
 * count).  The blobs stay in the reference's format (Scene.cpp:73-87) and the reference's traversal order applies unchanged: the
 * oracle and the reference's own OpenCL kernel consume them as they are; only the number of node visits and pair tests per ray
 * drops (battlefield-synth, first-bounce rays: 51.1 -> 43.9 visits, 3.44 -> 2.43 pair tests at quality 1).  Since round 6 the tree is
 * built over REFERENCES to triangles: a triangle whose box is much larger than itself gets several, each with the box of the part of it
 * between two median planes of the scene's box (spatial splits, Karras & Aila 2013; split_percent below), and then sits in several leaves,
 * its pair record written once per leaf — remap[] names a triangle more than once, pair_count grows by up to the budget.  Hit records of a
 * quality tree equal those of the quality-0 tree up to how a triangle happens to be paired (t/u/v within rounding, primId up to
 * exact-distance ties and rays that graze an edge).  threads 0 = RACC_BUILD_THREADS, else the CPUs this process may use. */
typedef struct racc_host_build_options {
    uint32_t struct_size;      /* sizeof(racc_host_build_options): fields a caller's older header lacks read as 0 */
    uint32_t quality;          /* 0, 1, 2 */
    uint32_t threads;
    uint32_t split_percent;    /* quality >= 1: spatial splits, the budget of extra triangle references as a percentage of the triangle count;
                                  0 = the library chooses: RACC_HOST_BUILD_DEFAULT_SPLIT_PERCENT, or three times that where the SAH estimate of the
                                  tree as built drops by more than 7 % with it (scenes of unconnected, overlapping triangles); RACC_BUILD_SPLIT_PERCENT
                                  overrides; RACC_HOST_BUILD_NO_SPLITS = none */
    uint32_t reserved[4];
} racc_host_build_options;
#define RACC_HOST_BUILD_DEFAULT_SPLIT_PERCENT 10u
#define RACC_HOST_BUILD_NO_SPLITS 0xFFFFFFFFu
int racc_host_scene_build_ex(const float* vertices, uint32_t vertex_count,
                             const uint32_t* indices, uint32_t index_count,
                             const racc_host_build_options* options,      /* NULL = all defaults */
                             racc_host_scene** out);
int racc_host_scene_free(racc_host_scene* scene);
/* Borrowed pointers into the build product, valid until racc_host_scene_free. */
int racc_host_scene_blobs(const racc_host_scene* scene,
                          const void** nodes64, uint32_t* node_count,
                          const void** pairs48, uint32_t* pair_count_padded, uint32_t* pair_count,
                          const uint32_t** remap, uint32_t* remap_count);
/* The intermediate BVH2 (Bvh2.h:15-29): 48 B nodes + permuted triangle ids. */
int racc_host_scene_bvh2(const racc_host_scene* scene,
                         const void** nodes48, uint32_t* node_count,
                         const uint32_t** triangles, uint32_t* triangle_count);

/* The device layout racc_hip_scene_upload gives the node blob (for tools and tests; needs no GPU).  Device record, 64 B: words 0-1 =
 * the two child references (re-numbered), words 2-3 unused, then the child boxes as six (min, max) plane pairs: Lx Ly Lz Rx Ry Rz.
 * order 1 (the default of racc_hip_scene_upload): every 128-byte line holds a node and, behind it, its inner child with the larger
 * box — a ray that goes on into that child finds its record in the line it has just fetched — lines in depth-first order; nodes whose
 * children are both leaves share lines pairwise; all-zero padding records (no reference points at them) fill the rest, so *count may
 * exceed node_count.  order 0: the 4096 nodes with the largest boxes first, the rest in the blob's order (round 1-3 layout).
 * order > 1 (what a context created with kernel_variant 60-63 uploads): the `order` nodes with the largest own boxes first — a
 * connected top of the tree, the records those kernels keep in LDS — and the subtrees below them in order 1's line pairs.
 * out64 may be NULL (then only *count is returned); capacity in records. */
int racc_host_scene_device_nodes(const void* nodes64, uint32_t node_count, uint32_t pair_count, uint32_t remap_count, int order,
                                 void* out64, uint32_t capacity, uint32_t* count);

#ifdef __cplusplus
}
#endif
#endif
