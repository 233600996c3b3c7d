/*
 * embree_shim.c — OPTIONAL second opinion / CPU baseline: the reference's CPU intersector is three lines of glue around
 * Intel Embree (RayAccelerator/Scene.cpp:198-213 scene build, :386-428 rtcIntersect8 + result transposition, :443-483 the
 * scalar tail).  Embree is a binary-only dependency that exists on neither box of this build, so this shim binds a SYSTEM
 * Embree 3.x / 4.x at run time (dlopen + dlsym: no Embree header or link dependency) when oracle/embree_adapter.py finds
 * one, and is never used otherwise.  TEST INFRASTRUCTURE ONLY (same rule as the rest of oracle/).
 *
 * The reference was written against Embree 2.x (rtcIntersect8 on an RTCRay8 SoA, Scene.cpp:386-416); Embree 3/4 moved to
 * RTCRayHit, so the glue here is its present-day spelling: one static triangle mesh, rtcIntersect1 per ray,
 * result = (primID, tfar, u, v), primID == RTC_INVALID_GEOMETRY_ID => miss (Scene.cpp:418-440).  Rays are processed in
 * slices of 1024 (cpuTestBatch, RayAccelerator.cpp:438) by `threads` pthreads, as the reference's CPU workers do.
 *
 * No Embree exists on any box of this build.  Its plumbing is executed end to end against an API-shaped mock (tests/cpp/embree_api_mock.c,
 * tests/test_oracle.py: every symbol, the RTCRayHit fields, the slice loop, the result conversion); against a real Embree the day a box has one.
 */
#include <dlfcn.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float origin[3], minT, dir[3], maxT; } shim_ray;           /* RayAccelerator.h:59-64 */
typedef struct { uint32_t triangle; float t, u, v; } shim_result;           /* RayAccelerator.h:66-76 */

/* Embree 3/4 RTCRayHit (rtcore_ray.h): identical layout in both major versions */
typedef struct __attribute__((aligned(16))) {
    float org_x, org_y, org_z, tnear, dir_x, dir_y, dir_z, time, tfar;
    unsigned mask, id, flags;
    float Ng_x, Ng_y, Ng_z, u, v;
    unsigned primID, geomID, instID[1];
} shim_rayhit;
/* Embree 3 RTCIntersectContext (rtcore_common.h) */
typedef struct { unsigned flags; void* filter; unsigned instID[1]; } shim_context3;

typedef struct {
    void* lib; int major;
    void* device; void* scene; void* geom;
    void* (*newDevice)(const char*);
    void (*releaseDevice)(void*);
    long (*getDeviceProperty)(void*, int);
    void* (*newScene)(void*);
    void (*releaseScene)(void*);
    void* (*newGeometry)(void*, int);
    void* (*setNewGeometryBuffer)(void*, int, unsigned, int, size_t, size_t);
    void (*commitGeometry)(void*);
    unsigned (*attachGeometry)(void*, void*);
    void (*releaseGeometry)(void*);
    void (*commitScene)(void*);
    void (*intersect1_v3)(void*, shim_context3*, shim_rayhit*);
    void (*intersect1_v4)(void*, shim_rayhit*, void*);
} shim;

#define SYM(field, name) do { *(void**)(&s->field) = dlsym(s->lib, name); if (!s->field) { shim_close(s); return 0; } } while (0)

void shim_close(shim* s) {
    if (!s) return;
    if (s->scene && s->releaseScene) s->releaseScene(s->scene);
    if (s->device && s->releaseDevice) s->releaseDevice(s->device);
    if (s->lib) dlclose(s->lib);
    free(s);
}

/* vertices: xyzw floats (RayAccelerator.h:51-53), indices: 3 per triangle.  Returns 0 on any failure. */
shim* shim_open(const char* library, const float* vertices, uint32_t vertexCount, const uint32_t* indices, uint32_t triangleCount) {
    shim* s = (shim*)calloc(1, sizeof(shim));
    if (!s) return 0;
    s->lib = dlopen(library, RTLD_NOW | RTLD_LOCAL);
    if (!s->lib) { free(s); return 0; }
    SYM(newDevice, "rtcNewDevice"); SYM(releaseDevice, "rtcReleaseDevice"); SYM(getDeviceProperty, "rtcGetDeviceProperty");
    SYM(newScene, "rtcNewScene"); SYM(releaseScene, "rtcReleaseScene"); SYM(newGeometry, "rtcNewGeometry");
    SYM(setNewGeometryBuffer, "rtcSetNewGeometryBuffer"); SYM(commitGeometry, "rtcCommitGeometry");
    SYM(attachGeometry, "rtcAttachGeometry"); SYM(releaseGeometry, "rtcReleaseGeometry"); SYM(commitScene, "rtcCommitScene");
    s->device = s->newDevice("threads=1");              /* our own pthreads drive the slices (the reference: isa=avx2,accel=bvh8.triangle4, RayAccelerator.cpp:422) */
    if (!s->device) { shim_close(s); return 0; }
    s->major = (int)s->getDeviceProperty(s->device, 1 /* RTC_DEVICE_PROPERTY_VERSION_MAJOR */);
    if (s->major == 3) *(void**)(&s->intersect1_v3) = dlsym(s->lib, "rtcIntersect1");
    else if (s->major >= 4) *(void**)(&s->intersect1_v4) = dlsym(s->lib, "rtcIntersect1");
    if (!s->intersect1_v3 && !s->intersect1_v4) { shim_close(s); return 0; }
    s->scene = s->newScene(s->device);
    void* g = s->newGeometry(s->device, 0 /* RTC_GEOMETRY_TYPE_TRIANGLE */);
    float* vb = (float*)s->setNewGeometryBuffer(g, 1 /* RTC_BUFFER_TYPE_VERTEX */, 0, 0x9003 /* RTC_FORMAT_FLOAT3 */, 16, vertexCount);
    unsigned* ib = (unsigned*)s->setNewGeometryBuffer(g, 0 /* RTC_BUFFER_TYPE_INDEX */, 0, 0x5003 /* RTC_FORMAT_UINT3 */, 12, triangleCount);
    if (!vb || !ib) { s->releaseGeometry(g); shim_close(s); return 0; }
    memcpy(vb, vertices, (size_t)vertexCount * 16);     /* Scene.cpp:203 */
    memcpy(ib, indices, (size_t)triangleCount * 12);    /* Scene.cpp:207 */
    s->commitGeometry(g);
    s->attachGeometry(s->scene, g);
    s->releaseGeometry(g);
    s->commitScene(s->scene);                           /* Scene.cpp:213 */
    return s;
}

static void trace_range(shim* s, const shim_ray* rays, shim_result* out, uint32_t b, uint32_t e) {
    for (uint32_t i = b; i < e; ++i) {
        shim_rayhit rh;
        memset(&rh, 0, sizeof(rh));
        rh.org_x = rays[i].origin[0]; rh.org_y = rays[i].origin[1]; rh.org_z = rays[i].origin[2]; rh.tnear = rays[i].minT;
        rh.dir_x = rays[i].dir[0]; rh.dir_y = rays[i].dir[1]; rh.dir_z = rays[i].dir[2]; rh.tfar = rays[i].maxT;
        rh.mask = 0xFFFFFFFFu; rh.geomID = 0xFFFFFFFFu; rh.primID = 0xFFFFFFFFu; rh.instID[0] = 0xFFFFFFFFu;
        if (s->intersect1_v3) { shim_context3 c; memset(&c, 0, sizeof(c)); c.instID[0] = 0xFFFFFFFFu; s->intersect1_v3(s->scene, &c, &rh); }
        else s->intersect1_v4(s->scene, &rh, 0);
        out[i].triangle = rh.geomID == 0xFFFFFFFFu ? 0xFFFFFFFFu : rh.primID;      /* Scene.cpp:418-428 */
        out[i].t = rh.geomID == 0xFFFFFFFFu ? 0.0f : rh.tfar;                       /* (miss colour is the environment's job) */
        out[i].u = rh.geomID == 0xFFFFFFFFu ? 0.0f : rh.u;
        out[i].v = rh.geomID == 0xFFFFFFFFu ? 0.0f : rh.v;
    }
}

typedef struct { shim* s; const shim_ray* rays; shim_result* out; uint32_t count; uint64_t* cursor; } job;

static void* worker(void* a) {
    job* j = (job*)a;
    for (;;) {
        const uint64_t k = __atomic_fetch_add(j->cursor, 1, __ATOMIC_RELAXED);
        const uint64_t b = k * 1024u;                   /* cpuTestBatch, RayAccelerator.cpp:438 */
        if (b >= j->count) break;
        trace_range(j->s, j->rays, j->out, (uint32_t)b, (uint32_t)(b + 1024u < j->count ? b + 1024u : j->count));
    }
    return 0;
}

void shim_trace(shim* s, const shim_ray* rays, shim_result* out, uint32_t count, uint32_t threads) {
    uint64_t cursor = 0;
    job j = { s, rays, out, count, &cursor };
    if (threads < 1) threads = 1;
    pthread_t* t = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    for (uint32_t i = 1; i < threads; ++i) pthread_create(&t[i], 0, worker, &j);
    worker(&j);
    for (uint32_t i = 1; i < threads; ++i) pthread_join(t[i], 0);
    free(t);
}

int shim_version(const shim* s) { return s ? s->major : 0; }
