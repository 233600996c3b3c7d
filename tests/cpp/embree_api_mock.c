/*
 * embree_api_mock.c — a MOCK of the twelve Embree 3/4 entry points oracle/embree_shim.c binds, for ONE purpose: to execute the optional
 * system-Embree adapter (oracle/embree_adapter.py + embree_shim.c) end to end on machines that have no Embree — i.e. everywhere
 * this build has run.  TEST INFRASTRUCTURE; it is not Embree, pins nothing about Embree's numerics, and is never a baseline: behind the API it
 * is a double-precision Moller-Trumbore over all triangles (the oracle's arbiter, re-spelled).  What a test with it proves: the adapter's
 * plumbing — dlopen/dlsym of every symbol, version probe, geometry buffers (16-byte vertex stride, 12-byte index stride), the RTCRayHit field
 * order the shim fills and reads, the slice loop over threads, the result conversion (geomID invalid => miss; primID, tfar, u, v otherwise).
 * What it cannot prove: that a real Embree lays RTCRayHit out as the shim assumes — the layout is taken from the public rtcore_ray.h of
 * Embree 3 and 4, and this mock is written from the same reading.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct __attribute__((aligned(16))) {
    float org_x, org_y, org_z, tnear, dir_x, dir_y, dir_z, time, tfar;
    unsigned mask, id, flags;
    float Ng_x, Ng_y, Ng_z, u, v;
    unsigned primID, geomID, instID[1];
} rayhit;

typedef struct { float* vertices; size_t vstride, vcount; unsigned* indices; size_t istride, icount; int committed; } geometry;
typedef struct { geometry* geom; int committed; } scene;
static int g_devices = 0;

void* rtcNewDevice(const char* cfg) { (void)cfg; ++g_devices; return &g_devices; }
void rtcReleaseDevice(void* d) { (void)d; --g_devices; }
long rtcGetDeviceProperty(void* d, int prop) { (void)d; return prop == 1 ? 4 : 0; }      /* RTC_DEVICE_PROPERTY_VERSION_MAJOR -> "4" */
void* rtcNewScene(void* d) { (void)d; return calloc(1, sizeof(scene)); }
void rtcReleaseScene(void* s) { scene* sc = (scene*)s; if (sc && sc->geom) { free(sc->geom->vertices); free(sc->geom->indices); free(sc->geom); } free(sc); }
void* rtcNewGeometry(void* d, int type) { (void)d; return type == 0 ? calloc(1, sizeof(geometry)) : 0; }
void* rtcSetNewGeometryBuffer(void* g, int type, unsigned slot, int format, size_t stride, size_t count) {
    geometry* ge = (geometry*)g;
    if (!ge || slot != 0) return 0;
    if (type == 1 && format == 0x9003) { ge->vertices = (float*)calloc(count + 1, stride); ge->vstride = stride; ge->vcount = count; return ge->vertices; }
    if (type == 0 && format == 0x5003) { ge->indices = (unsigned*)calloc(count + 1, stride); ge->istride = stride; ge->icount = count; return ge->indices; }
    return 0;
}
void rtcCommitGeometry(void* g) { ((geometry*)g)->committed = 1; }
unsigned rtcAttachGeometry(void* s, void* g) { ((scene*)s)->geom = (geometry*)g; return 0; }
void rtcReleaseGeometry(void* g) { (void)g; }      /* the scene keeps it */
void rtcCommitScene(void* s) { ((scene*)s)->committed = 1; }

void rtcIntersect1(void* s, rayhit* rh, void* args) {
    (void)args;
    const scene* sc = (const scene*)s;
    if (!sc || !sc->committed || !sc->geom || !sc->geom->committed) return;
    const geometry* g = sc->geom;
    const double o[3] = { rh->org_x, rh->org_y, rh->org_z }, d[3] = { rh->dir_x, rh->dir_y, rh->dir_z };
    for (size_t t = 0; t < g->icount; ++t) {
        const unsigned* ix = (const unsigned*)((const char*)g->indices + t * g->istride);
        const float* a = (const float*)((const char*)g->vertices + ix[0] * g->vstride);
        const float* b = (const float*)((const char*)g->vertices + ix[1] * g->vstride);
        const float* c = (const float*)((const char*)g->vertices + ix[2] * g->vstride);
        const double e1[3] = { (double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2] }, e2[3] = { (double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2] };
        const double p[3] = { d[1] * e2[2] - d[2] * e2[1], d[2] * e2[0] - d[0] * e2[2], d[0] * e2[1] - d[1] * e2[0] };
        const double det = e1[0] * p[0] + e1[1] * p[1] + e1[2] * p[2];
        if (det == 0.0) continue;
        const double tv[3] = { o[0] - a[0], o[1] - a[1], o[2] - a[2] };
        const double u = (tv[0] * p[0] + tv[1] * p[1] + tv[2] * p[2]) / det;
        const double q[3] = { tv[1] * e1[2] - tv[2] * e1[1], tv[2] * e1[0] - tv[0] * e1[2], tv[0] * e1[1] - tv[1] * e1[0] };
        const double v = (d[0] * q[0] + d[1] * q[1] + d[2] * q[2]) / det;
        const double tt = (e2[0] * q[0] + e2[1] * q[1] + e2[2] * q[2]) / det;
        if (u < 0.0 || v < 0.0 || u + v > 1.0 || !(tt > rh->tnear) || !(tt <= rh->tfar)) continue;
        rh->tfar = (float)tt; rh->u = (float)u; rh->v = (float)v; rh->primID = (unsigned)t; rh->geomID = 0;
        rh->Ng_x = (float)(e1[1] * e2[2] - e1[2] * e2[1]); rh->Ng_y = (float)(e1[2] * e2[0] - e1[0] * e2[2]); rh->Ng_z = (float)(e1[0] * e2[1] - e1[1] * e2[0]);
    }
}
