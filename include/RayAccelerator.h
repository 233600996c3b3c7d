// RayAccelerator.h — C++ interface of the MI355X build of the Ray Accelerator library.
//
// Source-compatible counterpart of the reference's public header (RayAccelerator/RayAccelerator.h:25-116):
// the same namespace, type names, field names, the same 11 free functions with the same argument meaning,
// ownership and error behaviour (null + one "RayAccelerator: ..." line on stderr), and byte-identical Ray /
// Result / Vertex / Color / RayStream records, so a renderer written against the reference (spawn/shade
// callbacks, RenderCallbacks, TiledRenderer-style code) compiles against this header unchanged.
//
// Deliberate differences, all forced by the platform change (Intel iGPU + OpenCL  ->  MI355X + HIP):
//   * Configuration::gpuContext is an opaque racc::GpuContext (see gpuContextForDevice) instead of a
//     cl_context (reference :33).  Null still means "no GPU", but this build has no CPU tracing mode (the
//     reference's is a binary-only Embree 2.x, not shipped), so createContext() then fails loudly.
//   * The uint8_t/uint16_t sizing fields (reference :35-41,45-47) are uint32_t here: a ≥256-thread host or a
//     1M-ray stream does not fit the reference's widths.  Assignments from existing code widen.
//   * Linux is supported (the reference #errors outside _WIN32/__APPLE__, :15-23); no <immintrin.h> needed.
// Implementation: rayaccel_amd/csrc/racc_api.cpp over the C-ABI in racc_hip.h.
#ifndef RACC_MI355X_RAYACCELERATOR_H
#define RACC_MI355X_RAYACCELERATOR_H

#include <stddef.h>
#include <stdint.h>

#if defined(_MSC_VER)
#define RACC_ALIGNED(n) __declspec(align(n))
#else
#define RACC_ALIGNED(n) __attribute__((aligned(n)))
#endif

namespace racc {

static const uint32_t invalidTriangle = ~(uint32_t)0;   // Result::triangle of a miss (reference :26)

struct Context;
struct Scene;
struct Environment;
struct GpuContextTag;
typedef GpuContextTag* GpuContext;                       // replaces cl_context (reference :33)

#define RACC_MAX_DEVICES 8                                // GPUs of one node

// Handle for GPU `ordinal` of this process (0-based; with one process per GPU pass LOCAL_RANK).
// Returns null if no gfx950 device with that ordinal exists.
GpuContext gpuContextForDevice(int ordinal);
// Handle for several GPUs driven by ONE context (the reference's cl_context picks devices[0], RayAccelerator.cpp:467-478):
// the scene and environment are replicated on each, ray streams are sharded over them as whole streams, results land in
// place — no exchange between GPUs on this path.  `ordinals` may repeat an entry (two engine contexts on one GPU: rehearsal).
GpuContext gpuContextForDevices(const int* ordinals, unsigned count);
GpuContext gpuContextForAllDevices();
// Fast traversal mode (opt-in; no counterpart in the reference; default off).  Contexts created from this handle trace with the compressed
// 4-wide kernel (racc_hip_options::kernel_variant 50: half the node bytes per ray, 6-15 % more rays per second).  What changes for the
// caller: the reported hit is still the closest hit by SURVEY.md §8(c)'s acceptance rule — the double-precision arbiter's — but where two
// triangles are hit at EXACTLY the same distance (a shared edge or vertex) `triangle` may be the other one than the reference's traversal
// order would report, and a hit the reference's own box test culls although its pair test accepts it is reported (the closer one).
// t, u, v of a reported triangle are computed by the reference's pair test, bit for bit.  The environment variable RACC_FAST_TRAVERSAL=1/0
// overrides the handle's setting for every context the process creates.
void setFastTraversal(GpuContext gpuContext, bool on);

struct Configuration {                                   // reference :32-42
    GpuContext gpuContext;
    bool allowCpuTracing;            // accepted for compatibility; ignored (no CPU tracing mode in this build)
    uint32_t cpuThreads;             // threads that run the spawn/shade callbacks
    uint32_t gpuSubmissionThreads;   // threads that submit ray streams to the GPU, one HIP stream each
    uint32_t maxRaysInFlight;
    uint32_t maxRaysPerSpawn;
    uint32_t cpuTestBatch;           // unused (CPU tracing)
    uint32_t cpuShadeBatch;
    uint32_t rayStreamBatchSize;     // rays a stream must hold before it is scheduled for intersection
};

struct ContextInfo {                                     // reference :44-49
    uint32_t threadCount;
    uint32_t rayStreamCount;
    uint32_t rayStreamSize;
    uint32_t maxRaysInFlight;
};

struct RACC_ALIGNED(16) Vertex { float x, y, z, w; };    // reference :51-53
struct RACC_ALIGNED(16) Color { float r, g, b, a; };     // reference :55-57

struct RACC_ALIGNED(32) Ray {                            // reference :59-64
    float origin[3];
    float minT;
    float dir[3];
    float maxT;
};

struct RACC_ALIGNED(16) Result {                         // reference :66-76
    uint32_t triangle;                                   // invalidTriangle => `miss` holds environment radiance
    union {
        struct { float t, u, v; } hit;                   // u = weight of the triangle's 2nd index, v = of the 3rd
        struct { float r, g, b; } miss;
    };
};

struct RayStream {                                       // reference :78-83
    uint32_t index;                                      // keys renderer-side payload arrays; results land in place
    uint32_t count;
    Ray* rays;
    Result* results;
};

struct Stats { uint64_t raysTraced; };                   // reference :85-87

struct RenderCallbacks {                                 // reference :89-93
    void* data;
    // Append up to maxRaysPerSpawn rays to *output; return false when no more primary rays exist this frame.
    bool (*spawn)(void* data, unsigned thread, RayStream* output);
    // Consume input[start,end) (rays + results) and append at most end-start follow-up rays to *output.
    void (*shade)(void* data, unsigned thread, const RayStream* input, unsigned start, unsigned end, RayStream* output);
};

static_assert(sizeof(Ray) == 32 && alignof(Ray) == 32, "Ray layout (RayAccelerator.h:59-64)");
static_assert(sizeof(Result) == 16 && alignof(Result) == 16, "Result layout (RayAccelerator.h:66-76)");
static_assert(sizeof(Vertex) == 16 && sizeof(Color) == 16, "Vertex/Color layout (RayAccelerator.h:51-57)");
static_assert(offsetof(RayStream, rays) == 8 && sizeof(RayStream) == 24, "RayStream layout (RayAccelerator.h:78-83)");

void init();                                             // reference :95   (sets FTZ/DAZ on the calling thread)
void deinit();                                           // reference :97

Configuration defaultConfiguration(GpuContext gpuContext);   // reference :99; MI355X-sized defaults

Context* createContext(Configuration configuration);     // reference :101
void destroy(Context* context);                          // reference :103
ContextInfo info(Context* context);                      // reference :105

// Copies its inputs (reference Scene.cpp:203-207,342-346).  vertices 16-byte aligned, indexCount % 3 == 0.
Scene* createScene(Context* context, const Vertex* vertices, unsigned vertexCount, const uint32_t* indices, unsigned indexCount);
void destroy(Scene* scene);                              // reference :109

Environment* createEnvironment(Context* context, const Color* colors, unsigned width, unsigned height);   // reference :111
void destroy(Environment* environment);                  // reference :113

// One frame: spawn until exhausted, intersect on the GPU, shade, repeat until no rays are in flight.
// Called from one application thread; blocks (reference RayAccelerator.cpp:738-759).
Stats render(Context* context, Scene* scene, Environment* environment, RenderCallbacks callbacks);
// Extension: null after a clean render(); otherwise why the last frame ended early (a device error; the reference ignores
// those, RayAccelerator.cpp:393-403).  The frame's remaining streams were dropped; the context stays usable.
const char* lastError(Context* context);

}  // namespace racc

#endif
