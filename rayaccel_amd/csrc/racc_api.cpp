// racc_api.cpp — the racc:: C++ interface (include/RayAccelerator.h) over the C-ABI (include/racc_hip.h).
//
// Counterpart of the reference's RayAccelerator.cpp: context creation (:448-727), the ray-stream state machine
// driven by CPU worker threads (spawn :48-90, shade :92-156, worker loop :248-333) and GPU submission threads
// (:335-414), and render() (:738-759).  The callback contract is the reference's: spawn/shade run concurrently on
// worker threads with thread in [0, cpuThreads), outside the scheduler lock; results land in place, in order.
//
// Re-designed for a discrete MI355X instead of a shared-memory iGPU:
//   * a GPU thread takes EVERY stream that is ready (not one) and hands the set to racc_hip_intersect_streams,
//     i.e. one persistent-kernel launch per scheduling round: a 27k-ray launch fills 5 % of the chip and still
//     pays the latency of its longest ray;
//   * partially filled streams are flushed to the GPU only when nothing on the CPU side can still add rays
//     (the reference flushes eagerly, RayAccelerator.cpp:360-363, which suits an 8,960-work-item iGPU);
//   * ray/result arrays live in one page-locked block (≙ CL_MEM_USE_HOST_PTR, :643-644) so PCIe copies are DMA;
//   * stream ids and sizes are 32-bit (the reference's uint16 ids / sizes cap a stream at 131,070 rays);
//   * no CPU tracing mode (the reference's is binary-only Embree): allowCpuTracing is ignored.

#include "RayAccelerator.h"
#include "racc_hip.h"

#include <xmmintrin.h>
#include <pmmintrin.h>
#include <sched.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace racc {

// What a GpuContext handle points at: the GPUs of this process a context drives (replaces the reference's cl_context,
// which picks devices[0] of one platform, RayAccelerator.cpp:467-478).  Entries may repeat an ordinal (rehearsal of the
// multi-GPU flow on a one-GPU box).
struct GpuContextTag {
    uint32_t count;
    int ordinals[RACC_MAX_DEVICES];
    bool fastTraversal = false;      // setFastTraversal: contexts created from this handle run the compressed 4-wide kernel (racc_hip_options::kernel_variant 50)
};

struct Scene {
    Context* context;
    std::vector<racc_hip_scene*> devices;     // one replica per GPU of the context (the scene is read-only: Scene.cpp:342-346)
};

struct Environment {
    Context* context;
    std::vector<racc_hip_env*> devices;
};

struct Context {
    Configuration configuration;
    std::vector<racc_hip_ctx*> hips;          // one engine context per GPU
    bool failed = false;                      // a device error ended the current frame early (lastError)
    char error[512] = {0};

    std::vector<RayStream> streams;
    char* block = nullptr;           // all rays/results, 4096-aligned per array (reference :523-531,616-631)
    size_t blockBytes = 0;
    bool blockPinned = false;
    uint32_t rayStreamSize = 0;

    std::vector<uint32_t> empty, waitingToBeFilled, readyForTest, readyForShade;

    std::mutex mutex;
    // One condition variable per role (round 5).  With a single one every spawn, shade slice and launch woke all twenty threads, which then
    // queued up on the mutex: with callbacks that cost nothing that convoy was what bounded racc::render (render_check --null-callbacks).
    // CPU workers all wait for the same thing (a stream to spawn into or to shade), so one resource freed = notify_one on wakeCpu.
    std::condition_variable wakeCpu, wakeGpu, wakeRender;
    void wakeEverybody() { wakeCpu.notify_all(); wakeGpu.notify_all(); wakeRender.notify_all(); }
    uint32_t shadeWaiters = 0;       // CPU workers inside shadeRays waiting for an output stream: they wait for less than the idle workers do
    // one stream went back into the fill lists: any ONE idle worker can use it as well as another — unless somebody in the middle of a stream
    // waits for exactly that (then everybody looks: the one notify must not go to a worker that cannot use it)
    void streamFreed() { if (shadeWaiters) wakeCpu.notify_all(); else wakeCpu.notify_one(); }
    bool shouldExit = false;
    bool moreRaysExist = false;
    uint32_t raysInFlight = 0;
    uint32_t cpuBusy = 0;            // workers currently inside a callback
    uint32_t gpuBusy = 0;            // submission threads currently inside a launch
    uint64_t rayCount = 0;

    Scene* currentScene = nullptr;
    Environment* currentEnvironment = nullptr;
    RenderCallbacks currentCallbacks{};

    std::vector<std::thread> threads;

    // RACC_PROFILE=1: where the wall time of render() goes (summed over threads), printed by destroy()
    bool profile = false;
    std::atomic<uint64_t> nsSpawn{0}, nsShade{0}, nsGpu{0}, nsRender{0}, gpuLaunches{0}, gpuStreams{0}, gpuRays{0};
};

namespace {

inline uint64_t nowNs() { return uint64_t(std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count()); }

void complain(const char* what) { std::fprintf(stderr, "RayAccelerator: %s\n", what); }

void complainHip(const char* what) { std::fprintf(stderr, "RayAccelerator: %s (%s)\n", what, racc_hip_last_error()); }

void setFlushToZero() {   // reference Threading.h:78-79, RayAccelerator.cpp:419-420
    _MM_SET_FLUSH_ZERO_MODE(_MM_FLUSH_ZERO_ON);
    _MM_SET_DENORMALS_ZERO_MODE(_MM_DENORMALS_ZERO_ON);
}

// CPUs this process may actually use: the affinity mask, capped by the cgroup-v2 CPU quota (a container with a
// 16-CPU quota on a 256-thread host must not start 32 shading threads: they would only be throttled).
unsigned usableCpus() {
    unsigned n = std::thread::hardware_concurrency();
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0 && CPU_COUNT(&set) > 0) n = unsigned(CPU_COUNT(&set));
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        long long quota = 0, period = 0;
        if (std::fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) {
            const unsigned q = unsigned((quota + period - 1) / period);
            if (q && q < n) n = q;
        }
        std::fclose(f);
    }
    return n ? n : 1u;
}

uint32_t takeOutputStream(Context* c) {   // reference :60,108: a partly filled stream first, else an empty one
    uint32_t id;
    if (!c->waitingToBeFilled.empty()) { id = c->waitingToBeFilled.back(); c->waitingToBeFilled.pop_back(); }
    else { id = c->empty.back(); c->empty.pop_back(); }
    return id;
}

void putBack(Context* c, uint32_t id) {   // reference :77-82,130-135
    const uint32_t n = c->streams[id].count;
    if (n >= c->configuration.rayStreamBatchSize) c->readyForTest.push_back(id);
    else if (n == 0) c->empty.push_back(id);
    else c->waitingToBeFilled.push_back(id);
}

bool spawnBlocked(const Context* c) {
    return !c->moreRaysExist || c->raysInFlight + c->configuration.maxRaysPerSpawn > c->configuration.maxRaysInFlight;
}

// reference spawnRays, :48-90.  Called and returns with the lock held.
bool spawnRays(Context* c, std::unique_lock<std::mutex>& lock, unsigned thread) {
    if (spawnBlocked(c) || (c->waitingToBeFilled.empty() && c->empty.empty())) return false;
    const RenderCallbacks cb = c->currentCallbacks;
    const uint32_t maxRaysPerSpawn = c->configuration.maxRaysPerSpawn;
    const uint32_t id = takeOutputStream(c);
    RayStream* stream = &c->streams[id];
    c->raysInFlight += maxRaysPerSpawn;
    ++c->cpuBusy;
    lock.unlock();
    const uint32_t before = stream->count;
    const uint64_t t0 = c->profile ? nowNs() : 0;
    const bool more = cb.spawn(cb.data, thread, stream);
    if (c->profile) c->nsSpawn += nowNs() - t0;
    const uint32_t added = stream->count - before;
    lock.lock();
    --c->cpuBusy;
    c->raysInFlight -= maxRaysPerSpawn - std::min(added, maxRaysPerSpawn);
    putBack(c, id);
    if (!more) c->moreRaysExist = false;
    c->wakeGpu.notify_all();       // a stream may be ready for the GPU; cpuBusy / moreRaysExist feed its flush rule
    c->streamFreed();              // one stream is back in the fill lists (or a reservation was released): one worker can go on
    if (!more) c->wakeRender.notify_one();
    return true;
}

// reference shadeRays, :92-156.  The output stream is kept across the slices of one input stream (the reference — and rounds 1-4 — took
// and returned one under the mutex for every slice of cpuShadeBatch rays: 146,000 lock round trips and wake-ups per second at 1.2 Grays/s);
// it goes back when it has reached a full batch or cannot take another slice, and at the end.
bool shadeRays(Context* c, std::unique_lock<std::mutex>& lock, unsigned thread) {
    if (c->readyForShade.empty() || (c->waitingToBeFilled.empty() && c->empty.empty())) return false;
    const RenderCallbacks cb = c->currentCallbacks;
    const uint32_t batch = c->configuration.cpuShadeBatch;
    const uint32_t id = c->readyForShade.back();
    c->readyForShade.pop_back();
    RayStream* stream = &c->streams[id];
    ++c->cpuBusy;
    uint32_t outId = 0;
    RayStream* out = nullptr;
    uint32_t addedSinceLock = 0;
    for (uint32_t start = 0; start < stream->count; start += batch) {      // (at the top of an iteration the mutex is held iff no output stream is)
        const uint32_t end = std::min(stream->count, start + batch);
        if (!out) {
            ++c->shadeWaiters;
            while (c->waitingToBeFilled.empty() && c->empty.empty()) c->wakeCpu.wait(lock);   // never with the reference's stream count
            --c->shadeWaiters;
            outId = takeOutputStream(c);
            out = &c->streams[outId];
            lock.unlock();
        }
        const uint32_t before = out->count;
        const uint64_t t0 = c->profile ? nowNs() : 0;
        cb.shade(cb.data, thread, stream, start, end, out);
        if (c->profile) c->nsShade += nowNs() - t0;
        addedSinceLock += out->count - before;
        // go on shading into `out`, without the mutex, while slices are left, it can take another whole one and has not reached a full batch
        if (end < stream->count && out->count < c->configuration.rayStreamBatchSize && out->count + batch <= c->rayStreamSize) continue;
        lock.lock();
        c->raysInFlight += addedSinceLock;      // (while they were not counted this thread was counted in cpuBusy: render() cannot have finished)
        addedSinceLock = 0;
        putBack(c, outId);
        out = nullptr;
        c->wakeGpu.notify_all();
        c->streamFreed();
    }
    c->raysInFlight -= stream->count;
    stream->count = 0;
    c->empty.push_back(id);
    --c->cpuBusy;
    c->wakeCpu.notify_all();       // a stream is empty again and rays have left the flight count: several workers may be able to spawn
    c->wakeGpu.notify_all();       // cpuBusy feeds the flush rule
    c->wakeRender.notify_one();
    return true;
}

bool finished(const Context* c) {
    return c->shouldExit && !c->moreRaysExist && c->empty.size() == c->streams.size();
}

void cpuWorker(Context* c, unsigned thread) {   // reference cpuWorkerThread (GPU-context branch), :272-305
    setFlushToZero();
    std::unique_lock<std::mutex> lock(c->mutex);
    for (;;) {
        if (finished(c)) break;
        if (c->currentCallbacks.spawn && spawnRays(c, lock, thread)) continue;
        if (c->currentCallbacks.shade && shadeRays(c, lock, thread)) continue;
        c->wakeCpu.wait(lock);
    }
}

void gpuWorker(Context* c, unsigned worker) {   // reference gpuWorkerThread, :335-414
    // submission thread i drives GPU i % nGPUs through that engine context's lane i / nGPUs: ray streams are sharded over the
    // GPUs as whole streams (rays never interact), results land in place in the page-locked block
    const unsigned nDev = unsigned(c->hips.size());
    const unsigned dev = worker % nDev, lane = worker / nDev;
    std::vector<uint32_t> ids;
    std::vector<const void*> rays;
    std::vector<void*> results;
    std::vector<uint32_t> counts;
    std::unique_lock<std::mutex> lock(c->mutex);
    for (;;) {
        if (finished(c)) break;
        ids.clear();
        if (!c->readyForTest.empty()) {
            if (nDev == 1) ids.swap(c->readyForTest);
            else {          // leave the other GPUs their share of what is ready
                const size_t take = (c->readyForTest.size() + nDev - 1) / nDev;
                ids.assign(c->readyForTest.end() - take, c->readyForTest.end());
                c->readyForTest.resize(c->readyForTest.size() - take);
            }
        } else if (!c->waitingToBeFilled.empty() && c->cpuBusy == 0 && c->readyForShade.empty() && spawnBlocked(c)) {
            // Nothing on the CPU side can add rays any more: flush the partly filled streams — but never ALL streams that can still take
            // rays.  Shading a traced stream needs an output stream (reference :108); the reference's stream count (:518-521) guarantees one
            // because a stream that has left the fill lists holds a full batch, and the rays in flight are bounded.  A flushed stream holds
            // fewer, so that argument no longer counts them: with callbacks that cost nothing, every one of the streams ended up traced and
            // waiting to be shaded with no stream left to shade into (round 5, render_check --null-callbacks: the frame hung once in ~8 runs).
            // So the flush leaves `keep` streams behind — the emptiest ones — whenever the empty list cannot provide them; they go out with a
            // later flush, once the flushed ones have been shaded and have come back empty.
            const size_t keep = std::max<size_t>(1, std::min<size_t>(c->configuration.cpuThreads, c->streams.size() / 4));
            const size_t spare = c->empty.size() >= keep ? 0 : keep - c->empty.size();
            if (spare >= c->waitingToBeFilled.size()) {      // (all of them are needed as outputs: whoever is being traced or shaded now will free streams)
                c->wakeGpu.wait(lock);
                continue;
            }
            std::sort(c->waitingToBeFilled.begin(), c->waitingToBeFilled.end(), [c](uint32_t a, uint32_t b) { return c->streams[a].count < c->streams[b].count; });
            ids.assign(c->waitingToBeFilled.begin() + spare, c->waitingToBeFilled.end());
            c->waitingToBeFilled.resize(spare);
        } else {
            c->wakeGpu.wait(lock);
            continue;
        }
        rays.clear(); results.clear(); counts.clear();
        uint64_t total = 0;
        for (uint32_t id : ids) {
            rays.push_back(c->streams[id].rays);
            results.push_back(c->streams[id].results);
            counts.push_back(c->streams[id].count);
            total += c->streams[id].count;
        }
        c->rayCount += total;                            // Stats.raysTraced (reference :372)
        ++c->gpuBusy;
        Scene* scene = c->currentScene;
        Environment* env = c->currentEnvironment;
        lock.unlock();
        const uint64_t t0 = c->profile ? nowNs() : 0;
        const int rc = racc_hip_intersect_streams(c->hips[dev], scene->devices[dev], env ? env->devices[dev] : nullptr, uint32_t(ids.size()),
                                                  rays.data(), results.data(), counts.data(), lane);
        if (c->profile) { c->nsGpu += nowNs() - t0; ++c->gpuLaunches; c->gpuStreams += ids.size(); c->gpuRays += total; }
        lock.lock();
        --c->gpuBusy;
        if (rc != RACC_HIP_OK) {
            // The reference ignores device errors here (:393-403).  We end the frame: these streams' results are not valid, so
            // they are dropped (never shaded), no new rays are spawned, what is already traced drains normally, and render()
            // returns with lastError(context) set.  The process, the context and the other GPUs stay usable.
            if (!c->failed) {
                c->failed = true;
                std::snprintf(c->error, sizeof(c->error), "GPU intersection failed on device entry %u (%s)", dev, racc_hip_last_error());
                std::fprintf(stderr, "RayAccelerator: %s\n", c->error);
            }
            c->moreRaysExist = false;
            for (uint32_t id : ids) {
                c->raysInFlight -= std::min(c->raysInFlight, c->streams[id].count);
                c->streams[id].count = 0;
                c->empty.push_back(id);
            }
        } else {
            for (uint32_t id : ids) c->readyForShade.push_back(id);
        }
        c->wakeCpu.notify_all();       // several streams to shade (or, after a failure, to spawn into)
        c->wakeGpu.notify_all();       // (a failure changes moreRaysExist: the flush rule)
        c->wakeRender.notify_one();
    }
}

}  // namespace

GpuContext gpuContextForDevices(const int* ordinals, unsigned count) {
    int n = 0;
    if (!ordinals || !count || count > RACC_MAX_DEVICES || racc_hip_device_count(&n) != RACC_HIP_OK) return nullptr;
    GpuContextTag* tag = new (std::nothrow) GpuContextTag();     // lives until the process ends (a handle, like a cl_context the app keeps)
    if (!tag) return nullptr;
    tag->count = count;
    for (unsigned i = 0; i < count; ++i) {
        if (ordinals[i] < 0 || ordinals[i] >= n) { delete tag; return nullptr; }
        tag->ordinals[i] = ordinals[i];
    }
    return tag;
}

GpuContext gpuContextForDevice(int ordinal) { return gpuContextForDevices(&ordinal, 1); }

void setFastTraversal(GpuContext gpuContext, bool on) { if (gpuContext) gpuContext->fastTraversal = on; }

GpuContext gpuContextForAllDevices() {
    int n = 0;
    if (racc_hip_device_count(&n) != RACC_HIP_OK || n <= 0) return nullptr;
    int ordinals[RACC_MAX_DEVICES];
    const unsigned count = unsigned(std::min(n, int(RACC_MAX_DEVICES)));
    for (unsigned i = 0; i < count; ++i) ordinals[i] = int(i);
    return gpuContextForDevices(ordinals, count);
}

void init() { setFlushToZero(); }   // reference :417-423 (rtcInit has no counterpart)

void deinit() {}                    // reference :425-427

Configuration defaultConfiguration(GpuContext gpuContext) {   // reference :429-446, re-sized for MI355X
    Configuration cfg{};
    cfg.gpuContext = gpuContext;
    cfg.allowCpuTracing = false;
    const unsigned hw = usableCpus();
    cfg.cpuThreads = std::min(hw > 2 ? hw - 2 : 1u, 32u);   // callbacks only; leave room for the submission threads
    cfg.gpuSubmissionThreads = 4 * (gpuContext ? gpuContext->count : 1u);   // per GPU: copy-in, kernel and copy-out of consecutive launches in flight (racc_hostpath.inc; the reference's own default is 4, RayAccelerator.cpp:436)
    cfg.maxRaysInFlight = 8u << 20;                         // reference 262,144 = 29 x its iGPU's 8,960 lanes.  Here a launch spends ~2 ms between spawn and shade (PCIe both ways, four of them in flight): at the link's 1.7 Grays/s that is 3.4 M rays before a stream comes back — 4 M starved the spawners (render_check --null-callbacks: 1.08-1.17 Grays/s, 8 M: 1.21-1.25; 700 MB of page-locked streams)
    cfg.maxRaysPerSpawn = 128 * 128;
    cfg.cpuTestBatch = 1024;
    cfg.cpuShadeBatch = 8 * 1024;
    cfg.rayStreamBatchSize = 128 * 1024;
    return cfg;
}

Context* createContext(Configuration cfg) {
    if (!cfg.gpuContext) {
        complain("no GPU context given and this build has no CPU tracing mode.");
        return nullptr;
    }
    if (!cfg.cpuThreads || !cfg.gpuSubmissionThreads || !cfg.rayStreamBatchSize || !cfg.maxRaysPerSpawn || !cfg.cpuShadeBatch) {
        complain("invalid configuration (zero threads or batch sizes).");
        return nullptr;
    }
    const unsigned nDev = cfg.gpuContext->count;
    if (cfg.gpuSubmissionThreads < nDev) cfg.gpuSubmissionThreads = nDev;                  // every GPU gets a submission thread
    if (cfg.gpuSubmissionThreads > RACC_HIP_MAX_LANES * nDev) cfg.gpuSubmissionThreads = RACC_HIP_MAX_LANES * nDev;
    Context* c = new (std::nothrow) Context();
    if (!c) { complain("Unable to allocate memory."); return nullptr; }
    c->configuration = cfg;
    c->profile = std::getenv("RACC_PROFILE") != nullptr;

    racc_hip_options opts{};
    opts.struct_size = sizeof(opts);
    opts.lanes = (cfg.gpuSubmissionThreads + nDev - 1) / nDev;
    {   // fast mode (opt-in, racc::setFastTraversal or RACC_FAST_TRAVERSAL=1): the compressed 4-wide kernel.  Default: the bit-exact one.
        const char* e = std::getenv("RACC_FAST_TRAVERSAL");
        if (e ? std::atoi(e) != 0 : cfg.gpuContext->fastTraversal) opts.kernel_variant = 50;
    }
    for (unsigned d = 0; d < nDev; ++d) {
        racc_hip_ctx* hip = nullptr;
        if (racc_hip_create(cfg.gpuContext->ordinals[d], &opts, &hip) != RACC_HIP_OK) {
            complainHip("Cannot create the GPU context");
            for (racc_hip_ctx* h : c->hips) racc_hip_destroy(h);
            delete c;
            return nullptr;
        }
        c->hips.push_back(hip);
    }

    // Stream count and size: reference :517-521.
    const uint32_t inFlight = cfg.gpuSubmissionThreads + cfg.cpuThreads * 2;
    c->rayStreamSize = cfg.rayStreamBatchSize + std::max(cfg.maxRaysPerSpawn, cfg.cpuShadeBatch);
    const uint32_t streamCount = inFlight + (cfg.maxRaysInFlight + cfg.rayStreamBatchSize - 1) / cfg.rayStreamBatchSize;
    auto up = [](size_t v) { return (v + 4095) & ~size_t(4095); };
    const size_t perStream = up(sizeof(Ray) * size_t(c->rayStreamSize)) + up(sizeof(Result) * size_t(c->rayStreamSize));
    c->blockBytes = perStream * streamCount;
    if (posix_memalign(reinterpret_cast<void**>(&c->block), 4096, c->blockBytes) != 0) {
        complain("Unable to allocate memory.");
        for (racc_hip_ctx* h : c->hips) racc_hip_destroy(h);
        delete c;
        return nullptr;
    }
    std::memset(c->block, 0, c->blockBytes);   // ≙ the clear kernel, reference :647-698
    // Page-lock the whole block once (one portable hipHostRegister over all streams: every GPU can DMA from/to it).
    c->blockPinned = racc_hip_register_host(c->hips[0], c->block, c->blockBytes) == RACC_HIP_OK;
    if (!c->blockPinned) complainHip("warning: ray streams are not page-locked; PCIe copies will be staged");

    c->streams.resize(streamCount);
    size_t off = 0;
    for (uint32_t i = 0; i < streamCount; ++i) {
        RayStream& s = c->streams[i];
        s.index = i;
        s.count = 0;
        s.rays = reinterpret_cast<Ray*>(c->block + off);
        off += up(sizeof(Ray) * size_t(c->rayStreamSize));
        s.results = reinterpret_cast<Result*>(c->block + off);
        off += up(sizeof(Result) * size_t(c->rayStreamSize));
        c->empty.push_back(i);
    }
    for (uint32_t i = 0; i < cfg.cpuThreads; ++i) c->threads.emplace_back(cpuWorker, c, i);
    for (uint32_t i = 0; i < cfg.gpuSubmissionThreads; ++i) c->threads.emplace_back(gpuWorker, c, i);
    return c;
}

void destroy(Context* c) {   // reference :761-788
    if (!c) return;
    {
        std::lock_guard<std::mutex> lock(c->mutex);
        c->shouldExit = true;
    }
    c->wakeEverybody();
    for (std::thread& t : c->threads) t.join();
    if (c->profile) {
        const double wall = double(c->nsRender.load()) * 1e-9;
        std::fprintf(stderr, "RayAccelerator profile: render %.3f s | spawn %.3f + shade %.3f thread-s over %u cpu threads (%.0f %% busy) | "
                             "gpu path %.3f thread-s over %u lanes (%.0f %% busy), %llu launches, %.1f streams and %.0f rays per launch\n",
                     wall, double(c->nsSpawn.load()) * 1e-9, double(c->nsShade.load()) * 1e-9, c->configuration.cpuThreads,
                     wall > 0 ? 100.0 * double(c->nsSpawn.load() + c->nsShade.load()) * 1e-9 / (wall * c->configuration.cpuThreads) : 0.0,
                     double(c->nsGpu.load()) * 1e-9, c->configuration.gpuSubmissionThreads,
                     wall > 0 ? 100.0 * double(c->nsGpu.load()) * 1e-9 / (wall * c->configuration.gpuSubmissionThreads) : 0.0,
                     (unsigned long long)c->gpuLaunches.load(), c->gpuLaunches ? double(c->gpuStreams.load()) / double(c->gpuLaunches.load()) : 0.0,
                     c->gpuLaunches ? double(c->gpuRays.load()) / double(c->gpuLaunches.load()) : 0.0);
    }
    if (c->blockPinned) racc_hip_unregister_host(c->hips[0], c->block);
    for (racc_hip_ctx* h : c->hips) racc_hip_destroy(h);
    std::free(c->block);
    delete c;
}

ContextInfo info(Context* c) {   // reference :729-736
    ContextInfo i{};
    i.threadCount = c->configuration.cpuThreads;
    i.rayStreamCount = uint32_t(c->streams.size());
    i.rayStreamSize = c->rayStreamSize;
    i.maxRaysInFlight = c->configuration.maxRaysInFlight;
    return i;
}

Scene* createScene(Context* c, const Vertex* vertices, unsigned vertexCount, const uint32_t* indices, unsigned indexCount) {
    if (!c || !vertices || !indices) { complain("createScene: null argument."); return nullptr; }
    racc_host_scene* host = nullptr;
    if (racc_host_scene_build(&vertices->x, vertexCount, indices, indexCount, &host) != RACC_HIP_OK) {
        complainHip("Cannot build the scene");
        return nullptr;
    }
    const void *nodes = nullptr, *pairs = nullptr;
    const uint32_t* remap = nullptr;
    uint32_t nNodes = 0, nPairsPadded = 0, nPairs = 0, nRemap = 0;
    racc_host_scene_blobs(host, &nodes, &nNodes, &pairs, &nPairsPadded, &nPairs, &remap, &nRemap);
    Scene* s = new (std::nothrow) Scene{c, {}};
    if (!s) { racc_host_scene_free(host); complain("Unable to allocate memory."); return nullptr; }
    for (racc_hip_ctx* hip : c->hips) {          // one replica per GPU
        racc_hip_scene* dev = nullptr;
        if (racc_hip_scene_upload(hip, nodes, nNodes, pairs, nPairsPadded, remap, nRemap, &dev) != RACC_HIP_OK) {
            complainHip("Cannot upload the scene");
            racc_host_scene_free(host);
            destroy(s);
            return nullptr;
        }
        s->devices.push_back(dev);
    }
    racc_host_scene_free(host);
    return s;
}

void destroy(Scene* s) {   // reference Scene.cpp:359-372
    if (!s) return;
    for (size_t d = 0; d < s->devices.size(); ++d) racc_hip_scene_free(s->context->hips[d], s->devices[d]);
    delete s;
}

Environment* createEnvironment(Context* c, const Color* colors, unsigned width, unsigned height) {
    if (!c || !colors) { complain("createEnvironment: null argument."); return nullptr; }
    Environment* e = new (std::nothrow) Environment{c, {}};
    if (!e) { complain("Unable to allocate memory."); return nullptr; }
    for (racc_hip_ctx* hip : c->hips) {
        racc_hip_env* dev = nullptr;
        if (racc_hip_env_upload(hip, &colors->r, width, height, &dev) != RACC_HIP_OK) {
            complainHip("Cannot upload the environment");
            destroy(e);
            return nullptr;
        }
        e->devices.push_back(dev);
    }
    return e;
}

void destroy(Environment* e) {   // reference Environment.cpp:62-67
    if (!e) return;
    for (size_t d = 0; d < e->devices.size(); ++d) racc_hip_env_free(e->context->hips[d], e->devices[d]);
    delete e;
}

const char* lastError(Context* c) { return c && c->failed ? c->error : nullptr; }

Stats render(Context* c, Scene* scene, Environment* environment, RenderCallbacks callbacks) {   // reference :738-759
    if (!c || !scene || !callbacks.spawn || !callbacks.shade) {      // (the reference would hang or crash on these)
        complain("render: null context, scene or callback.");
        return Stats{};
    }
    const uint64_t t0 = c->profile ? nowNs() : 0;
    std::unique_lock<std::mutex> lock(c->mutex);
    c->failed = false;
    c->currentScene = scene;
    c->currentEnvironment = environment;
    c->currentCallbacks = callbacks;
    c->moreRaysExist = true;
    c->wakeEverybody();
    auto done = [c] { return !c->moreRaysExist && c->raysInFlight == 0 && c->cpuBusy == 0 && c->gpuBusy == 0; };
    // RACC_RENDER_WATCHDOG_S=<seconds> (diagnostics): a frame that makes no progress for that long prints the scheduler's state and aborts
    static const double limit = [] { const char* e = std::getenv("RACC_RENDER_WATCHDOG_S"); return e ? std::atof(e) : 0.0; }();
    if (limit > 0.0) {
        uint64_t lastCount = ~0ull;
        while (!c->wakeRender.wait_for(lock, std::chrono::duration<double>(limit), done)) {
            const uint64_t progress = c->rayCount * 1024u + c->raysInFlight % 1024u + c->readyForShade.size();
            if (progress == lastCount) {
                std::fprintf(stderr, "RayAccelerator watchdog: no progress for %.0f s: moreRaysExist %d raysInFlight %u cpuBusy %u gpuBusy %u | streams %zu: empty %zu, waitingToBeFilled %zu, "
                                     "readyForTest %zu, readyForShade %zu | traced so far %llu\n", limit, int(c->moreRaysExist), c->raysInFlight, c->cpuBusy, c->gpuBusy, c->streams.size(),
                             c->empty.size(), c->waitingToBeFilled.size(), c->readyForTest.size(), c->readyForShade.size(), (unsigned long long)c->rayCount);
                std::abort();
            }
            lastCount = progress;
        }
    } else {
        c->wakeRender.wait(lock, done);
    }
    Stats stats{};
    stats.raysTraced = c->rayCount;
    c->rayCount = 0;
    if (c->profile) c->nsRender += nowNs() - t0;
    return stats;
}

}  // namespace racc
