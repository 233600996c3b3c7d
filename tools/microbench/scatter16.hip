// Calibration of rocprofv3's WRITE_SIZE for the traversal kernel's result stores: N records of 16 bytes are written
// exactly once each, in four patterns, by a kernel that also streams a table through the L2 the way the traversal streams the scene.
//   mode 0: a wave writes 64 consecutive records at once (1 KiB contiguous per store instruction)
//   mode 1: the same records, but every lane writes its record at its own time (lane l after l * `gap` table reads): the eight
//           records of a 128-byte line reach the L2 far apart, with table traffic in between — the traversal kernel's pattern
//   mode 2: mode 1 with write-through stores (global_store_dwordx4 ... sc0 sc1)
//   mode 3: mode 0 with write-through stores
// Run under `rocprofv3 --pmc WRITE_SIZE` (and, separately, FETCH_SIZE): compulsory = N * 16 bytes.
// Build: hipcc --offload-arch=gfx950 -O3 -o scatter16 scatter16.hip ; run: ./scatter16 [records = 1048576] [gap = 64]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(256) scatter(float4* out, unsigned n, const float4* table, unsigned tableRecs, unsigned gap) {
    const unsigned lane = threadIdx.x & 63u;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        float acc = 0.f;
        unsigned cur = (i * 2654435761u) % tableRecs;
        const unsigned reads = (MODE == 1 || MODE == 2) ? lane * gap : 8u;
        for (unsigned k = 0; k < reads; ++k) {      // dependent table reads: keeps the L2 turning over and spreads the lanes in time
            const float4 t = table[cur];
            acc += t.x;
            cur = (__float_as_uint(t.y) + cur * 1664525u + 1013904223u) % tableRecs;
        }
        const f32x4 v = {__uint_as_float(i), acc, 1.f, 2.f};
        if (MODE >= 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(out + i), "v"(v) : "memory");
        else out[i] = make_float4(v.x, v.y, v.z, v.w);
    }
}

int main(int argc, char** argv) {
    const unsigned n = argc > 1 ? unsigned(atoi(argv[1])) : 1u << 20;
    const unsigned gap = argc > 2 ? unsigned(atoi(argv[2])) : 64u;
    const unsigned tableRecs = 4u << 20;      // 64 MiB: more than the eight L2s together
    float4 *out, *table;
    hipMalloc(&out, size_t(n) * 16); hipMalloc(&table, size_t(tableRecs) * 16);
    hipMemset(table, 0, size_t(tableRecs) * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 4; ++mode) {
        hipMemset(out, 0, size_t(n) * 16);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        switch (mode) {
            case 0: hipLaunchKernelGGL(scatter<0>, dim3(1024), dim3(256), 0, 0, out, n, table, tableRecs, gap); break;
            case 1: hipLaunchKernelGGL(scatter<1>, dim3(1024), dim3(256), 0, 0, out, n, table, tableRecs, gap); break;
            case 2: hipLaunchKernelGGL(scatter<2>, dim3(1024), dim3(256), 0, 0, out, n, table, tableRecs, gap); break;
            default: hipLaunchKernelGGL(scatter<3>, dim3(1024), dim3(256), 0, 0, out, n, table, tableRecs, gap); break;
        }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("mode %d: %u records x 16 B = %.1f MB compulsory, %.3f ms\n", mode, n, n * 16.0 / 1e6, ms);
    }
    return 0;
}
