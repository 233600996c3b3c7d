// Calibration of rocprofv3's FETCH_SIZE for THIS engine's access pattern (round-4 verdict, measurement item b): the guide's "double
// FETCH_SIZE on gfx950" is calibrated for wide streaming reads only.  Here a 2 GiB table of 64-byte records (no cache holds it: 8 x 4 MiB
// of L2, 256 MiB of Infinity Cache) is read with a KNOWN number of distinct 128-byte lines, every line exactly once, through the traversal
// kernel's own fetch — quad-cooperative buffer_load_dwordx4 ... lds, four loads per 64 records (gather64.hip mode 2):
//   mode 0  streaming: every wave reads 4 KiB contiguous per iteration with per-lane dwordx4 loads            (the guide's calibrated case)
//   mode 1  gather, ONE 64 B record of every touched line (the even one); lines in hashed order                (half of every line is never asked for)
//   mode 2  gather, BOTH records of every touched line, fetched by consecutive loads of the same wave        (whole lines asked for)
//   mode 3  gather, both records of every touched line, but the second one a whole pass later                  (the line has left the caches by then)
// Every mode touches LINES distinct lines.  If the fabric moves whole 128-byte lines for a 64-byte request, modes 1 and 2 show the same
// FETCH_SIZE and take the same time, and mode 3 twice that; if it moved 64-byte sectors, mode 2 would show twice mode 1.  FETCH_SIZE x 1024 /
// LINES is what the counter tallies per line; TCC_EA0_RDREQ counts the requests.  Run plain for timings, under
// `rocprofv3 --pmc FETCH_SIZE` / `--pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum` / `--pmc TCC_MISS_sum` for the counters
// (tools/microbench/run_microbench.sh does both and writes fetchcal.json).
// Build: hipcc --offload-arch=gfx950 -O3 -o fetchcal fetchcal.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <int J> __device__ __forceinline__ unsigned quadBroadcast(unsigned v) { return __builtin_amdgcn_mov_dpp(v, J * 0x55, 0xF, 0xF, true); }

constexpr unsigned kLineBits = 24;                      // 2^24 lines of 128 B = 2 GiB
constexpr unsigned kLines = 1u << kLineBits;
constexpr unsigned kTouched = 1u << 22;                 // lines each mode touches: 4 Mi lines = 512 MiB of lines out of 2 GiB

__device__ __forceinline__ unsigned lineOf(unsigned slot) { return (slot * 2654435761u) & (kLines - 1u); }      // odd multiplier: a bijection on 2^24

template <int MODE>
__global__ void __launch_bounds__(256) fetchcal(const float4* __restrict__ recs, float* out) {
    __shared__ __attribute__((aligned(16))) unsigned char stage[4][4 * 1040];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned k = lane & 3u;
    const unsigned gwave = blockIdx.x * 4u + wave, nwaves = gridDim.x * 4u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(recs), 64, kLines * 2u, 0x00020000);
    const unsigned myStage = (lane & 3u) * 1040u + (lane >> 2) * 64u;
    float acc = 0.f;
    // wave iterations over the whole grid (64 records each): mode 1 covers 64 lines per iteration, modes 0 and 2 cover 32, mode 3 makes two passes of mode 1's
    const unsigned total = (MODE == 1) ? kTouched / 64u : (MODE == 3) ? 2u * (kTouched / 64u) : kTouched / 32u;
    for (unsigned it = gwave; it < total; it += nwaves) {
        float4 a, b, c, d;
        if (MODE == 0) {
            const float4* p = recs + (size_t(it) * 32u * 8u) + lane;          // 32 lines = 4 KiB = 256 float4: four per lane, coalesced
            a = p[0]; b = p[64]; c = p[128]; d = p[192];
        } else {
            unsigned rec;
            if (MODE == 1) rec = lineOf(it * 64u + lane) * 2u;                                   // the even record of 64 different lines
            else if (MODE == 2) rec = lineOf(it * 32u + (lane >> 1)) * 2u + (lane & 1u);           // both records of 32 lines, neighbours in the wave
            else { const unsigned pass = it / (kTouched / 64u), i2 = it % (kTouched / 64u); rec = lineOf(i2 * 64u + lane) * 2u + pass; }   // even records first, odd ones a pass later
            unsigned char* base = stage[wave];
#define STEP(j) __builtin_amdgcn_struct_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(base + j * 1040), 16, quadBroadcast<j>(rec), k * 16u, 0, 0, 0);
            STEP(0) STEP(1) STEP(2) STEP(3)
#undef STEP
            __builtin_amdgcn_s_waitcnt(0x0F70);
            const float4* mine = reinterpret_cast<const float4*>(base + myStage);
            a = mine[0]; b = mine[1]; c = mine[2]; d = mine[3];
        }
        acc += (a.x + b.y) * (c.z + d.w);
    }
    out[blockIdx.x * 256u + tid] = acc;
}

int main() {
    float4* recs; float* out;
    const size_t bytes = size_t(kLines) * 128u;
    const unsigned blocks = 256 * 5;
    if (hipMalloc(&recs, bytes) != hipSuccess || hipMalloc(&out, blocks * 256 * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(recs, 0x3c, bytes);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("table %.0f MiB, %u lines of 128 B; every mode touches %u distinct lines = %.0f MiB of lines\n", bytes / 1048576.0, kLines, kTouched, kTouched * 128.0 / 1048576.0);
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) fetchcal<0><<<blocks, 256>>>(recs, out);
            if (mode == 1) fetchcal<1><<<blocks, 256>>>(recs, out);
            if (mode == 2) fetchcal<2><<<blocks, 256>>>(recs, out);
            if (mode == 3) fetchcal<3><<<blocks, 256>>>(recs, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double asked = (mode == 1 ? 64.0 : 128.0) * kTouched;
            printf("mode %d: %.3f ms, bytes asked for %.0f MiB, lines touched %u (x128 B = %.0f MiB%s): asked %.1f GB/s, lines %.1f GB/s (err %d)\n", mode, ms, asked / 1048576.0,
                   kTouched, kTouched * 128.0 / 1048576.0, mode == 3 ? ", each fetched twice" : "", asked / ms / 1e6, kTouched * 128.0 * (mode == 3 ? 2 : 1) / ms / 1e6, int(hipGetLastError()));
        }
    }
    return 0;
}
