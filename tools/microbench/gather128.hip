// Microbenchmark behind DESIGN.md §3 "wide nodes": is a random gather of 128-byte records (a full vector-L1 line each) served
// at a higher byte rate than a gather of 64-byte records (half a line each)?
//   mode 0: 64 B records, quad-cooperative LDS-DMA (4 loads per wave-gather) + 4 ds_read_b128        (= gather64 mode 2)
//   mode 1: 128 B records, octet-cooperative LDS-DMA (8 loads per wave-gather: in load j the eight lanes of an octet read
//           the record of the octet's lane j, 16 B each) + 8 ds_read_b128
//   mode 2: 128 B records, every lane loads its own record with 8 x global_load_dwordx4
//   mode 3: 128 B records, quad-cooperative LDS-DMA, 8 loads: load 2j / 2j+1 fetch the first / second half of the record of
//           the quad's lane j
// Each mode runs on a 24.7 MB table (Infinity Cache resident) and on a 2 MB one (L2 resident).
// Build: hipcc --offload-arch=gfx950 -O3 -o gather128 gather128.hip ; run: ./gather128
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(256) gather(const float4* __restrict__ recs, unsigned nrec, const unsigned* __restrict__ idx, unsigned iters, float* out) {
    __shared__ __attribute__((aligned(16))) unsigned char stage[4][8 * 1040];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned cur = idx[(blockIdx.x * 256u + tid) % nrec];
    float acc = 0.f;
    constexpr unsigned REC = MODE == 0 ? 64u : 128u;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(recs), REC, nrec, 0x00020000);
    unsigned char* base = stage[wave];
    for (unsigned it = 0; it < iters; ++it) {
        float4 v[8];
        for (int j = 0; j < 8; ++j) v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 0) {
            const unsigned k = lane & 3u;
            const unsigned mine = (lane & 3u) * 1040u + (lane >> 2) * 64u;
#define STEP(j) __builtin_amdgcn_struct_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(base + j * 1040), 16, __builtin_amdgcn_mov_dpp(cur, j * 0x55, 0xF, 0xF, true), k * 16u, 0, 0, 0);
            STEP(0) STEP(1) STEP(2) STEP(3)
#undef STEP
            __builtin_amdgcn_s_waitcnt(0x0F70);
            const float4* p = reinterpret_cast<const float4*>(base + mine);
            v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
        } else if (MODE == 1) {
            const unsigned k = lane & 7u;
            const unsigned mine = (lane & 7u) * 1040u + (lane >> 3) * 128u;      // piece j holds 8 records of 128 B
#define STEP(j) __builtin_amdgcn_struct_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(base + j * 1040), 16, __builtin_amdgcn_ds_bpermute(int(((lane & ~7u) + j) * 4u), int(cur)), k * 16u, 0, 0, 0);
            STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
#undef STEP
            __builtin_amdgcn_s_waitcnt(0x0F70);
            const float4* p = reinterpret_cast<const float4*>(base + mine);
            for (int j = 0; j < 8; ++j) v[j] = p[j];
        } else if (MODE == 2) {
            const float4* p = recs + size_t(cur) * 8;
            for (int j = 0; j < 8; ++j) v[j] = p[j];
        } else {
            const unsigned k = lane & 3u;
            const unsigned mine = (lane & 3u) * 2080u + (lane >> 2) * 64u;       // halves of the record 1040 B apart
#define STEP(j) __builtin_amdgcn_struct_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(base + j * 1040), 16, __builtin_amdgcn_mov_dpp(cur, (j >> 1) * 0x55, 0xF, 0xF, true), (j & 1) * 64u + k * 16u, 0, 0, 0);
            STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7)
#undef STEP
            __builtin_amdgcn_s_waitcnt(0x0F70);
            const float4* p = reinterpret_cast<const float4*>(base + mine);
            const float4* q = reinterpret_cast<const float4*>(base + mine + 1040u);
            for (int j = 0; j < 4; ++j) { v[j] = p[j]; v[4 + j] = q[j]; }
        }
        float s = 0.f;
        for (int j = 0; j < 8; ++j) s += (v[j].x + v[j].y) * (v[j].z + v[j].w);
        acc += s;
        cur = (cur * 1664525u + 1013904223u + __float_as_uint(v[0].x)) % nrec;      // next record depends on the data: a dependent chain like traversal
    }
    out[blockIdx.x * 256u + tid] = acc;
}

int main() {
    const size_t maxBytes = 385432ull * 64;          // battlefield-synth's node blob
    std::vector<float> h(maxBytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = float(i % 977) * 1e-3f;
    float4* recs; unsigned* idx; float* out;
    const unsigned blocks = 256 * 5, iters = 2000;
    hipMalloc(&recs, maxBytes); hipMalloc(&idx, 385432 * 4); hipMalloc(&out, blocks * 256 * 4);
    hipMemcpy(recs, h.data(), maxBytes, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int small = 0; small < 2; ++small) {
        const size_t bytes = small ? (2u << 20) : maxBytes;
        for (int mode = 0; mode < 4; ++mode) {
            const unsigned rec = mode == 0 ? 64u : 128u;
            const unsigned nrec = unsigned(bytes / rec);
            std::vector<unsigned> hi(nrec);
            for (unsigned i = 0; i < nrec; ++i) hi[i] = unsigned((i * 2654435761ull) % nrec);
            hipMemcpy(idx, hi.data(), nrec * 4, hipMemcpyHostToDevice);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) gather<0><<<blocks, 256>>>(recs, nrec, idx, iters, out);
                if (mode == 1) gather<1><<<blocks, 256>>>(recs, nrec, idx, iters, out);
                if (mode == 2) gather<2><<<blocks, 256>>>(recs, nrec, idx, iters, out);
                if (mode == 3) gather<3><<<blocks, 256>>>(recs, nrec, idx, iters, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double cyc = ms * 1e-3 * 2.4e9 / (double(blocks) * 4 * iters / 256);
                printf("table %5.1f MB mode %d (%3u B records): %.3f ms, %.1f cycles per wave-gather per CU at 2.4 GHz = %.1f B/clk/CU (err %d)\n",
                       bytes / 1048576.0, mode, rec, ms, cyc, 64.0 * rec / cyc, int(hipGetLastError()));
            }
        }
    }
    return 0;
}
