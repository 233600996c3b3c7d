// Microbenchmark behind DESIGN.md §3 "quad-cooperative node fetch": how fast can a wave gather 64 random 64-byte records?
//   mode 0: every lane loads its own record with 4 x global_load_dwordx4 (what traverseKernelV2 does per inner step)
//   mode 1: quad-cooperative: in load j the four lanes of a quad read the 64 contiguous bytes of the record of the quad's
//           lane j (16 B each) -> registers (no redistribution; lower bound for the fetch itself)
//   mode 2: mode 1 through LDS-DMA (buffer_load_dwordx4 ... lds) + each lane reads its own record back with 4 ds_read_b128
// Build: hipcc --offload-arch=gfx950 -O3 -o gather64 gather64.hip ; run: ./gather64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int J> __device__ __forceinline__ unsigned quadBroadcast(unsigned v) {   // lane j of every quad -> its four lanes
    return __builtin_amdgcn_mov_dpp(v, J * 0x55, 0xF, 0xF, true);
}

template <int MODE, int ACTIVE = 64>
__global__ void __launch_bounds__(256) gather(const float4* __restrict__ recs, unsigned nrec, const unsigned* __restrict__ idx, unsigned iters, float* out) {
    __shared__ __attribute__((aligned(16))) unsigned char stage[4][4 * 1040];
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned k = lane & 3u;
    unsigned cur = idx[(blockIdx.x * 256u + tid) % nrec];
    float acc = 0.f;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(recs), 64, nrec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrcRaw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4*>(recs), 0, nrec * 64u, 0x00020000);
    const unsigned myStage = (lane & 3u) * 1040u + (lane >> 2) * 64u;
    for (unsigned it = 0; it < iters; ++it) {
        float4 a, b, c, d;
        a = b = c = d = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 0) {
            if (ACTIVE == 64 || ((lane * 2654435761u) >> 26) < unsigned(ACTIVE)) {      // a scattered subset of the lanes
                const float4* p = recs + size_t(cur) * 4;
                a = p[0]; b = p[1]; c = p[2]; d = p[3];
            }
        } else if (MODE == 1) {
            float4 v[4];
#define STEP(j) v[j] = recs[size_t(quadBroadcast<j>(cur)) * 4 + k];
            STEP(0) STEP(1) STEP(2) STEP(3)
#undef STEP
            a = v[0]; b = v[1]; c = v[2]; d = v[3];     // (not redistributed: only the fetch cost)
        } else if (MODE == 3) {         // per lane, raw buffer loads -> registers
            const unsigned off = cur * 64u;
            a = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrcRaw, off, 0, 0));
            b = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrcRaw, off + 16u, 0, 0));
            c = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrcRaw, off + 32u, 0, 0));
            d = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrcRaw, off + 48u, 0, 0));
        } else if (MODE == 4) {         // quad-cooperative, raw buffer loads -> registers (not redistributed)
            float4 v[4];
#define STEP(j) v[j] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrcRaw, quadBroadcast<j>(cur) * 64u + k * 16u, 0, 0));
            STEP(0) STEP(1) STEP(2) STEP(3)
#undef STEP
            a = v[0]; b = v[1]; c = v[2]; d = v[3];
        } else if (MODE == 5) {         // per lane through LDS-DMA: piece j of every lane's record lands at j*1024 + lane*16
            unsigned char* base = stage[wave];
            const unsigned off = cur * 64u;
#define STEP(j) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcRaw, (__attribute__((address_space(3))) void*)(base + j * 1024), 16, off, 0, j * 16, 0);
            STEP(0) STEP(1) STEP(2) STEP(3)
#undef STEP
            __builtin_amdgcn_s_waitcnt(0x0F70);
            const float4* mine = reinterpret_cast<const float4*>(base + lane * 16u);
            a = mine[0]; b = mine[64]; c = mine[128]; d = mine[192];
        } else if (MODE == 8 || MODE == 9) {
            // cooperative LDS-DMA with only ACTIVE of the 64 rays live.  8: scattered lanes, four loads, dead rays fetch an
            // out-of-range element; 9: the live rays compacted to the first lanes, ceil(ACTIVE / 16) loads
            unsigned char* base = stage[wave];
            const bool act = MODE == 9 ? lane < unsigned(ACTIVE) : ((lane * 2654435761u) >> 26) < unsigned(ACTIVE);
            const unsigned idxOrOut = act ? cur : 0x7FFFFFFFu;
#define STEP(j) if (MODE == 8 || j * 16 < ACTIVE) __builtin_amdgcn_struct_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(base + j * 1040), 16, quadBroadcast<j>(idxOrOut), k * 16u, 0, 0, 0);
            STEP(0) STEP(1) STEP(2) STEP(3)
#undef STEP
            __builtin_amdgcn_s_waitcnt(0x0F70);
            if (act) {
                const float4* mine = reinterpret_cast<const float4*>(base + myStage);
                a = mine[0]; b = mine[1]; c = mine[2]; d = mine[3];
            }
        } else {
            unsigned char* base = stage[wave];
#define STEP(j) __builtin_amdgcn_struct_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(base + j * 1040), 16, quadBroadcast<j>(cur), k * 16u, 0, 0, 0);
            STEP(0) STEP(1) STEP(2) STEP(3)
#undef STEP
            __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)   (gfx9 encoding: vmcnt lo [3:0], expcnt [6:4], lgkmcnt [11:8], vmcnt hi [15:14])
            const float4* mine = reinterpret_cast<const float4*>(base + myStage);
            a = mine[0]; b = mine[1]; c = mine[2]; d = mine[3];
        }
        acc += (a.x + a.y + a.z + a.w) * (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) * (d.x + d.y + d.z + d.w);
        cur = (cur * 1664525u + 1013904223u + __float_as_uint(a.x)) % nrec;      // next record depends on the data: a dependent chain like traversal
    }
    out[blockIdx.x * 256u + tid] = acc;
}

int main() {
    const unsigned nrec = 385432;          // battlefield-synth node count
    std::vector<float> h(size_t(nrec) * 16);
    for (size_t i = 0; i < h.size(); ++i) h[i] = float(i % 977) * 1e-3f;
    std::vector<unsigned> hi(nrec);
    for (unsigned i = 0; i < nrec; ++i) hi[i] = (i * 2654435761u) % nrec;
    float4* recs; unsigned* idx; float* out;
    const unsigned blocks = 256 * 6, iters = 2000;
    hipMalloc(&recs, h.size() * 4); hipMalloc(&idx, nrec * 4); hipMalloc(&out, blocks * 256 * 4);
    hipMemcpy(recs, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(idx, hi.data(), nrec * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 14; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) gather<0><<<blocks, 256>>>(recs, nrec, idx, iters, out);
            if (mode == 1) gather<1><<<blocks, 256>>>(recs, nrec, idx, iters, out);
            if (mode == 2) gather<2><<<blocks, 256>>>(recs, nrec, idx, iters, out);
            if (mode == 3) gather<3><<<blocks, 256>>>(recs, nrec, idx, iters, out);
            if (mode == 4) gather<4><<<blocks, 256>>>(recs, nrec, idx, iters, out);
            if (mode == 5) gather<5><<<blocks, 256>>>(recs, nrec, idx, iters, out);
            if (mode == 6) gather<0, 32><<<blocks, 256>>>(recs, nrec, idx, iters, out);     // mode 0 with ~half the lanes active
            if (mode == 7) gather<0, 16><<<blocks, 256>>>(recs, nrec, idx, iters, out);     // ~a quarter
            if (mode == 8) gather<8, 40><<<blocks, 256>>>(recs, nrec, idx, iters, out);     // cooperative LDS-DMA, ~40 scattered live rays, 4 loads
            if (mode == 9) gather<9, 40><<<blocks, 256>>>(recs, nrec, idx, iters, out);     // the same 40 rays compacted: 3 loads
            if (mode == 10) gather<0, 40><<<blocks, 256>>>(recs, nrec, idx, iters, out);    // per lane, ~40 live
            if (mode == 11) gather<8, 24><<<blocks, 256>>>(recs, nrec, idx, iters, out);
            if (mode == 12) gather<9, 24><<<blocks, 256>>>(recs, nrec, idx, iters, out);    // 2 loads
            if (mode == 13) gather<0, 24><<<blocks, 256>>>(recs, nrec, idx, iters, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double gathers = double(blocks) * 256 * iters;
            printf("mode %d: %.3f ms, %.2f G record-gathers/s, %.1f cycles per wave-gather per CU at 2.4 GHz (err %d)\n", mode, ms, gathers / ms / 1e6,
                   ms * 1e-3 * 2.4e9 / (double(blocks) * 4 * iters / 256), int(hipGetLastError()));
        }
    }
    return 0;
}
