// pcie.hip — what the host link gives the host-buffer path: page-locked H2D and D2H copies alone and together (PCIe is full duplex),
// on one or two streams per direction, per chunk size.  Output: GB/s per direction.   hipcc --offload-arch=gfx950 -O3 -o pcie pcie.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const size_t total = size_t(256) << 20;      // per direction per pass
    char *hin, *hout, *din, *dout;
    CK(hipHostMalloc(reinterpret_cast<void**>(&hin), total, hipHostMallocDefault));
    CK(hipHostMalloc(reinterpret_cast<void**>(&hout), total, hipHostMallocDefault));
    CK(hipMalloc(reinterpret_cast<void**>(&din), total)); CK(hipMalloc(reinterpret_cast<void**>(&dout), total));
    for (size_t i = 0; i < total; i += 4096) hin[i] = char(i);
    hipStream_t s[4];
    for (auto& x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    const size_t chunks[] = {size_t(1) << 20, size_t(4) << 20, size_t(8) << 20, size_t(32) << 20};
    for (size_t chunk : chunks)
        for (int mode = 0; mode < 5; ++mode) {      // 0 H2D one stream, 1 D2H one stream, 2 both one stream each, 3 both two streams each, 4 H2D two streams
            double best = 0, bestIn = 0, bestOut = 0;
            for (int rep = 0; rep < 4; ++rep) {
                CK(hipDeviceSynchronize());
                const auto t0 = std::chrono::steady_clock::now();
                size_t k = 0;
                for (size_t off = 0; off < total; off += chunk, ++k) {
                    if (mode == 0 || mode == 2) CK(hipMemcpyAsync(din + off, hin + off, chunk, hipMemcpyHostToDevice, s[0]));
                    if (mode == 3 || mode == 4) CK(hipMemcpyAsync(din + off, hin + off, chunk, hipMemcpyHostToDevice, s[k & 1]));
                    if (mode == 1 || mode == 2) CK(hipMemcpyAsync(hout + off, dout + off, chunk, hipMemcpyDeviceToHost, s[2]));
                    if (mode == 3) CK(hipMemcpyAsync(hout + off, dout + off, chunk, hipMemcpyDeviceToHost, s[2 + (k & 1)]));
                }
                CK(hipDeviceSynchronize());
                const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                const double in = (mode == 1 ? 0.0 : double(total)) / sec / 1e9, out = (mode == 0 || mode == 4 ? 0.0 : double(total)) / sec / 1e9;
                if (in + out > best) { best = in + out; bestIn = in; bestOut = out; }
            }
            std::printf("chunk %3zu MiB mode %d: H2D %.1f GB/s  D2H %.1f GB/s\n", chunk >> 20, mode, bestIn, bestOut);
        }
    return 0;
}
