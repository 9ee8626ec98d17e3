// How fast does page-locked host memory fill over PCIe when the same bytes travel as 1, 2 or 4 concurrent hipMemcpyAsync (one stream each:
// the runtime hands them to its SDMA engines)?  hipcc --offload-arch=gfx950 -O2 -o /tmp/d2h_streams_probe tools/d2h_streams_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const size_t bytes = size_t{1} << 30;
    void *d = nullptr, *h = nullptr;
    CK(hipMalloc(&d, bytes));
    CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    CK(hipMemset(d, 1, bytes));
    for (int ns : {1, 2, 4, 1, 2, 4}) {
        std::vector<hipStream_t> st(ns);
        for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        const size_t part = bytes / ns;
        for (int rep = 0; rep < 2; rep++) {           // first rep warms the pages
            CK(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < 4; r++)
                for (int i = 0; i < ns; i++) CK(hipMemcpyAsync((char*)h + i * part, (char*)d + i * part, part, hipMemcpyDeviceToHost, st[i]));
            for (auto& s : st) CK(hipStreamSynchronize(s));
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep) printf("%d stream(s): %.2f GB/s\n", ns, 4.0 * bytes / dt / 1e9);
        }
        for (auto& s : st) CK(hipStreamDestroy(s));
    }
    return 0;
}
