// pcie_rates.hip -- host<->device copy rates on the box: pageable vs pinned, and host memcpy into a
// pinned staging buffer with 1..8 threads.  hipcc -O2 --offload-arch=gfx950 -o /tmp/pcie_rates tools/pcie_rates.hip -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = 1ull << 30;
    uint8_t *d, *pin, *page = (uint8_t*)malloc(N);
    memset(page, 1, N);
    CK(hipMalloc((void**)&d, N));
    CK(hipHostMalloc((void**)&pin, N));
    memset(pin, 2, N);
    for (int rep = 0; rep < 2; rep++) {
        double t = now();
        CK(hipMemcpy(d, page, N, hipMemcpyHostToDevice));
        printf("H2D pageable  %.1f GB/s\n", N / (now() - t) / 1e9);
        t = now();
        CK(hipMemcpy(d, pin, N, hipMemcpyHostToDevice));
        printf("H2D pinned    %.1f GB/s\n", N / (now() - t) / 1e9);
        t = now();
        CK(hipMemcpy(page, d, N, hipMemcpyDeviceToHost));
        printf("D2H pageable  %.1f GB/s\n", N / (now() - t) / 1e9);
        t = now();
        CK(hipMemcpy(pin, d, N, hipMemcpyDeviceToHost));
        printf("D2H pinned    %.1f GB/s\n", N / (now() - t) / 1e9);
        for (int nt : {1, 2, 4, 8}) {
            t = now();
            std::vector<std::thread> th;
            for (int k = 0; k < nt; k++) th.emplace_back([&, k] { memcpy(pin + N / nt * k, page + N / nt * k, N / nt); });
            for (auto& x : th) x.join();
            printf("memcpy page->pinned %d threads  %.1f GB/s\n", nt, N / (now() - t) / 1e9);
        }
        t = now();
        CK(hipHostRegister(page, N, hipHostRegisterDefault));
        double tr = now() - t;
        t = now();
        CK(hipMemcpy(d, page, N, hipMemcpyHostToDevice));
        printf("hipHostRegister %.1f GB/s, then H2D %.1f GB/s\n", N / tr / 1e9, N / (now() - t) / 1e9);
        CK(hipHostUnregister(page));
        t = now();
        uint8_t* fresh = (uint8_t*)malloc(N);
        CK(hipMemcpy(fresh, d, N, hipMemcpyDeviceToHost));
        printf("D2H into fresh malloc %.1f GB/s\n", N / (now() - t) / 1e9);
        free(fresh);
    }
    return 0;
}
