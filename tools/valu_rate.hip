// Micro-benchmark: issue rate of the integer VALU ops the DP kernel is built from (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short pk16 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(256) void k(int* out, int iters, int seed) {
    int a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    int b = seed | 1;
    for (int i = 0; i < iters; i++) {
#define STEP(x)                                                                                                  \
    if (OP == 0) x = __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(pk16, x), __builtin_bit_cast(pk16, b))); \
    else if (OP == 1) x = __builtin_bit_cast(int, __builtin_bit_cast(pk16, x) + __builtin_bit_cast(pk16, b));   \
    else if (OP == 2) x = max(x, b);                                                                             \
    else if (OP == 3) x = x + b;                                                                                 \
    else if (OP == 4) x = (x == b) ? 1 : -1;                                                                     \
    else if (OP == 5) x = max(max(x, b), i);                                                                     \
    else if (OP == 6) x = __builtin_bit_cast(int, __builtin_fmaf(__builtin_bit_cast(float, x), 1.0001f, 0.5f));
        STEP(a0) STEP(a1) STEP(a2) STEP(a3) STEP(a4) STEP(a5) STEP(a6) STEP(a7)
        asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
template <int OP>
void run(const char* name, int* d) {
    const int iters = 20000;
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 256 * wps;  // 256 CUs x (wps waves per SIMD x 4 SIMDs / 4 waves per block)
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 100, 1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double instr = (double)blocks * 4 * iters * 8;  // wave-instructions of the op under test
        double per_simd_per_us = instr / 1024 / (ms * 1e3);
        printf("%-14s waves/SIMD=%d  %.2f ms  %.0f wave-instr/us/SIMD (= %.2f cycles/instr at 2.4 GHz)\n", name, wps, ms,
               per_simd_per_us, 2400.0 / per_simd_per_us);
    }
}
int main() {
    int* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("v_pk_max_i16", d); run<1>("v_pk_add_u16", d); run<2>("v_max_i32", d); run<3>("v_add_u32", d);
    run<4>("cmp+cndmask", d); run<5>("v_max3_i32", d); run<6>("v_fma_f32", d);
    return 0;
}
