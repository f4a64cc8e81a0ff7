// Sustained-load check: does the plain-VALU issue rate measured by ubench_valu.hip (0.1 ms bursts) hold over long kernels?
// Build: hipcc -O3 --offload-arch=gfx950 -o scripts/ubench_sustain scripts/ubench_sustain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float seed) {
    float a[8], b[8];
    float x = seed + threadIdx.x * 1e-3f, y = 1.0f + 1e-7f * threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = x + i; b[i] = y + i; }
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#define X(i) asm volatile("v_add_f32 %0, %1, %0\n\tv_add_f32 %0, %1, %0\n\tv_add_f32 %0, %1, %0\n\tv_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            REP8(X)
#undef X
        } else if (MODE == 1) {   // 8 independent chains, different source registers per instruction
#define X(i) asm volatile("v_add_f32 %0, %1, %0\n\tv_mul_f32 %0, %2, %0\n\tv_fmac_f32 %0, %1, %2\n\tv_sub_f32 %0, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 3) & 7]));
            REP8(X)
#undef X
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, float* out, int wps, int iters, int reps) {
    const int threads = 256 * wps, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%s waves/SIMD=%d iters=%d: ns/instr/SIMD per launch:", name, wps, iters);
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf(" %.3f", ms * 1e6 / ((double)iters * 32 * wps));
    }
    printf("\n");
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    run<0>("v_add chain", out, 2, 2000, 3);
    run<0>("v_add chain", out, 2, 200000, 12);   // ~13 ms per launch
    run<0>("v_add chain", out, 4, 200000, 6);
    run<1>("add/mul/fmac/sub", out, 2, 2000, 3);
    run<1>("add/mul/fmac/sub", out, 2, 200000, 12);
    run<1>("add/mul/fmac/sub", out, 4, 200000, 6);
    return 0;
}
