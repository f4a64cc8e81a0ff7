// ubench_atomic: rate of device-scope float atomic adds on gfx950 (the image-gradient scatter of ds_warp_bwd).
//   hipcc -O3 --offload-arch=gfx950 -o scripts/ubench_atomic scripts/ubench_atomic.hip && scripts/ubench_atomic
// pattern 0: every lane its own address, coalesced per wave (64 consecutive floats per instruction)
// pattern 1: as 0, but consecutive waves overlap by half (two atomics land on every address)
// pattern 2: plain stores to the same addresses as 0, for comparison
// pattern 3: as 0 but a stride of 2 floats between lanes (half-used lines)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int PAT>
__global__ __launch_bounds__(256) void k(float* __restrict__ out, int per_wave) {
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    size_t base = wave * per_wave * 64;
    if (PAT == 1) base = base / 2;
    for (int i = 0; i < per_wave; ++i) {
        float* p = out + base + (size_t)i * 64 * (PAT == 3 ? 2 : 1) + lane * (PAT == 3 ? 2 : 1);
        if (PAT == 2) *p = 1.0f;
        else atomicAdd(p, 1.0f);
    }
}

int main() {
    const int waves = 46080, per_wave = 32;             // 94 M lane-operations, the scale of one B=12 192x640 scatter x 8
    const size_t n = (size_t)waves * per_wave * 64 * 2;
    float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemset(d, 0, n * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int pat = 0; pat < 4; ++pat) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(a, 0));
            switch (pat) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(waves / 4), dim3(256), 0, 0, d, per_wave); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(waves / 4), dim3(256), 0, 0, d, per_wave); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(waves / 4), dim3(256), 0, 0, d, per_wave); break;
                default: hipLaunchKernelGGL(k<3>, dim3(waves / 4), dim3(256), 0, 0, d, per_wave); break;
            }
            CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            if (ms < best) best = ms;
        }
        const double ops = (double)waves * per_wave * 64;
        printf("pattern %d: %.3f ms, %.1f G lane-ops/s, %.2f TB/s of 4-byte operands\n", pat, best, ops / best / 1e6, ops * 4 / best / 1e9);
    }
    return 0;
}
