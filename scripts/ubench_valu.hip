// Micro-benchmark: issue rate of the VALU forms the SSIM kernels are built from (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 -o scripts/ubench_valu scripts/ubench_valu.hip
// Prints, per instruction form and waves/SIMD, shader cycles per wave-instruction per SIMD (s_memtime)
// and the wall-clock equivalent.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DPPL " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define DPPR " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define ROWL " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"

constexpr int kIters = 2000;

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, long long* cyc, float seed) {
    float a[8], b[8];
    float x = seed + threadIdx.x * 1e-3f, y = 1.0f + 1e-7f * threadIdx.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = x + i; b[i] = y + i; }
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[4], q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { p[i] = f2{x + i, x - i}; q[i] = f2{y, y + i}; }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; ++it) {
        if (MODE == 0) {
#define X(i) asm volatile("v_add_f32 %0, %1, %0\n\tv_add_f32 %0, %1, %0\n\tv_add_f32 %0, %1, %0\n\tv_add_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            REP8(X)
#undef X
        } else if (MODE == 1) {
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %0" DPPL "v_add_f32_dpp %0, %1, %0" DPPR "v_add_f32_dpp %0, %1, %0" DPPL "v_add_f32_dpp %0, %1, %0" DPPR : "+v"(a[i]) : "v"(b[i]));
            REP8(X)
#undef X
        } else if (MODE == 2) {
#define X(i) asm volatile("v_add_f32_dpp %0, %1, %0" ROWL "v_add_f32_dpp %0, %1, %0" ROWL "v_add_f32_dpp %0, %1, %0" ROWL "v_add_f32_dpp %0, %1, %0" ROWL : "+v"(a[i]) : "v"(b[i]));
            REP8(X)
#undef X
        } else if (MODE == 3) {
#define X(i) asm volatile("v_pk_add_f32 %0, %1, %0\n\tv_pk_add_f32 %0, %1, %0\n\tv_pk_add_f32 %0, %1, %0\n\tv_pk_add_f32 %0, %1, %0" : "+v"(p[i & 3]) : "v"(q[i & 3]));
            REP8(X)
#undef X
        } else if (MODE == 4) {
#define X(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0\n\tv_pk_fma_f32 %0, %1, %1, %0\n\tv_pk_fma_f32 %0, %1, %1, %0\n\tv_pk_fma_f32 %0, %1, %1, %0" : "+v"(p[i & 3]) : "v"(q[i & 3]));
            REP8(X)
#undef X
        } else if (MODE == 5) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %1, %0\n\tv_fma_f32 %0, %1, %1, %0\n\tv_fma_f32 %0, %1, %1, %0\n\tv_fma_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            REP8(X)
#undef X
        } else if (MODE == 6) {
#define X(i) asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %0, %0\n\tv_rcp_f32 %0, %0\n\tv_rcp_f32 %0, %0" : "+v"(a[i]));
            REP8(X)
#undef X
        } else if (MODE == 7) {
#define X(i) asm volatile("v_mov_b32_dpp %0, %1" DPPL "v_mov_b32_dpp %0, %1" DPPR "v_mov_b32_dpp %0, %1" DPPL "v_mov_b32_dpp %0, %1" DPPR : "+v"(a[i]) : "v"(b[i]));
            REP8(X)
#undef X
        } else if (MODE == 8) {
#define X(i) asm volatile("v_fmac_f32_dpp %0, %1, %1" DPPL "v_fmac_f32_dpp %0, %1, %1" DPPR "v_fmac_f32_dpp %0, %1, %1" DPPL "v_fmac_f32_dpp %0, %1, %1" DPPR : "+v"(a[i]) : "v"(b[i]));
            REP8(X)
#undef X
        } else if (MODE == 9) {
#define X(i) asm volatile("v_pk_mul_f32 %0, %1, %0\n\tv_pk_mul_f32 %0, %1, %0\n\tv_pk_mul_f32 %0, %1, %0\n\tv_pk_mul_f32 %0, %1, %0" : "+v"(p[i & 3]) : "v"(q[i & 3]));
            REP8(X)
#undef X
        } else if (MODE == 10) {  // two dependent-free streams mixing plain add and dpp add
#define X(i) asm volatile("v_add_f32 %0, %1, %0\n\tv_add_f32_dpp %0, %1, %0" DPPL "v_mul_f32 %0, %1, %0\n\tv_add_f32_dpp %0, %1, %0" DPPR : "+v"(a[i]) : "v"(b[i]));
            REP8(X)
#undef X
        } else if (MODE == 11) {
#define X(i) asm volatile("v_fmac_f32 %0, %1, %1\n\tv_fmac_f32 %0, %1, %1\n\tv_fmac_f32 %0, %1, %1\n\tv_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(b[i]));
            REP8(X)
#undef X
        } else if (MODE == 12) {
#define X(i) asm volatile("v_div_scale_f32 %0, vcc, %1, %1, %0\n\tv_div_scale_f32 %0, vcc, %1, %1, %0\n\tv_div_scale_f32 %0, vcc, %1, %1, %0\n\tv_div_scale_f32 %0, vcc, %1, %1, %0" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            REP8(X)
#undef X
        } else if (MODE == 13) {
#define X(i) asm volatile("v_div_fmas_f32 %0, %1, %1, %0\n\tv_div_fmas_f32 %0, %1, %1, %0\n\tv_div_fmas_f32 %0, %1, %1, %0\n\tv_div_fmas_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            REP8(X)
#undef X
        } else if (MODE == 14) {
#define X(i) asm volatile("v_div_fixup_f32 %0, %1, %1, %0\n\tv_div_fixup_f32 %0, %1, %1, %0\n\tv_div_fixup_f32 %0, %1, %1, %0\n\tv_div_fixup_f32 %0, %1, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            REP8(X)
#undef X
        } else if (MODE == 15) {
#define X(i) asm volatile("v_cmp_ge_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %1, %0, vcc\n\tv_cmp_ge_f32 vcc, %1, %0\n\tv_cndmask_b32 %0, %1, %0, vcc" : "+v"(a[i]) : "v"(b[i]) : "vcc");
            REP8(X)
#undef X
        } else if (MODE == 16) {
#define X(i) asm volatile("v_lshl_add_u64 %0, %0, 2, %1\n\tv_lshl_add_u64 %0, %0, 2, %1\n\tv_lshl_add_u64 %0, %0, 2, %1\n\tv_lshl_add_u64 %0, %0, 2, %1" : "+v"(p[i & 3]) : "v"(q[i & 3]));
            REP8(X)
#undef X
        } else if (MODE == 17) {
#define X(i) asm volatile("v_mul_f32 %0, %1, %0\n\tv_sub_f32 %0, %1, %0\n\tv_mul_f32 %0, %1, %0\n\tv_sub_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            REP8(X)
#undef X
        } else if (MODE == 18) {
#define X(i) asm volatile("v_med3_f32 %0, %1, %0, 1.0\n\tv_max_f32 %0, %1, %0\n\tv_med3_f32 %0, %1, %0, 1.0\n\tv_min_f32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            REP8(X)
#undef X
        } else if (MODE == 19) {
#define X(i) asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %0, %0" : "+v"(a[i]));
            REP8(X)
#undef X
        } else if (MODE == 20) {
#define X(i) asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %0, %1\n\tv_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b[i]));
            REP8(X)
#undef X
        } else if (MODE == 21) {
#define X(i) asm volatile("v_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %0, %1, %2, %0\n\tv_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i+1)&7]));
            REP8(X)
#undef X
        } else if (MODE == 22) {
#define X(i) asm volatile("v_mul_lo_u32 %0, %1, %0\n\tv_mul_lo_u32 %0, %1, %0\n\tv_mul_lo_u32 %0, %1, %0\n\tv_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
            REP8(X)
#undef X
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int MODE>
void run(const char* name, float* out, long long* cyc) {
    for (int wps : {2, 4}) {
        const int threads = 256 * wps, blocks = 256;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const int nw = blocks * threads / 64;
        std::vector<long long> h(nw);
        hipMemcpy(h.data(), cyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += (double)v; avg /= nw;
        const double instr = (double)kIters * 32;
        // s_memtime ticks at 100 MHz-constant? report both tick ratio and wall
        printf("%-22s waves/SIMD=%d  memtime_ticks/instr/SIMD=%.3f  wall_ns/instr/SIMD=%.4f  (=%.2f cyc @2.4GHz)\n", name, wps,
               avg / (instr * wps), ms * 1e6 / (instr * wps), ms * 1e6 / (instr * wps) * 2.4);
    }
}

int main() {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 1024 * sizeof(float));
    hipMalloc(&cyc, 256 * 16 * sizeof(long long));
    run<0>("v_add_f32", out, cyc);
    run<1>("v_add_f32_dpp wave", out, cyc);
    run<2>("v_add_f32_dpp row", out, cyc);
    run<3>("v_pk_add_f32", out, cyc);
    run<4>("v_pk_fma_f32", out, cyc);
    run<5>("v_fma_f32", out, cyc);
    run<6>("v_rcp_f32", out, cyc);
    run<7>("v_mov_b32_dpp wave", out, cyc);
    run<8>("v_fmac_f32_dpp wave", out, cyc);
    run<9>("v_pk_mul_f32", out, cyc);
    run<10>("mix add/dpp/mul", out, cyc);
    run<11>("v_fmac_f32 (vop2)", out, cyc);
    run<12>("v_div_scale_f32", out, cyc);
    run<13>("v_div_fmas_f32", out, cyc);
    run<14>("v_div_fixup_f32", out, cyc);
    run<15>("v_cmp+v_cndmask", out, cyc);
    run<16>("v_lshl_add_u64", out, cyc);
    run<17>("v_mul/v_sub", out, cyc);
    run<18>("v_med3/max/min", out, cyc);
    run<19>("v_exp_f32", out, cyc);
    run<20>("v_mov_b32", out, cyc);
    run<21>("v_fma_f32 3 srcs", out, cyc);
    run<22>("v_mul_lo_u32", out, cyc);
    return 0;
}
