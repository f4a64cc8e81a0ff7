// ds_common.h -- device-side helpers shared by the depthsynth gfx950 kernels.
//
// Numeric contract (pinned by oracle/ds_oracle.c against the reference's outputs):
//   * this library is compiled with -ffp-contract=off: a*b+c is two roundings unless written fmaf();
//   * the three small matmuls of reproject_from_depth (synthesis.py:100,109,112) are k-ordered fmaf chains;
//   * grid_sample's four corner products are an fmaf chain nw,ne,sw,se (torch CPU kernel, bit-exact);
//   * avg_pool2d is a sequential row-major 9-tap sum followed by a correctly rounded divide by 9.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DS_OK 0
#define DS_ERR_INVALID_ARG (-1)
#define DS_ERR_UNSUPPORTED (-2)
#define DS_ERR_WORKSPACE (-3)

namespace ds {

constexpr float kC1 = 0.0001f;  // float32(0.01**2), metrics.py:46
constexpr float kC2 = 0.0009f;  // float32(0.03**2), metrics.py:46

// ---------------------------------------------------------------------------------------------
// lane neighbours inside a 64-wide wavefront (DPP wave shifts fold into the consuming VALU op)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float lane_left(float v) {   // value held by lane-1 (lane 0: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_right(float v) {  // value held by lane+1 (lane 63: 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ int lane_left_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ int lane_right_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, false);
}

// Makes a register value opaque to the optimiser (no instruction is emitted).  The rolling kernels call
// it on their row registers at the top of every row step: without it the compiler CSEs the DPP-shifted
// copies of a row across the three steps that use it (3 live registers per row value instead of one,
// and the shifts stay separate v_mov_dpp instead of folding into the adds).
__device__ __forceinline__ void opaque(float& v) { asm volatile("" : "+v"(v)); }

// Ties the (future) DPP source v to the running sum acc: no instruction, but the shift of v can no longer
// be scheduled ahead of acc.  LLVM's DPP combiner only folds a v_mov_dpp into its consumer when the two
// are within 20 instructions; un-tied, all shifts of a row step are hoisted to its top and stay v_movs.
__device__ __forceinline__ void tie(float& acc, float& v) { asm("" : "+v"(acc), "+v"(v)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// correctly rounded x/9 and x/3 without the generic divide sequence (Markstein: y = RN(1/b),
// q = RN(x*y), r = x - q*b exactly via fma, q' = RN(q + r*y) is the correctly rounded quotient for
// these divisors; inputs here are sums of pixel values/products, far from the subnormal range).
// tests/test_gpu_parity.py::test_ssim_bit_exact pins the result against the oracle's true divide.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float div9(float x) {
    const float y = 1.0f / 9.0f;
    float q = x * y;
    float r = fmaf(-q, 9.0f, x);
    return fmaf(r, y, q);
}
__device__ __forceinline__ float div3(float x) {
    const float y = 1.0f / 3.0f;
    float q = x * y;
    float r = fmaf(-q, 3.0f, x);
    return fmaf(r, y, q);
}

__device__ __forceinline__ int reflect_idx(int i, int L) {  // reflect pad 1, then clamp (lanes outside the band)
    i = i < 0 ? -i : (i > L - 1 ? 2 * (L - 1) - i : i);
    return min(max(i, 0), L - 1);
}

// ---------------------------------------------------------------------------------------------
// a2: synthesis.reproject_from_depth, one pixel (synthesis.py:100-114)
// ---------------------------------------------------------------------------------------------
struct Cam {            // wave-uniform: the compiler keeps these in SGPRs
    float Kinv[9];      // intrinsic_inv[:3,:3]
    float K[9];         // intrinsic_mat[:3,:3]
    float M[12];        // pose (3,4)
};

__device__ __forceinline__ void load_cam(Cam& c, const float* __restrict__ K4, const float* __restrict__ Kinv4,
                                         const float* __restrict__ pose) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            c.K[3 * j + k] = K4[4 * j + k];
            c.Kinv[3 * j + k] = Kinv4[4 * j + k];
        }
#pragma unroll
    for (int i = 0; i < 12; ++i) c.M[i] = pose[i];
}

struct Proj {
    float c[3], P[3], T[3], q[3], z, u, v;
};

__device__ __forceinline__ void project_pixel(const Cam& cam, float fx, float fy, float d, Proj& o) {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float acc = fx * cam.Kinv[3 * j + 0];
        acc = fmaf(fy, cam.Kinv[3 * j + 1], acc);
        acc = fmaf(1.0f, cam.Kinv[3 * j + 2], acc);
        o.c[j] = acc;
        o.P[j] = acc * d;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float acc = o.P[0] * cam.M[4 * j + 0];
        acc = fmaf(o.P[1], cam.M[4 * j + 1], acc);
        acc = fmaf(o.P[2], cam.M[4 * j + 2], acc);
        acc = fmaf(1.0f, cam.M[4 * j + 3], acc);
        o.T[j] = acc;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float acc = o.T[0] * cam.K[3 * j + 0];
        acc = fmaf(o.T[1], cam.K[3 * j + 1], acc);
        acc = fmaf(o.T[2], cam.K[3 * j + 2], acc);
        o.q[j] = acc;
    }
    o.z = (fabsf(o.q[2]) < 1e-6f) ? 1e-6f : o.q[2];
    o.u = o.q[0] / o.z;
    o.v = o.q[1] / o.z;
}

// ---------------------------------------------------------------------------------------------
// a3: synthesis.synthesize, one pixel: normalise with the SOURCE dims (synthesis.py:49-51), un-normalise
// as grid_sample(align_corners=True), clip to the border, bilinear weights.
// ---------------------------------------------------------------------------------------------
struct Samp {
    float gx, gy;        // normalised coords (what the reference leaves in proj_pixels)
    float ixu, iyu;      // un-normalised, un-clipped
    int x0, y0;
    float wx, ex, ny, sy;
    float w_nw, w_ne, w_sw, w_se;
};

__device__ __forceinline__ void normalise_coords(float u, float v, int Hs, int Ws, float& gx, float& gy) {
    gx = (u / (float)(Ws - 1)) * 2.0f - 1.0f;
    gy = (v / (float)(Hs - 1)) * 2.0f - 1.0f;
}

__device__ __forceinline__ void sample_setup_norm(float gx, float gy, int Hs, int Ws, Samp& s) {
    s.gx = gx;
    s.gy = gy;
    float ix = (gx + 1.0f) * ((float)(Ws - 1) / 2.0f);
    float iy = (gy + 1.0f) * ((float)(Hs - 1) / 2.0f);
    s.ixu = ix;
    s.iyu = iy;
    ix = fminf((float)(Ws - 1), fmaxf(ix, 0.0f));
    iy = fminf((float)(Hs - 1), fmaxf(iy, 0.0f));
    float xf = floorf(ix), yf = floorf(iy);
    s.x0 = (int)xf;
    s.y0 = (int)yf;
    s.wx = ix - xf;
    s.ex = 1.0f - s.wx;
    s.ny = iy - yf;
    s.sy = 1.0f - s.ny;
    s.w_nw = s.sy * s.ex;
    s.w_ne = s.sy * s.wx;
    s.w_sw = s.ny * s.ex;
    s.w_se = s.ny * s.wx;
}

__device__ __forceinline__ void sample_setup(float u, float v, int Hs, int Ws, Samp& s) {
    float gx, gy;
    normalise_coords(u, v, Hs, Ws, gx, gy);
    sample_setup_norm(gx, gy, Hs, Ws, s);
}

// corner offsets/validity shared by every channel of one pixel
struct Corners {
    int o_nw, o_ne, o_sw, o_se;   // element offsets inside one (Hs,Ws) plane (clamped to a valid address)
    bool xin, yin;                // x0+1 < Ws, y0+1 < Hs
};

__device__ __forceinline__ void corner_offsets(const Samp& s, int Hs, int Ws, Corners& c) {
    c.xin = s.x0 + 1 < Ws;
    c.yin = s.y0 + 1 < Hs;
    int x1 = c.xin ? s.x0 + 1 : s.x0;
    int y1 = c.yin ? s.y0 + 1 : s.y0;
    c.o_nw = s.y0 * Ws + s.x0;
    c.o_ne = s.y0 * Ws + x1;
    c.o_sw = y1 * Ws + s.x0;
    c.o_se = y1 * Ws + x1;
}

__device__ __forceinline__ void gather4(const float* __restrict__ plane, const Corners& c, float& nw, float& ne,
                                        float& sw, float& se) {
    nw = plane[c.o_nw];
    float a = plane[c.o_ne], b = plane[c.o_sw], d = plane[c.o_se];
    ne = c.xin ? a : 0.0f;
    sw = c.yin ? b : 0.0f;
    se = (c.xin && c.yin) ? d : 0.0f;
}

__device__ __forceinline__ float bilinear(float nw, float ne, float sw, float se, const Samp& s) {
    float acc = nw * s.w_nw;
    acc = fmaf(ne, s.w_ne, acc);
    acc = fmaf(sw, s.w_sw, acc);
    return fmaf(se, s.w_se, acc);
}

// ---------------------------------------------------------------------------------------------
// SSIM window formula from the five sequential window sums (metrics.py:37-50)
// ---------------------------------------------------------------------------------------------
struct TgtStats { float my, sy_raw, my2; };  // my = pool(y), sy = pool(y*y) - my*my, my2 = my*my

__device__ __forceinline__ void target_stats(float ay, float ayy, TgtStats& t) {
    t.my = div9(ay);
    t.my2 = t.my * t.my;
    t.sy_raw = div9(ayy) - t.my2;
}

struct SsimOut { float mx, sx, sxy, n, dd, v; };

__device__ __forceinline__ void ssim_from_sums(float ax, float axx, float axy, const TgtStats& t, SsimOut& o) {
    o.mx = div9(ax);
    float mx2 = o.mx * o.mx;
    float mxy = o.mx * t.my;
    o.sx = div9(axx) - mx2;
    o.sxy = div9(axy) - mxy;
    // (2*mx)*my == 2*(mx*my) and 2*s+C rounds once either way: the doublings are exact
    o.n = fmaf(2.0f, mxy, kC1) * fmaf(2.0f, o.sxy, kC2);
    o.dd = ((mx2 + t.my2) + kC1) * ((o.sx + t.sy_raw) + kC2);
    o.v = (1.0f - o.n / o.dd) * 0.5f;
}

__device__ __forceinline__ float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

// XCD-aware bijective block remap (cdna guide T1): blocks b, b+8, b+16... share an XCD/L2, so give
// each of the 8 groups one contiguous chunk of the work list.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, k = bid >> 3;
    int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

}  // namespace ds

#define DS_CHECK_LAUNCH()                                  \
    do {                                                   \
        hipError_t e_ = hipGetLastError();                 \
        if (e_ != hipSuccess) return (int)e_;              \
    } while (0)
