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

#ifndef DS_OK
#define DS_OK 0
#endif
#define DS_ERR_INVALID_ARG (-1)
#define DS_ERR_UNSUPPORTED (-2)
#define DS_ERR_WORKSPACE (-3)
#define DS_ERR_SCRATCH_LIMIT (-4)

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

// ---------------------------------------------------------------------------------------------
// 3x3 window sums of the rolling-wave kernels.
//
// t, m, b are one lane's values of three consecutive rows; the left/right neighbours are the adjacent
// lanes, read through DPP wave shifts folded into the adds (v_add_f32_dpp).  The sum is sequential and
// row-major -- ((((((((L(t)+t)+R(t))+L(m))+m)+R(m))+L(b))+b)+R(b)) -- which is the rounding order of torch's
// avg_pool2d (the leading 0+ is exact).  Written as asm blocks because hipcc only folds a DPP move into
// its consumer when the two end up within 20 instructions of each other, which it does not arrange in
// kernels of this size (the un-folded form costs +40 % instructions and 3x the registers).
//
// Hazard bookkeeping (the compiler pads nothing inside an asm statement): a DPP operand must have been
// written at least 2 VALU instructions earlier.  Every block therefore opens with its v_mul products
// (>= 3 non-DPP instructions), and reads its own products through DPP no earlier than 3 instructions
// after writing them.  All 64 lanes are active wherever these are used.
// ---------------------------------------------------------------------------------------------
#define DS_DPP_L " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define DS_DPP_R " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define DS_W0(acc, v) "v_add_f32_dpp " acc ", " v ", " v DS_DPP_L      /* L(v) + v      */
#define DS_WL(acc, v) "v_add_f32_dpp " acc ", " v ", " acc DS_DPP_L    /* acc += L(v)   */
#define DS_WR(acc, v) "v_add_f32_dpp " acc ", " v ", " acc DS_DPP_R    /* acc += R(v)   */
#define DS_WC(acc, v) "v_add_f32 " acc ", " v ", " acc "\n\t"         /* acc += v      */

// target statistics: ay = win(y), ayy = win(y*y)
__device__ __forceinline__ void win_target(float yt, float ym, float yb, float& ay, float& ayy) {
    float p0, p1, p2;
    asm volatile(
        "v_mul_f32 %[p0], %[yt], %[yt]\n\t"
        "v_mul_f32 %[p1], %[ym], %[ym]\n\t"
        "v_mul_f32 %[p2], %[yb], %[yb]\n\t"
        DS_W0("%[ay]", "%[yt]") DS_W0("%[ayy]", "%[p0]")
        DS_WR("%[ay]", "%[yt]") DS_WR("%[ayy]", "%[p0]")
        DS_WL("%[ay]", "%[ym]") DS_WL("%[ayy]", "%[p1]")
        DS_WC("%[ay]", "%[ym]") DS_WC("%[ayy]", "%[p1]")
        DS_WR("%[ay]", "%[ym]") DS_WR("%[ayy]", "%[p1]")
        DS_WL("%[ay]", "%[yb]") DS_WL("%[ayy]", "%[p2]")
        DS_WC("%[ay]", "%[yb]") DS_WC("%[ayy]", "%[p2]")
        DS_WR("%[ay]", "%[yb]") DS_WR("%[ayy]", "%[p2]")
        : [ay] "=&v"(ay), [ayy] "=&v"(ayy), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2)
        : [yt] "v"(yt), [ym] "v"(ym), [yb] "v"(yb));
}

// image-pair statistics: ax = win(x), axx = win(x*x), axy = win(x*y)
__device__ __forceinline__ void win_pair(float xt, float xm, float xb, float yt, float ym, float yb, float& ax, float& axx,
                                         float& axy) {
    float p0, p1, p2, q0, q1, q2;
    asm volatile(
        "v_mul_f32 %[p0], %[xt], %[xt]\n\t"
        "v_mul_f32 %[q0], %[xt], %[yt]\n\t"
        "v_mul_f32 %[p1], %[xm], %[xm]\n\t"
        "v_mul_f32 %[q1], %[xm], %[ym]\n\t"
        "v_mul_f32 %[p2], %[xb], %[xb]\n\t"
        "v_mul_f32 %[q2], %[xb], %[yb]\n\t"
        DS_W0("%[ax]", "%[xt]") DS_W0("%[axx]", "%[p0]") DS_W0("%[axy]", "%[q0]")
        DS_WR("%[ax]", "%[xt]") DS_WR("%[axx]", "%[p0]") DS_WR("%[axy]", "%[q0]")
        DS_WL("%[ax]", "%[xm]") DS_WL("%[axx]", "%[p1]") DS_WL("%[axy]", "%[q1]")
        DS_WC("%[ax]", "%[xm]") DS_WC("%[axx]", "%[p1]") DS_WC("%[axy]", "%[q1]")
        DS_WR("%[ax]", "%[xm]") DS_WR("%[axx]", "%[p1]") DS_WR("%[axy]", "%[q1]")
        DS_WL("%[ax]", "%[xb]") DS_WL("%[axx]", "%[p2]") DS_WL("%[axy]", "%[q2]")
        DS_WC("%[ax]", "%[xb]") DS_WC("%[axx]", "%[p2]") DS_WC("%[axy]", "%[q2]")
        DS_WR("%[ax]", "%[xb]") DS_WR("%[axx]", "%[p2]") DS_WR("%[axy]", "%[q2]")
        : [ax] "=&v"(ax), [axx] "=&v"(axx), [axy] "=&v"(axy), [p0] "=&v"(p0), [p1] "=&v"(p1), [p2] "=&v"(p2),
          [q0] "=&v"(q0), [q1] "=&v"(q1), [q2] "=&v"(q2)
        : [xt] "v"(xt), [xm] "v"(xm), [xb] "v"(xb), [yt] "v"(yt), [ym] "v"(ym), [yb] "v"(yb));
}

// h = v + wL * (lane-1's v) + wR * (lane+1's v) for three values at once: the horizontal spread of the
// backward kernels (wL, wR are 1, or 2 where the reflect padding folds a border window back).  The three
// moves in front keep every DPP source at least 2 VALU instructions old.
__device__ __forceinline__ void spread3(float a, float b, float c, float wL, float wR, float& ha, float& hb, float& hc) {
    asm volatile(
        "v_mov_b32 %[ha], %[a]\n\t"
        "v_mov_b32 %[hb], %[b]\n\t"
        "v_mov_b32 %[hc], %[c]\n\t"
        "v_fmac_f32_dpp %[ha], %[a], %[wL]" DS_DPP_L
        "v_fmac_f32_dpp %[hb], %[b], %[wL]" DS_DPP_L
        "v_fmac_f32_dpp %[hc], %[c], %[wL]" DS_DPP_L
        "v_fmac_f32_dpp %[ha], %[a], %[wR]" DS_DPP_R
        "v_fmac_f32_dpp %[hb], %[b], %[wR]" DS_DPP_R
        "v_fmac_f32_dpp %[hc], %[c], %[wR]" DS_DPP_R
        : [ha] "=&v"(ha), [hb] "=&v"(hb), [hc] "=&v"(hc)
        : [a] "v"(a), [b] "v"(b), [c] "v"(c), [wL] "v"(wL), [wR] "v"(wR));
}

// ---------------------------------------------------------------------------------------------
// Row loads of the rolling kernels: buffer_load with the image base in an SGPR descriptor, the wave-uniform
// plane/row offset in an SGPR (soffset) and the lane's column offset in one VGPR -- no per-load VALU address
// arithmetic.  Offsets are 32-bit: one descriptor per image, whose planes must span < 4 GiB (checked on the host).
// ---------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000);
}
__device__ __forceinline__ float buf_load_f32(rsrc_t r, unsigned lane_bytes, unsigned uniform_elems) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_bytes, uniform_elems * 4u, 0));
}
__device__ __forceinline__ int buf_load_u8(rsrc_t r, unsigned lane_bytes, unsigned uniform_bytes) {
    return (int)__builtin_amdgcn_raw_buffer_load_b8(r, lane_bytes, uniform_bytes, 0);
}

// the same with the non-temporal bit (gfx940+: aux bit 1 = nt): data that is used once should not push re-used lines out of L2
__device__ __forceinline__ float buf_load_f32_nt(rsrc_t r, unsigned lane_bytes, unsigned uniform_elems) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, lane_bytes, uniform_elems * 4u, 2));
}
__device__ __forceinline__ void buf_store_f32_nt(float v, rsrc_t r, unsigned lane_bytes, unsigned uniform_bytes) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, lane_bytes, uniform_bytes, 2);
}

typedef float float2_t __attribute__((ext_vector_type(2)));
// two adjacent floats at a 4-byte aligned address (gfx950 buffer loads need no natural alignment)
__device__ __forceinline__ float2_t buf_load_f32x2(rsrc_t r, unsigned lane_bytes, unsigned uniform_bytes) {
    return __builtin_bit_cast(float2_t, __builtin_amdgcn_raw_buffer_load_b64(r, lane_bytes, uniform_bytes, 0));
}
__device__ __forceinline__ void buf_store_f32(float v, rsrc_t r, unsigned lane_bytes, unsigned uniform_bytes) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, lane_bytes, uniform_bytes, 0);
}

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

// n/d, correctly rounded, for operands that need no range scaling: the instruction sequence hipcc emits for an
// IEEE f32 divide minus its v_div_scale / v_div_fixup range handling (12 -> 8 instructions).  The SSIM
// denominator is >= C1*C2 = 9e-8 and both operands stay below ~2^40 for pixel magnitudes below 2^8, far from the
// exponents where v_div_scale would rescale, so the result is bit-identical to the full sequence;
// tests/test_gpu_parity.py pins it against the oracle's true divide.
__device__ __forceinline__ float div_noscale(float n, float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = fmaf(-d, r0, 1.0f);
    const float r1 = fmaf(e0, r0, r0);
    const float q0 = n * r1;
    const float e1 = fmaf(-d, q0, n);
    const float q1 = fmaf(e1, r1, q0);
    const float e2 = fmaf(-d, q1, n);
    return fmaf(e2, r1, q1);
}

// The same sequence split for several numerators over one denominator: r1 = rcp_refined(d) once (or, for a divisor
// known on the host, the correctly rounded 1.0f/d), then 5 instructions per quotient.  Used by the warp kernels for
// u = q0/z, v = q1/z (|z| >= 1e-6 by construction) and for the division by the source extent (Ws-1, Hs-1 >= 1):
// operands far from the exponents where v_div_scale would rescale, results bit-identical to the IEEE divide
// (tests/test_gpu_parity.py::test_warp_forward_bit_exact pins them against the oracle's true divide).
__device__ __forceinline__ float rcp_refined(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = fmaf(-d, r0, 1.0f);
    return fmaf(e0, r0, r0);
}
__device__ __forceinline__ float div_by(float n, float d, float r1) {
    const float q0 = n * r1;
    const float e1 = fmaf(-d, q0, n);
    const float q1 = fmaf(e1, r1, q0);
    const float e2 = fmaf(-d, q1, n);
    return fmaf(e2, r1, q1);
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

// A value every lane of the wave holds alike, pinned to a scalar register: where the compiler cannot prove that a
// uniform-address load is not clobbered (kernels that also store) it loads through the vector path and the 30 floats of
// a camera sit in VGPRs -- in photo_fb_kernel that was 60 of the 128 a wave has.
__device__ __forceinline__ float wave_uniform(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

__device__ __forceinline__ void load_cam(Cam& c, const float* __restrict__ K4, const float* __restrict__ Kinv4,
                                         const float* __restrict__ pose) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            c.K[3 * j + k] = wave_uniform(K4[4 * j + k]);
            c.Kinv[3 * j + k] = wave_uniform(Kinv4[4 * j + k]);
        }
#pragma unroll
    for (int i = 0; i < 12; ++i) c.M[i] = wave_uniform(pose[i]);
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

// corner offsets/validity shared by every channel of one pixel.  The two texels of a corner row are adjacent in
// memory, so each row is ONE 8-byte load (4-byte aligned; gfx950 global loads need no natural alignment): 2
// gather instructions per channel instead of 4 -- the warp kernels are bound by gather address processing.
// At the right border (x0 == Ws-1) the pair is shifted one texel left and the values re-assigned.
struct Corners {
    int o_top, o_bot;   // element offset of the LEFT texel of the loaded pair in rows y0 / y1 (a valid address)
    bool shift;         // x0 == Ws-1: the pair starts at x0-1, so nw/sw are its second texel and ne/se do not exist
    bool xin, yin;      // x0+1 < Ws, y0+1 < Hs
};

struct __attribute__((packed, aligned(4))) TexelPair { float a, b; };

__device__ __forceinline__ void corner_offsets(const Samp& s, int Hs, int Ws, Corners& c) {
    c.xin = s.x0 + 1 < Ws;
    c.yin = s.y0 + 1 < Hs;
    c.shift = !c.xin;
    const int xc = c.xin ? s.x0 : s.x0 - 1;   // Ws >= 2 (checked on the host)
    const int y1 = c.yin ? s.y0 + 1 : s.y0;
    c.o_top = s.y0 * Ws + xc;
    c.o_bot = y1 * Ws + xc;
}

__device__ __forceinline__ void gather4(const float* __restrict__ plane, const Corners& c, float& nw, float& ne,
                                        float& sw, float& se) {
    const TexelPair t = *reinterpret_cast<const TexelPair*>(plane + c.o_top);
    const TexelPair b = *reinterpret_cast<const TexelPair*>(plane + c.o_bot);
    nw = c.shift ? t.b : t.a;
    ne = c.shift ? 0.0f : t.b;
    sw = c.yin ? (c.shift ? b.b : b.a) : 0.0f;
    se = (c.xin && c.yin) ? b.b : 0.0f;
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

struct SsimOut { float mx, sx, sxy, n, dd, v, rdd; };  // rdd = refined 1/dd (by-product of the exact n/dd)

__device__ __forceinline__ void ssim_from_sums(float ax, float axx, float axy, const TgtStats& t, SsimOut& o) {
    o.mx = div9(ax);
    float mx2 = o.mx * o.mx;
    float mxy = o.mx * t.my;
    o.sx = div9(axx) - mx2;
    o.sxy = div9(axy) - mxy;
    // (2*mx)*my == 2*(mx*my) and 2*s+C rounds once either way: the doublings are exact
    o.n = fmaf(2.0f, mxy, kC1) * fmaf(2.0f, o.sxy, kC2);
    o.dd = ((mx2 + t.my2) + kC1) * ((o.sx + t.sy_raw) + kC2);
    o.rdd = rcp_refined(o.dd);
    o.v = (1.0f - div_by(o.n, o.dd, o.rdd)) * 0.5f;   // == div_noscale(n, dd)
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
