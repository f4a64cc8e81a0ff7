// ds_warp.hip -- back-project -> transform -> project -> bilinear gather, forward and backward.
//
// Reference: synthesis.reproject_from_depth (synthesis.py:83-116), synthesis.synthesize
// (synthesis.py:36-54) and their autograd; callers train.py:72-77.
//
// All kernels here hold one pixel per lane on planar NCHW images, consecutive lanes on consecutive x, so depth reads /
// warped writes are 256-B coalesced per wave and the 4x3 corner gathers hit neighbouring source texels (L1/L2-
// resident lines); no LDS except the pose-gradient block reduction.  The two hot ones (warp_fwd_kernel,
// warp_bwd_kernel) are tiled and written for instruction count (see the comment above WarpGeom in ds_warp_dev.h's
// users below); the per-function entry points of the reference's unfused call sequence keep the simple form.
#include "ds_common.h"
#include "ds_warp_dev.h"
#include "ds_internal.h"
#include "depthsynth.h"

using namespace ds;

namespace {

constexpr int kThreads = 256;

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void reproject_kernel(const float* __restrict__ K, const float* __restrict__ Kinv,
                                                             const float* __restrict__ pose,
                                                             const float* __restrict__ depth, int H, int W,
                                                             float* __restrict__ coords) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= H * W) return;
    Cam cam;
    load_cam(cam, K, Kinv, pose + 12 * n);
    const int y = p / W, x = p - y * W;
    Proj pr;
    project_pixel(cam, (float)x, (float)y, depth[(size_t)n * H * W + p], pr);
    float2* o = reinterpret_cast<float2*>(coords) + (size_t)n * H * W + p;
    *o = make_float2(pr.u, pr.v);
}

// ---------------------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(kThreads) void grid_sample_fwd_kernel(const float* __restrict__ src, float* __restrict__ grid,
                                                                   int Cdyn, int Hs, int Ws, int H, int W,
                                                                   float* __restrict__ out) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= H * W) return;
    const int Cc = C > 0 ? C : Cdyn;
    float2* g = reinterpret_cast<float2*>(grid) + (size_t)n * H * W + p;
    float2 uv = *g;
    Samp s;
    sample_setup(uv.x, uv.y, Hs, Ws, s);
    *g = make_float2(s.gx, s.gy);  // the reference normalises its argument in place (synthesis.py:50-51)
    Corners co;
    corner_offsets(s, Hs, Ws, co);
    const float* sp = src + (size_t)n * Cc * Hs * Ws;
    float* op = out + (size_t)n * Cc * H * W + p;
    for (int c = 0; c < Cc; ++c) {  // fully unrolled by the optimiser when C is the template constant
        float nw, ne, sw, se;
        gather4(sp + (size_t)c * Hs * Ws, co, nw, ne, sw, se);
        op[(size_t)c * H * W] = bilinear(nw, ne, sw, se, s);
    }
}

// grid gradient of one pixel from the corner texels (torch grid_sampler_2d backward, bilinear/border):
// zero when the un-clipped coordinate lies on or outside the border; un-normalise * normalise factors.
__device__ __forceinline__ void finish_grid_grad(const Samp& s, int Hs, int Ws, float& gix, float& giy) {
    if (!(s.ixu > 0.0f && s.ixu < (float)(Ws - 1))) gix = 0.0f;
    if (!(s.iyu > 0.0f && s.iyu < (float)(Hs - 1))) giy = 0.0f;
    gix = gix * ((float)(Ws - 1) / 2.0f) * (2.0f / (float)(Ws - 1));
    giy = giy * ((float)(Hs - 1) / 2.0f) * (2.0f / (float)(Hs - 1));
}

__device__ __forceinline__ void scatter_src_grad(float* __restrict__ gp, const Corners& co, const Samp& s, float g) {
    if (g == 0.0f) return;
    const int sh = co.shift ? 1 : 0;
    atomicAdd(gp + co.o_top + sh, g * s.w_nw);
    if (co.xin) atomicAdd(gp + co.o_top + 1, g * s.w_ne);
    if (co.yin) atomicAdd(gp + co.o_bot + sh, g * s.w_sw);
    if (co.xin && co.yin) atomicAdd(gp + co.o_bot + 1, g * s.w_se);
}

template <int C>
__global__ __launch_bounds__(kThreads) void grid_sample_bwd_kernel(const float* __restrict__ src,
                                                                   const float* __restrict__ grid_norm,
                                                                   const float* __restrict__ g_out, int Cdyn, int Hs,
                                                                   int Ws, int H, int W, float* __restrict__ g_grid,
                                                                   float* __restrict__ g_src) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= H * W) return;
    const int Cc = C > 0 ? C : Cdyn;
    float2 gn = reinterpret_cast<const float2*>(grid_norm)[(size_t)n * H * W + p];
    Samp s;
    sample_setup_norm(gn.x, gn.y, Hs, Ws, s);
    Corners co;
    corner_offsets(s, Hs, Ws, co);
    const float* sp = src + (size_t)n * Cc * Hs * Ws;
    const float* gp = g_out + (size_t)n * Cc * H * W + p;
    float gix = 0.f, giy = 0.f;
    for (int c = 0; c < Cc; ++c) {
        float nw, ne, sw, se;
        gather4(sp + (size_t)c * Hs * Ws, co, nw, ne, sw, se);
        float g = gp[(size_t)c * H * W];
        gix = fmaf(g, fmaf(ne - nw, s.sy, (se - sw) * s.ny), gix);
        giy = fmaf(g, fmaf(sw - nw, s.ex, (se - ne) * s.wx), giy);
        if (g_src) scatter_src_grad(g_src + ((size_t)n * Cc + c) * Hs * Ws, co, s, g);
    }
    finish_grid_grad(s, Hs, Ws, gix, giy);
    reinterpret_cast<float2*>(g_grid)[(size_t)n * H * W + p] = make_float2(gix, giy);
}

// ---------------------------------------------------------------------------------------------------
// The two hot kernels (ds_warp_fwd / ds_warp_bwd) are VALU-heavy (~190 / ~300 instructions per pixel in their first
// version, 60 % of their run time), so they are written for instruction count:
//   * tiles instead of a flat pixel index (no integer division): a wave owns 64 adjacent columns of kRows rows, the
//     four waves of a block sit below each other;
//   * buffer loads / stores with the image base in an SGPR descriptor, a 32-bit lane offset and a uniform offset (no
//     64-bit address arithmetic per access);
//   * the divisions of the forward projection share one refined reciprocal per denominator (u = q0/z, v = q1/z) or
//     take it from the host (division by Ws-1, Hs-1): 5 instructions per correctly rounded quotient instead of 12;
//   * the border cases of the bilinear gather select WEIGHTS once per pixel, not texel values once per channel.
template <int C>
__global__ __launch_bounds__(kThreads) void warp_fwd_kernel(const float* __restrict__ K, const float* __restrict__ Kinv,
                                                            const float* __restrict__ pose,
                                                            const float* __restrict__ depth,
                                                            const float* __restrict__ src, int Cdyn, WarpGeom G,
                                                            float* __restrict__ warped) {
    warp_fwd_tile<C>(K, Kinv, pose, depth, src, Cdyn, G, warped, blockIdx.y, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------
// warp backward: per pixel g_depth; per (n) 12-float pose gradient reduced wave -> block -> partial
// buffer -> pose_finalize_kernel (fixed order: deterministic, no atomics); optional image scatter.
constexpr int kBwdPix = 4;  // pixels per thread (reproject_bwd_kernel)

// autograd of project_pixel: (g_u, g_v) = d loss / d(u, v) -> g_depth, += pose gradient
__device__ __forceinline__ void project_bwd_pixel(const Cam& cam, const Proj& pr, float gix, float giy, float& g_d,
                                                  float (&gM)[12]) {
    const float rz = 1.0f / pr.z;
    float gq0 = gix * rz, gq1 = giy * rz;
    float gq2 = (fabsf(pr.q[2]) < 1e-6f) ? 0.0f : -(fmaf(pr.u, gix, pr.v * giy)) * rz;
    float gT[3], gP[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) gT[k] = fmaf(cam.K[6 + k], gq2, fmaf(cam.K[3 + k], gq1, cam.K[k] * gq0));
#pragma unroll
    for (int i = 0; i < 3; ++i) gP[i] = fmaf(cam.M[8 + i], gT[2], fmaf(cam.M[4 + i], gT[1], cam.M[i] * gT[0]));
    g_d = fmaf(pr.c[2], gP[2], fmaf(pr.c[1], gP[1], pr.c[0] * gP[0]));
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        gM[4 * j + 0] = fmaf(gT[j], pr.P[0], gM[4 * j + 0]);
        gM[4 * j + 1] = fmaf(gT[j], pr.P[1], gM[4 * j + 1]);
        gM[4 * j + 2] = fmaf(gT[j], pr.P[2], gM[4 * j + 2]);
        gM[4 * j + 3] += gT[j];
    }
}

// block-level reduction of 12 floats per thread -> partial[12]; every thread of the block must call it
__device__ __forceinline__ void block_reduce12(float (&gM)[12], float* __restrict__ partial, float (*lds)[12]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < 12; ++i) gM[i] = wave_sum(gM[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) lds[wave][i] = gM[i];
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        float a = 0.f;
        for (int w = 0; w < nw; ++w) a += lds[w][threadIdx.x];
        partial[threadIdx.x] = a;
    }
}

#ifndef DS_BWD_ROWS
#define DS_BWD_ROWS 8
#endif
constexpr int kBwdRows = DS_BWD_ROWS;  // rows per wave: the 12-float pose-gradient wave reduction is paid once per 8 pixels
constexpr int kBwdRowsSmall = 2;       // for launches too small to fill the chip with kBwdRows

// Image-gradient scatter (IMG).  Direct scatter = 4 device-scope float atomics per pixel and channel: 382 us against 31 us
// for the rest of the backward at B=12, 192x640.  Collecting the taps in an LDS tile first was measured too: the LDS float
// atomics (~20 cycles per wave instruction) then cost what the global ones did (213 us; scripts/ubench_atomic.hip for the
// global rates: 324 G lane-operations/s coalesced and aligned).  What is done instead needs neither: for a smooth warp the
// taps of the 64 pixels of a wave row fall into a window of three source rows x 66 columns that moves down one row per
// pixel row, so
//   * lane x owns source column x + cb of the window; the taps of lanes x-1, x, x+1 that fall into that column are
//     combined in registers (two lane shuffles per value), three running sums per channel (rows r0, r0+1, r0+2);
//   * when the window moves on, the finished row r0 goes to memory with ONE coalesced atomic per texel and channel
//     (atomic because the windows of neighbouring waves overlap at their edges); about 1.25 atomics per pixel and
//     channel instead of 4, all of them full aligned lines;
//   * a pixel whose taps do not fit the window (a near pixel among far ones: its disparity differs by more than a texel)
//     is broadcast from its lane and taken over by the two lanes that own its columns if rows and columns are still
//     inside the window, otherwise that lane scatters it directly with four atomics per channel; if more than a few
//     lanes of a row misfit (strong minification, source larger than the grid) the whole row is scattered directly.
//     The result never depends on the path taken.
// B=12, 192x640, image gradient on: 382 us direct, 213 us LDS tile, 128 us this way with bench.py's random per-pixel
// depth (about 5 % misfits), 91 us with a smooth depth map; 34 us without the image gradient (scripts/img_grad_bench.py).
struct ImgWindow {
    float a[3][3];   // [channel][window row]
    float e[3][3];   // the same for the window's outer columns x0+cb-1 (held by lane 0) and x63+cb+1 (lane 63)
    int r0, cb;      // source row of window row 0; lane x owns source column x + cb
    bool live;
};

// ROWS = rows per wave: kBwdRows for launches that fill the chip, kBwdRowsSmall for small ones (a wave walks its rows one
// after the other, each a chain of load -> project -> gather -> (scatter): with few workgroups the chain length is the
// run time; 24x80 x 24 images with the image gradient: 30 us at 8 rows per wave)
template <int C, bool IMG = false, int ROWS = kBwdRows>
__global__ __launch_bounds__(kThreads) void warp_bwd_kernel(const float* __restrict__ K, const float* __restrict__ Kinv,
                                                            const float* __restrict__ pose,
                                                            const float* __restrict__ depth,
                                                            const float* __restrict__ src,
                                                            const float* __restrict__ g_warped, int Cdyn, WarpGeom G,
                                                            float* __restrict__ g_depth, float* __restrict__ g_src,
                                                            float* __restrict__ partials) {
    __shared__ float lds[kThreads / 64][12];
    const int n = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int bx = blockIdx.x % G.nbx, by = blockIdx.x / G.nbx;
    const int yb = (by * (kThreads / 64) + wave) * ROWS;
    const int x = bx * 64 + lane, xc = min(x, G.W - 1);
    const int Cc = C > 0 ? C : Cdyn;
    const unsigned HW = (unsigned)(G.H * G.W), HsWs = (unsigned)(G.Hs * G.Ws);
    Cam cam;
    load_cam(cam, K, Kinv, pose + 12 * n);
    const rsrc_t dp = make_rsrc(depth + (size_t)n * HW);
    const rsrc_t sp = make_rsrc(src + (size_t)n * Cc * HsWs);
    const rsrc_t gp = make_rsrc(g_warped + (size_t)n * Cc * HW);
    const rsrc_t op = make_rsrc(g_depth + (size_t)n * HW);
    const unsigned xb = (unsigned)xc * 4u;
    float gM[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) gM[i] = 0.f;
    // un-normalise * normalise factors of the grid gradient (grid_sampler backward, then synthesis.py:50-51)
    const float sx = (G.w1 * 0.5f) * (2.0f * G.rw1), sy = (G.h1 * 0.5f) * (2.0f * G.rh1);
    ImgWindow win;
    win.live = false;
    win.r0 = win.cb = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c)
        win.a[c][0] = win.a[c][1] = win.a[c][2] = win.e[c][0] = win.e[c][1] = win.e[c][2] = 0.f;
    // finished window rows go to memory: one coalesced atomic per texel and channel, one more for the two outer columns
    auto emit_row = [&](int slot) {
        const int r = win.r0 + slot, col = x + win.cb;
        if (r < 0 || r >= G.Hs) return;  // uniform
        float* q = g_src + (size_t)n * 3 * HsWs + (size_t)r * G.Ws;
        if (col >= 0 && col < G.Ws) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (win.a[c][slot] != 0.0f) atomicAdd(q + (size_t)c * HsWs + col, win.a[c][slot]);
        }
        if (lane == 0 || lane == 63) {
            const int ecol = col + (lane == 0 ? -1 : 1);  // inside the image whenever the sum is not zero
#pragma unroll
            for (int c = 0; c < 3; ++c)
                if (win.e[c][slot] != 0.0f) atomicAdd(q + (size_t)c * HsWs + ecol, win.e[c][slot]);
        }
    };
    auto clear_window = [&]() {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            win.a[c][0] = win.a[c][1] = win.a[c][2] = win.e[c][0] = win.e[c][1] = win.e[c][2] = 0.f;
    };
#pragma unroll 2
    for (int i = 0; i < ROWS; ++i) {
        const int y = yb + i;
        if (y >= G.H) break;  // wave-uniform
        const unsigned ro = (unsigned)(y * G.W);
        const float d = buf_load_f32(dp, xb, ro);
        float g[C > 0 ? C : 1];
        if (C > 0) {
#pragma unroll
            for (int c = 0; c < C; ++c) g[c] = buf_load_f32(gp, xb, c * HW + ro);
        }
        Proj pr;
        float rz;
        project_pixel_lean(cam, (float)xc, (float)y, d, pr, rz);
        Tap t;
        make_tap(pr.u, pr.v, G, t);
        float gix = 0.f, giy = 0.f;
        auto channel = [&](int c, float gc) {
            const float2_t tp = buf_load_f32x2(sp, t.o_top, c * HsWs * 4u);
            const float2_t bt = buf_load_f32x2(sp, t.o_bot, c * HsWs * 4u);
            // corner texels; a corner outside the image only ever meets a zero weight or a zeroed gradient below
            const float nw = t.shift ? tp.y : tp.x, ne = tp.y, sw = t.shift ? bt.y : bt.x, se = bt.y;
            gix = fmaf(gc, fmaf(ne - nw, t.sy, (se - sw) * t.ny), gix);
            giy = fmaf(gc, fmaf(sw - nw, t.ex, (se - ne) * t.wx), giy);
        };
        if (C > 0) {
#pragma unroll
            for (int c = 0; c < C; ++c) channel(c, g[c]);
        } else {
            for (int c = 0; c < Cc; ++c) channel(c, buf_load_f32(gp, xb, c * HW + ro));
        }
        if (IMG) {
            const bool valid = x < G.W;
            const int dx = t.xl - x;
            int rel = 0, rr = 0;
            bool lane_ok = false;
            // pixels of this row that do not fit the window (r0_, cb_): left texel in column x+cb-1 or x+cb, top row r0 or
            // r0+1.  A few misfits (near pixels among far ones) are scattered directly, lane by lane; more than
            // kMaxMisfits and the window is wrong for this row
            constexpr int kMaxMisfits = 12;
            auto misfits = [&](int r0_, int cb_) {
                rel = dx - (cb_ - 1);
                rr = t.y0 - r0_;
                lane_ok = valid && (unsigned)rel <= 1u && (unsigned)rr <= 1u;
                return __popcll(__ballot(valid && !lane_ok));
            };
            bool ok = false;
            if (win.live) {
                if (misfits(win.r0 + 1, win.cb) <= kMaxMisfits) {   // the usual case: the window moves down one row
                    emit_row(0);
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        win.a[c][0] = win.a[c][1];
                        win.a[c][1] = win.a[c][2];
                        win.a[c][2] = 0.f;
                        win.e[c][0] = win.e[c][1];
                        win.e[c][1] = win.e[c][2];
                        win.e[c][2] = 0.f;
                    }
                    win.r0 += 1;
                    ok = true;
                } else {
                    ok = misfits(win.r0, win.cb) <= kMaxMisfits;    // or stays
                }
            }
            if (!ok) {  // a new window around the tap of the wave's first pixel (always inside the image)
                const int d0 = __builtin_amdgcn_readfirstlane(dx), y0 = __builtin_amdgcn_readfirstlane(t.y0);
                int best = 65, br = 0, bc = 0;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int r0_ = y0 - (k & 1), cb_ = d0 + 1 - (k >> 1);
                    const int m = misfits(r0_, cb_);
                    if (m < best) { best = m; br = r0_; bc = cb_; }
                }
                if (best <= kMaxMisfits) {
                    if (win.live) {
                        emit_row(0);
                        emit_row(1);
                        emit_row(2);
                    }
                    clear_window();
                    win.r0 = br;
                    win.cb = bc;
                    win.live = true;
                    misfits(br, bc);
                    ok = true;
                }
            }
            bool direct = valid && !ok;  // pixels that go to memory with four atomics per channel
            if (ok) {
                // the pixel's four weights as a 3 (window rows) x 3 (owner columns x-1, x, x+1) stencil
                const float wat = lane_ok ? t.wa_t : 0.f, wbt = lane_ok ? t.wb_t : 0.f;
                const float wab = lane_ok ? t.wa_b : 0.f, wbb = lane_ok ? t.wb_b : 0.f;
                const bool r1 = rel != 0, d1 = rr != 0;
                const float ct = r1 ? wat : wbt, cm = r1 ? wab : wbb;        // own column: top, bottom
                const float rt = r1 ? wbt : 0.f, rm = r1 ? wbb : 0.f;        // column of the right neighbour
                const float lt = r1 ? 0.f : wat, lm = r1 ? 0.f : wab;        // column of the left neighbour
                const float kc[3] = {d1 ? 0.f : ct, d1 ? ct : cm, d1 ? cm : 0.f};
                const float kr[3] = {d1 ? 0.f : rt, d1 ? rt : rm, d1 ? rm : 0.f};
                const float kl[3] = {d1 ? 0.f : lt, d1 ? lt : lm, d1 ? lm : 0.f};
                float fromL[3], fromR[3];
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    fromL[s] = __shfl_up(kr[s], 1, 64);     // what lane x-1 sends to its right neighbour's column
                    fromR[s] = __shfl_down(kl[s], 1, 64);   // what lane x+1 sends to its left neighbour's column
                    if (lane == 0) fromL[s] = 0.f;
                    if (lane == 63) fromR[s] = 0.f;
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float gc = lane_ok ? g[c] : 0.f;
                    const float gl = __shfl_up(gc, 1, 64), gr = __shfl_down(gc, 1, 64);
#pragma unroll
                    for (int s = 0; s < 3; ++s) win.a[c][s] += fmaf(gr, fromR[s], fmaf(gl, fromL[s], gc * kc[s]));
                    // the outer columns of the window have no owner lane: the two edge lanes keep those sums
#pragma unroll
                    for (int s = 0; s < 3; ++s) win.e[c][s] = fmaf(gc, lane == 0 ? kl[s] : kr[s], win.e[c][s]);
                }
                // misfit pixels, one at a time: if the tap still lands on columns and rows the window holds, the two owner
                // lanes take it over (broadcast from the misfit lane); otherwise that lane scatters directly
                unsigned long long todo = __ballot(valid && !lane_ok);
                const int col = x + win.cb, col_first = bx * 64 + win.cb;
                while (todo) {
                    const int m = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const int mxl = __builtin_amdgcn_readlane(t.xl, m);
                    const int top = __builtin_amdgcn_readlane(t.y0, m) - win.r0;  // window row of its top texels
                    if (top >= 0 && top <= 1 && mxl >= col_first && mxl + 1 <= col_first + 63) {
                        const float mat = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t.wa_t), m));
                        const float mbt = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t.wb_t), m));
                        const float mab = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t.wa_b), m));
                        const float mbb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t.wb_b), m));
                        const float wt = col == mxl ? mat : (col == mxl + 1 ? mbt : 0.f);
                        const float wb = col == mxl ? mab : (col == mxl + 1 ? mbb : 0.f);
                        const float k0 = top == 0 ? wt : 0.f, k1 = top == 0 ? wb : wt, k2 = top == 0 ? 0.f : wb;
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float mg = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, g[c]), m));
                            win.a[c][0] = fmaf(mg, k0, win.a[c][0]);
                            win.a[c][1] = fmaf(mg, k1, win.a[c][1]);
                            win.a[c][2] = fmaf(mg, k2, win.a[c][2]);
                        }
                    } else if (lane == m) {
                        direct = true;
                    }
                }
            }
            if (direct) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float gc = g[c];
                    if (gc != 0.0f) {
                        float* q = g_src + ((size_t)n * 3 + c) * HsWs;
                        if (t.wa_t != 0.0f) atomicAdd(q + (t.o_top >> 2), gc * t.wa_t);
                        if (t.wb_t != 0.0f) atomicAdd(q + (t.o_top >> 2) + 1, gc * t.wb_t);
                        if (t.wa_b != 0.0f) atomicAdd(q + (t.o_bot >> 2), gc * t.wa_b);
                        if (t.wb_b != 0.0f) atomicAdd(q + (t.o_bot >> 2) + 1, gc * t.wb_b);
                    }
                }
            }
        } else if (g_src && x < G.W) {  // generic channel count: float atomics straight to memory
            Samp s;
            sample_setup(pr.u, pr.v, G.Hs, G.Ws, s);
            Corners co;
            corner_offsets(s, G.Hs, G.Ws, co);
            for (int c = 0; c < Cc; ++c)
                scatter_src_grad(g_src + ((size_t)n * Cc + c) * HsWs, co, s, buf_load_f32(gp, xb, c * HW + ro));
        }
        // torch grid_sampler_2d backward: zero on or outside the border of the un-clipped coordinate
        gix = (t.ixu > 0.0f && t.ixu < G.w1) ? gix * sx : 0.0f;
        giy = (t.iyu > 0.0f && t.iyu < G.h1) ? giy * sy : 0.0f;
        if (x >= G.W) gix = giy = 0.0f;  // lanes beyond the image contribute nothing to the pose gradient
        // autograd of project_pixel (see project_bwd_pixel); 1/z from the forward's refined reciprocal
        const float gq0 = gix * rz, gq1 = giy * rz;
        const float gq2 = (fabsf(pr.q[2]) < 1e-6f) ? 0.0f : -(fmaf(pr.u, gix, pr.v * giy)) * rz;
        float gT[3], gP[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) gT[k] = fmaf(cam.K[6 + k], gq2, fmaf(cam.K[3 + k], gq1, cam.K[k] * gq0));
#pragma unroll
        for (int k = 0; k < 3; ++k) gP[k] = fmaf(cam.M[8 + k], gT[2], fmaf(cam.M[4 + k], gT[1], cam.M[k] * gT[0]));
        const float gd = fmaf(pr.c[2], gP[2], fmaf(pr.c[1], gP[1], pr.c[0] * gP[0]));
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            gM[4 * j + 0] = fmaf(gT[j], pr.P[0], gM[4 * j + 0]);
            gM[4 * j + 1] = fmaf(gT[j], pr.P[1], gM[4 * j + 1]);
            gM[4 * j + 2] = fmaf(gT[j], pr.P[2], gM[4 * j + 2]);
            gM[4 * j + 3] += gT[j];
        }
        if (x < G.W) buf_store_f32(gd, op, (unsigned)x * 4u, ro * 4u);
    }
    if (IMG && win.live) {
        emit_row(0);
        emit_row(1);
        emit_row(2);
    }
    block_reduce12(gM, partials + ((size_t)n * gridDim.x + blockIdx.x) * 12, lds);
}

// backward of reproject_from_depth alone: g_coords (N,H,W,2) -> g_depth, pose partials
__global__ __launch_bounds__(kThreads) void reproject_bwd_kernel(const float* __restrict__ K, const float* __restrict__ Kinv,
                                                                 const float* __restrict__ pose,
                                                                 const float* __restrict__ depth,
                                                                 const float* __restrict__ g_coords, int H, int W,
                                                                 float* __restrict__ g_depth, float* __restrict__ partials) {
    __shared__ float lds[kThreads / 64][12];
    const int n = blockIdx.y;
    const int HW = H * W;
    Cam cam;
    load_cam(cam, K, Kinv, pose + 12 * n);
    float gM[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) gM[i] = 0.f;
#pragma unroll
    for (int it = 0; it < kBwdPix; ++it) {
        const int p = (blockIdx.x * kBwdPix + it) * kThreads + threadIdx.x;
        if (p < HW) {
            const int y = p / W, x = p - y * W;
            Proj pr;
            project_pixel(cam, (float)x, (float)y, depth[(size_t)n * HW + p], pr);
            const float2 g = reinterpret_cast<const float2*>(g_coords)[(size_t)n * HW + p];
            float gd;
            project_bwd_pixel(cam, pr, g.x, g.y, gd, gM);
            g_depth[(size_t)n * HW + p] = gd;
        }
    }
    block_reduce12(gM, partials + ((size_t)n * gridDim.x + blockIdx.x) * 12, lds);
}

// ---------------------------------------------------------------------------------------------------
// a1: synthesis.construct_pose_batch (synthesis.py:6-33) and its autograd, one thread per pose
// ---------------------------------------------------------------------------------------------------
__global__ void pose_from_vec_fwd_kernel(const float* __restrict__ vec, int N, float* __restrict__ pose) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* r = vec + 6 * n;
    const float th = sqrtf((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
    const float den = th + 1e-6f;
    const float u0 = r[0] / den, u1 = r[1] / den, u2 = r[2] / den;
    const float U[9] = {0.f, -u2, u1, u2, 0.f, -u0, -u1, u0, 0.f};
    const float s = sinf(th), c1 = 1.0f - cosf(th);
    float* P = pose + 12 * n;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float uu = U[3 * i] * U[j];
            uu = fmaf(U[3 * i + 1], U[3 + j], uu);
            uu = fmaf(U[3 * i + 2], U[6 + j], uu);
            P[4 * i + j] = ((i == j ? 1.0f : 0.0f) + s * U[3 * i + j]) + c1 * uu;
        }
        P[4 * i + 3] = r[3 + i];
    }
}

__global__ void pose_from_vec_bwd_kernel(const float* __restrict__ vec, const float* __restrict__ g_pose, int N,
                                         float* __restrict__ g_vec) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float* r = vec + 6 * n;
    const float* G4 = g_pose + 12 * n;
    const float th = sqrtf((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
    const float den = th + 1e-6f;
    const float u[3] = {r[0] / den, r[1] / den, r[2] / den};
    const float U[9] = {0.f, -u[2], u[1], u[2], 0.f, -u[0], -u[1], u[0], 0.f};
    const float s = sinf(th), c = cosf(th), c1 = 1.0f - c;
    float G[9], UU[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            G[3 * i + j] = G4[4 * i + j];
            UU[3 * i + j] = U[3 * i] * U[j] + U[3 * i + 1] * U[3 + j] + U[3 * i + 2] * U[6 + j];
        }
    float g_s = 0.f, g_c1 = 0.f;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        g_s += G[i] * U[i];
        g_c1 += G[i] * UU[i];
    }
    // dL/dU = s*G + c1*(G U^T + U^T G)
    float gU[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) a += G[3 * i + k] * U[3 * j + k] + U[3 * k + i] * G[3 * k + j];
            gU[3 * i + j] = s * G[3 * i + j] + c1 * a;
        }
    const float gu[3] = {gU[7] - gU[5], gU[2] - gU[6], gU[3] - gU[1]};
    float g_th = c * g_s + s * g_c1;
    float gr[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        gr[k] = gu[k] / den;
        g_th -= gu[k] * r[k] / (den * den);
    }
    if (th > 0.f) {
#pragma unroll
        for (int k = 0; k < 3; ++k) gr[k] += g_th * r[k] / th;
    }
    float* o = g_vec + 6 * n;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = gr[k];
        o[3 + k] = G4[4 * k + 3];
    }
}

// partials (N, nblk, 12) -> g_pose (N,12): fixed summation order, double accumulator
__global__ void pose_finalize_kernel(const float* __restrict__ partials, int nblk, float* __restrict__ g_pose) {
    pose_finalize_wave(partials, blockIdx.x, nblk, g_pose, threadIdx.x);  // one 64-lane wave per pose
}

inline int pix_blocks(int HW) { return (HW + kThreads - 1) / kThreads; }
inline int bwd_blocks(int HW) { return (HW + kThreads * kBwdPix - 1) / (kThreads * kBwdPix); }

inline int warp_tiles(int H, int W, int rows) {  // blocks per image: 64 columns x (4 waves x rows) rows each
    const int per_block = (kThreads / 64) * rows;
    return ((W + 63) / 64) * ((H + per_block - 1) / per_block);
}

inline bool bad_dims(int N, int C, int Hs, int Ws, int H, int W) {
    return N <= 0 || C <= 0 || Hs < 2 || Ws < 2 || H <= 0 || W <= 0 || N > 65535 ||
           (long long)H * W > (1ll << 30) || (long long)Hs * Ws > (1ll << 30) ||
           // the tiled kernels address the C planes of one image with 32-bit byte offsets
           (long long)C * H * W * 4 >= (1ll << 32) || (long long)C * Hs * Ws * 4 >= (1ll << 32);
}

}  // namespace

extern "C" {

int ds_reproject(const float* K, const float* Kinv, const float* pose, const float* depth, int N, int H, int W,
                 float* coords, void* stream) {
    if (!K || !Kinv || !pose || !depth || !coords || bad_dims(N, 1, 2, 2, H, W)) return DS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(reproject_kernel, dim3(pix_blocks(H * W), N), dim3(kThreads), 0, (hipStream_t)stream, K, Kinv, pose,
                       depth, H, W, coords);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

int ds_grid_sample_fwd(const float* src, float* grid, int N, int C, int Hs, int Ws, int H, int W, float* out,
                       void* stream) {
    if (!src || !grid || !out || bad_dims(N, C, Hs, Ws, H, W)) return DS_ERR_INVALID_ARG;
    dim3 g(pix_blocks(H * W), N), b(kThreads);
    if (C == 3)
        hipLaunchKernelGGL(grid_sample_fwd_kernel<3>, g, b, 0, (hipStream_t)stream, src, grid, C, Hs, Ws, H, W, out);
    else
        hipLaunchKernelGGL(grid_sample_fwd_kernel<0>, g, b, 0, (hipStream_t)stream, src, grid, C, Hs, Ws, H, W, out);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

int ds_grid_sample_bwd(const float* src, const float* grid_norm, const float* g_out, int N, int C, int Hs, int Ws,
                       int H, int W, float* g_grid, float* g_src, void* stream) {
    if (!src || !grid_norm || !g_out || !g_grid || bad_dims(N, C, Hs, Ws, H, W)) return DS_ERR_INVALID_ARG;
    dim3 g(pix_blocks(H * W), N), b(kThreads);
    if (C == 3)
        hipLaunchKernelGGL(grid_sample_bwd_kernel<3>, g, b, 0, (hipStream_t)stream, src, grid_norm, g_out, C, Hs, Ws, H, W,
                           g_grid, g_src);
    else
        hipLaunchKernelGGL(grid_sample_bwd_kernel<0>, g, b, 0, (hipStream_t)stream, src, grid_norm, g_out, C, Hs, Ws, H, W,
                           g_grid, g_src);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

int ds_warp_fwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src, int N,
                int C, int Hs, int Ws, int H, int W, float* warped, void* stream) {
    if (!K || !Kinv || !pose || !depth || !src || !warped || bad_dims(N, C, Hs, Ws, H, W)) return DS_ERR_INVALID_ARG;
    const WarpGeom G = warp_geom(Hs, Ws, H, W);
    dim3 g(warp_tiles(H, W, kFwdRows), N), b(kThreads);
    if (C == 3)
        hipLaunchKernelGGL(warp_fwd_kernel<3>, g, b, 0, (hipStream_t)stream, K, Kinv, pose, depth, src, C, G, warped);
    else
        hipLaunchKernelGGL(warp_fwd_kernel<0>, g, b, 0, (hipStream_t)stream, K, Kinv, pose, depth, src, C, G, warped);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

size_t ds_warp_bwd_workspace_bytes(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    const int a = bwd_blocks(H * W), b = warp_tiles(H, W, kBwdRowsSmall);  // ds_reproject_bwd / ds_warp_bwd partials (either tiling)
    return (size_t)N * (a > b ? a : b) * 12 * sizeof(float);
}

int ds_warp_bwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                const float* g_warped, int N, int C, int Hs, int Ws, int H, int W, float* g_depth, float* g_pose,
                float* g_src, void* workspace, size_t workspace_bytes, void* stream) {
    if (!g_pose) return DS_ERR_INVALID_ARG;
    int nblk = 0;
    int rc = ds::warp_bwd_partials(K, Kinv, pose, depth, src, g_warped, N, C, Hs, Ws, H, W, g_depth, g_src, workspace,
                                   workspace_bytes, &nblk, (hipStream_t)stream);
    if (rc != DS_OK) return rc;
    hipLaunchKernelGGL(pose_finalize_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, (const float*)workspace, nblk, g_pose);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

int ds_reproject_bwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* g_coords,
                     int N, int H, int W, float* g_depth, float* g_pose, void* workspace, size_t workspace_bytes,
                     void* stream) {
    if (!K || !Kinv || !pose || !depth || !g_coords || !g_depth || !g_pose || bad_dims(N, 1, 2, 2, H, W))
        return DS_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < ds_warp_bwd_workspace_bytes(N, H, W)) return DS_ERR_WORKSPACE;
    const int nblk = bwd_blocks(H * W);
    float* partials = (float*)workspace;
    hipLaunchKernelGGL(reproject_bwd_kernel, dim3(nblk, N), dim3(kThreads), 0, (hipStream_t)stream, K, Kinv, pose, depth,
                       g_coords, H, W, g_depth, partials);
    DS_CHECK_LAUNCH();
    hipLaunchKernelGGL(pose_finalize_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, partials, nblk, g_pose);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

int ds_pose_from_vec_fwd(const float* vec, int N, float* pose, void* stream) {
    if (!vec || !pose || N <= 0) return DS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(pose_from_vec_fwd_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, vec, N, pose);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

int ds_pose_from_vec_bwd(const float* vec, const float* g_pose, int N, float* g_vec, void* stream) {
    if (!vec || !g_pose || !g_vec || N <= 0) return DS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(pose_from_vec_bwd_kernel, dim3((N + 63) / 64), dim3(64), 0, (hipStream_t)stream, vec, g_pose, N, g_vec);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

}  // extern "C"

int ds::warp_bwd_partials(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                          const float* g_warped, int N, int C, int Hs, int Ws, int H, int W, float* g_depth, float* g_src,
                          void* workspace, size_t workspace_bytes, int* nblk_out, hipStream_t stream) {
    if (!K || !Kinv || !pose || !depth || !src || !g_warped || !g_depth || bad_dims(N, C, Hs, Ws, H, W))
        return DS_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < ds_warp_bwd_workspace_bytes(N, H, W)) return DS_ERR_WORKSPACE;
    const WarpGeom G = warp_geom(Hs, Ws, H, W);
    // small launches (fewer workgroups than would give every SIMD two waves) take the short tiles; only the 3-channel
    // kernels are instantiated for them (the generic-C kernel keeps kBwdRows, and its grid with it)
    const bool small = C == 3 && (long long)warp_tiles(H, W, kBwdRows) * N < 2ll * ds::cu_count();
    const int nblk = warp_tiles(H, W, small ? kBwdRowsSmall : kBwdRows);
    dim3 g(nblk, N), b(kThreads);
    float* partials = (float*)workspace;
    if (C == 3 && g_src) {
        if (small)
            hipLaunchKernelGGL((warp_bwd_kernel<3, true, kBwdRowsSmall>), g, b, 0, stream, K, Kinv, pose, depth, src, g_warped, C,
                               G, g_depth, g_src, partials);
        else
            hipLaunchKernelGGL((warp_bwd_kernel<3, true>), g, b, 0, stream, K, Kinv, pose, depth, src, g_warped, C, G, g_depth,
                               g_src, partials);
    } else if (C == 3) {
        if (small)
            hipLaunchKernelGGL((warp_bwd_kernel<3, false, kBwdRowsSmall>), g, b, 0, stream, K, Kinv, pose, depth, src, g_warped,
                               C, G, g_depth, nullptr, partials);
        else
            hipLaunchKernelGGL(warp_bwd_kernel<3>, g, b, 0, stream, K, Kinv, pose, depth, src, g_warped, C, G, g_depth, nullptr,
                               partials);
    } else {
        hipLaunchKernelGGL(warp_bwd_kernel<0>, g, b, 0, stream, K, Kinv, pose, depth, src, g_warped, C, G, g_depth, g_src,
                           partials);
    }
    DS_CHECK_LAUNCH();
    *nblk_out = nblk;
    return DS_OK;
}
