// ds_warp_dev.h -- device helpers of the warp path shared by ds_warp.hip and the plane sweep in ds_misc.hip.
#pragma once
#include "ds_common.h"

namespace ds {

struct WarpGeom {     // uniform launch geometry, filled by the host
    int Hs, Ws, H, W, nbx;
    float w1, h1;     // (float)(Ws-1), (float)(Hs-1)
    float rw1, rh1;   // their correctly rounded reciprocals
};

struct Tap {
    float ixu, iyu;                 // un-normalised, un-clipped coordinates (border rule of the backward)
    float wx, ex, ny, sy;           // fractional weights
    float wa_t, wb_t, wa_b, wb_b;   // weights of the loaded pairs: (a,b) of the top row, (a,b) of the bottom row
    unsigned o_top, o_bot;          // byte offset of the left texel of each pair inside one source plane
    bool shift;                     // x0 == Ws-1: the pair starts one texel to the left
    int xl, y0, y1;                 // texel coordinates of the pairs (column of the left texel, the two rows)
};

// synthesis.synthesize's coordinate handling for one pixel (synthesis.py:49-54): normalise with the SOURCE dims,
// un-normalise as grid_sample(align_corners=True), clip to the border, bilinear weights.  Same arithmetic as
// ds_common.h sample_setup; the out-of-range corners get weight selection instead of value selection:
//   x0 == Ws-1  => ix == Ws-1 exactly, wx == 0: the pair is loaded one texel to the left and its second texel takes
//                  the nw/sw weight (the first gets 0, 0*texel + ... is exact for finite texels);
//   y0 == Hs-1  => ny == 0: the bottom pair re-reads the top row with weights that are already 0.
__device__ __forceinline__ void make_tap(float u, float v, const WarpGeom& G, Tap& t) {
    const float gx = div_by(u, G.w1, G.rw1) * 2.0f - 1.0f;
    const float gy = div_by(v, G.h1, G.rh1) * 2.0f - 1.0f;
    float ix = (gx + 1.0f) * (G.w1 * 0.5f);
    float iy = (gy + 1.0f) * (G.h1 * 0.5f);
    t.ixu = ix;
    t.iyu = iy;
    ix = fminf(G.w1, fmaxf(ix, 0.0f));
    iy = fminf(G.h1, fmaxf(iy, 0.0f));
    const float xf = floorf(ix), yf = floorf(iy);
    const int x0 = (int)xf, y0 = (int)yf;
    t.wx = ix - xf;
    t.ex = 1.0f - t.wx;
    t.ny = iy - yf;
    t.sy = 1.0f - t.ny;
    const float w_nw = t.sy * t.ex, w_ne = t.sy * t.wx, w_sw = t.ny * t.ex, w_se = t.ny * t.wx;
    t.shift = x0 + 1 >= G.Ws;
    const int xc = t.shift ? x0 - 1 : x0;          // Ws >= 2 (checked on the host)
    const int y1 = y0 + 1 < G.Hs ? y0 + 1 : y0;
    t.xl = xc; t.y0 = y0; t.y1 = y1;
    t.o_top = (unsigned)(y0 * G.Ws + xc) * 4u;
    t.o_bot = (unsigned)(y1 * G.Ws + xc) * 4u;
    t.wa_t = t.shift ? 0.0f : w_nw;
    t.wb_t = t.shift ? w_nw : w_ne;
    t.wa_b = t.shift ? 0.0f : w_sw;
    t.wb_b = t.shift ? w_sw : w_se;
}

// project_pixel with the two divisions sharing one refined reciprocal (kept in rz for the backward)
__device__ __forceinline__ void project_pixel_lean(const Cam& cam, float fx, float fy, float d, Proj& o, float& rz) {
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
    rz = rcp_refined(o.z);
    o.u = div_by(o.q[0], o.z, rz);
    o.v = div_by(o.q[1], o.z, rz);
}

#ifndef DS_FWD_ROWS
#define DS_FWD_ROWS 4
#endif
constexpr int kFwdRows = DS_FWD_ROWS;  // rows per wave: kFwdRows depth loads, then kFwdRows x C x 2 corner-pair gathers in flight

// One tile of ds_warp_fwd: image n, tile index `tile` (64 columns x 4 waves x kFwdRows rows), 256 threads.  A device
// function so that the forward loss step can run these tiles next to other work in one launch (ds_photo.hip).
template <int C>
__device__ __forceinline__ void warp_fwd_tile(const float* __restrict__ K, const float* __restrict__ Kinv,
                                              const float* __restrict__ pose, const float* __restrict__ depth,
                                              const float* __restrict__ src, int Cdyn, const WarpGeom& G,
                                              float* __restrict__ warped, int n, int tile) {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int bx = tile % G.nbx, by = tile / G.nbx;
    const int yb = (by * 4 + wave) * kFwdRows;
    if (yb >= G.H) return;
    const int x = bx * 64 + lane, xc = min(x, G.W - 1);
    const int Cc = C > 0 ? C : Cdyn;
    const unsigned HW = (unsigned)(G.H * G.W), HsWs = (unsigned)(G.Hs * G.Ws);
    Cam cam;
    load_cam(cam, K, Kinv, pose + 12 * n);
    const rsrc_t dp = make_rsrc(depth + (size_t)n * HW);
    const rsrc_t sp = make_rsrc(src + (size_t)n * Cc * HsWs);
    const rsrc_t op = make_rsrc(warped + (size_t)n * Cc * HW);
    const unsigned xb = (unsigned)xc * 4u;
    float d[kFwdRows];
#pragma unroll
    for (int i = 0; i < kFwdRows; ++i) d[i] = buf_load_f32(dp, xb, (unsigned)(min(yb + i, G.H - 1) * G.W));
    Tap t[kFwdRows];
#pragma unroll
    for (int i = 0; i < kFwdRows; ++i) {
        Proj pr;
        float rz;
        project_pixel_lean(cam, (float)xc, (float)min(yb + i, G.H - 1), d[i], pr, rz);
        make_tap(pr.u, pr.v, G, t[i]);
    }
    if (C > 0) {
        float2_t top[kFwdRows][C > 0 ? C : 1], bot[kFwdRows][C > 0 ? C : 1];
#pragma unroll
        for (int i = 0; i < kFwdRows; ++i)
#pragma unroll
            for (int c = 0; c < C; ++c) {
                top[i][c] = buf_load_f32x2(sp, t[i].o_top, c * HsWs * 4u);
                bot[i][c] = buf_load_f32x2(sp, t[i].o_bot, c * HsWs * 4u);
            }
#pragma unroll
        for (int i = 0; i < kFwdRows; ++i) {
            if (x < G.W && yb + i < G.H) {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    float acc = top[i][c].x * t[i].wa_t;
                    acc = fmaf(top[i][c].y, t[i].wb_t, acc);
                    acc = fmaf(bot[i][c].x, t[i].wa_b, acc);
                    acc = fmaf(bot[i][c].y, t[i].wb_b, acc);
                    buf_store_f32(acc, op, (unsigned)x * 4u, (c * HW + (unsigned)((yb + i) * G.W)) * 4u);
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < kFwdRows; ++i) {
            if (x < G.W && yb + i < G.H) {
                for (int c = 0; c < Cc; ++c) {
                    const float2_t tp = buf_load_f32x2(sp, t[i].o_top, c * HsWs * 4u);
                    const float2_t bt = buf_load_f32x2(sp, t[i].o_bot, c * HsWs * 4u);
                    float acc = tp.x * t[i].wa_t;
                    acc = fmaf(tp.y, t[i].wb_t, acc);
                    acc = fmaf(bt.x, t[i].wa_b, acc);
                    acc = fmaf(bt.y, t[i].wb_b, acc);
                    buf_store_f32(acc, op, (unsigned)x * 4u, (c * HW + (unsigned)((yb + i) * G.W)) * 4u);
                }
            }
        }
    }
}

// host side of WarpGeom
inline WarpGeom warp_geom(int Hs, int Ws, int H, int W) {
    WarpGeom G;
    G.Hs = Hs; G.Ws = Ws; G.H = H; G.W = W;
    G.nbx = (W + 63) / 64;
    G.w1 = (float)(Ws - 1); G.h1 = (float)(Hs - 1);
    G.rw1 = 1.0f / G.w1; G.rh1 = 1.0f / G.h1;
    return G;
}

}  // namespace ds
