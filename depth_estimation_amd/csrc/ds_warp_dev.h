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
