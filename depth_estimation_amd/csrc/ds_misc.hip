// ds_misc.hip -- plane sweep / cost volume / soft-argmin depth, edge-aware smoothness, depth metrics.
//
// Reference: synthesis.synthesize_from_depths (synthesis.py:57-80), metrics.cost_volumes (metrics.py:5-18),
// synthesis.depth_from_costs (synthesis.py:119-138), metrics.regularization_term (metrics.py:75-97),
// metrics.rmse/rmsle/abs_rel/sq_rel (metrics.py:100-153).
#include "ds_common.h"
#include "ds_warp_dev.h"
#include "ds_internal.h"
#include <algorithm>
#include <mutex>
#include "depthsynth.h"

using namespace ds;

namespace {

constexpr int kThreads = 256;
constexpr int kMaxPlanes = 128;

struct Planes { float d[kMaxPlanes]; };  // fp32 plane depths (torch.ones(h,w)*depth is fp32, synthesis.py:69)

inline int pix_blocks(int HW) { return (HW + kThreads - 1) / kThreads; }

// ---------------------------------------------------------------------------------------------------
// a4: D constant-depth warps of one source, materialised (D,C,H,W)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void plane_warp_kernel(const float* __restrict__ K, const float* __restrict__ Kinv,
                                                              const float* __restrict__ pose, Planes planes,
                                                              const float* __restrict__ src, int C, int Hs, int Ws, int H,
                                                              int W, float* __restrict__ out) {
    const int k = blockIdx.y;
    const int p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= H * W) return;
    Cam cam;
    load_cam(cam, K, Kinv, pose);
    const int y = p / W, x = p - y * W;
    Proj pr;
    project_pixel(cam, (float)x, (float)y, planes.d[k], pr);
    Samp s;
    sample_setup(pr.u, pr.v, Hs, Ws, s);
    Corners co;
    corner_offsets(s, Hs, Ws, co);
    float* op = out + (size_t)k * C * H * W + p;
    for (int c = 0; c < C; ++c) {
        float nw, ne, sw, se;
        gather4(src + (size_t)c * Hs * Ws, co, nw, ne, sw, se);
        op[(size_t)c * H * W] = bilinear(nw, ne, sw, se, s);
    }
}

// a5: cost (D,H,W) = mean_c |t - s|, channel sum in order then a correctly rounded /C
__global__ __launch_bounds__(kThreads) void cost_volume_kernel(const float* __restrict__ target,
                                                               const float* __restrict__ synth, int C, int HW,
                                                               float* __restrict__ cost) {
    const int k = blockIdx.y;
    const int p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= HW) return;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) {
        float a = fabsf(target[(size_t)c * HW + p] - synth[((size_t)k * C + c) * HW + p]);
        acc = (c == 0) ? a : acc + a;
    }
    cost[(size_t)k * HW + p] = (C == 3) ? div3(acc) : acc / (float)C;
}

__device__ __forceinline__ float index_to_depth(float m, int D, int sid) {
    const float ln2 = 0.693147180559945f;  // torch.log(torch.tensor(2.0))
    return sid ? (expf((m + 1.0f) * ln2 / (float)D) - 1.0f) : (m / (float)D + 1.0f / (float)D);
}

// a6: two passes over the D costs of a pixel (softmin = softmax of the negated, max-shifted values)
__global__ __launch_bounds__(kThreads) void depth_from_costs_kernel(const float* __restrict__ cost, int D, int HW, int sid,
                                                                    float* __restrict__ depth) {
    const int n = blockIdx.y;
    const int p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= HW) return;
    const float* c = cost + (size_t)n * D * HW + p;
    float mn = INFINITY;
    for (int k = 0; k < D; ++k) {
        float e = expf(c[(size_t)k * HW]);
        mn = fminf(mn, e * e - 1.0f);
    }
    float den = 0.f, num = 0.f;
    for (int k = 0; k < D; ++k) {
        float e = expf(c[(size_t)k * HW]);
        float w = expf(mn - (e * e - 1.0f));
        den += w;
        num = fmaf(w, (float)k, num);
    }
    depth[(size_t)n * HW + p] = index_to_depth(num / den, D, sid);
}

// fused a4+a5+a6: one thread per (source, pixel), online softmin over the planes; nothing materialised.
// In a real train step this is 20x the warped pixels of the loss path (S x 80 planes at quarter resolution), and it is
// VALU-bound, so the per-plane work is written for instruction count like the warp kernels (ds_warp_dev.h): a wave
// owns 64 adjacent columns of one row (no integer division), the ray K^-1 [x,y,1] is formed once per pixel, the
// exact divides share one refined reciprocal per denominator or take it from the host, the border cases of the
// bilinear gather select weights once per plane instead of texel values per channel, buffer loads carry the image
// base in an SGPR descriptor, and the online softmin needs one exponential per plane instead of two.
template <int C>
__global__ __launch_bounds__(kThreads) void plane_sweep_depth_kernel(const float* __restrict__ K,
                                                                     const float* __restrict__ Kinv,
                                                                     const float* __restrict__ pose, Planes planes, int D,
                                                                     const float* __restrict__ src,
                                                                     const float* __restrict__ target, int per_target,
                                                                     int Cdyn, WarpGeom G, int sid,
                                                                     float* __restrict__ depth) {
    const int n = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int bx = blockIdx.x % G.nbx, by = blockIdx.x / G.nbx;
    const int y = by * (kThreads / 64) + wave;
    if (y >= G.H) return;
    const int x = bx * 64 + lane, xc = min(x, G.W - 1);
    const int Cc = C > 0 ? C : Cdyn;
    const unsigned HW = (unsigned)(G.H * G.W), HsWs = (unsigned)(G.Hs * G.Ws);
    Cam cam;
    load_cam(cam, K, Kinv, pose + 12 * n);
    const rsrc_t sp = make_rsrc(src + (size_t)n * Cc * HsWs);
    const rsrc_t tp = make_rsrc(target + (size_t)(n / per_target) * Cc * HW);  // uniform: n is blockIdx.y
    const unsigned po = (unsigned)xc * 4u, ro = (unsigned)(y * G.W);
    // the ray of this pixel (synthesis.py:100), shared by all planes
    float ray[3];
    {
        const float fx = (float)xc, fy = (float)y;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float acc = fx * cam.Kinv[3 * j + 0];
            acc = fmaf(fy, cam.Kinv[3 * j + 1], acc);
            ray[j] = fmaf(1.0f, cam.Kinv[3 * j + 2], acc);
        }
    }
    float tg[C > 0 ? C : 1];
    if (C > 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) tg[c] = buf_load_f32(tp, po, c * HW + ro);
    }
    const float w1m = G.w1 - 1.0f, h1m = G.h1 - 1.0f;   // Ws-2, Hs-2 (both >= 0: checked on the host)
    float run_min = INFINITY, den = 0.f, num = 0.f;
    for (int k = 0; k < D; ++k) {
        const float d = planes.d[k];
        float P[3], T[3], q[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) P[j] = ray[j] * d;                                   // synthesis.py:105
#pragma unroll
        for (int j = 0; j < 3; ++j) {                                                    // :108-109
            float acc = P[0] * cam.M[4 * j + 0];
            acc = fmaf(P[1], cam.M[4 * j + 1], acc);
            acc = fmaf(P[2], cam.M[4 * j + 2], acc);
            T[j] = fmaf(1.0f, cam.M[4 * j + 3], acc);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {                                                    // :112
            float acc = T[0] * cam.K[3 * j + 0];
            acc = fmaf(T[1], cam.K[3 * j + 1], acc);
            q[j] = fmaf(T[2], cam.K[3 * j + 2], acc);
        }
        const float z = (fabsf(q[2]) < 1e-6f) ? 1e-6f : q[2];                            // :113
        const float rz = rcp_refined(z);
        // :114, :49-54 -- make_tap's arithmetic with the border cases folded into the floor instead of selected afterwards:
        // a coordinate clipped onto the last column (row) has floor = Ws-1 (Hs-1) and fraction 0; taking Ws-2 (Hs-2) with
        // fraction exactly 1 instead loads the same two texels in the order (left, right) = (weight 0, full weight) and gives
        // the same products, so no shifted pair, no second row select and the bottom pair always one row down
        // (profiles/r03_kb3_plane_sweep.txt: 137 -> 128 VALU instructions per plane and pixel)
        const float u = div_by(q[0], z, rz), v_ = div_by(q[1], z, rz);
        const float gx = div_by(u, G.w1, G.rw1) * 2.0f - 1.0f;
        const float gy = div_by(v_, G.h1, G.rh1) * 2.0f - 1.0f;
        const float ix = fminf(G.w1, fmaxf((gx + 1.0f) * (G.w1 * 0.5f), 0.0f));
        const float iy = fminf(G.h1, fmaxf((gy + 1.0f) * (G.h1 * 0.5f), 0.0f));
        const float xf = fminf(floorf(ix), w1m), yf = fminf(floorf(iy), h1m);
        const float wx = ix - xf, ex = 1.0f - wx, ny = iy - yf, sy = 1.0f - ny;
        const float wa_t = sy * ex, wb_t = sy * wx, wa_b = ny * ex, wb_b = ny * wx;
        const unsigned o_top = (unsigned)((int)yf * G.Ws + (int)xf) * 4u, o_bot = o_top + (unsigned)G.Ws * 4u;
        float acc = 0.f;
        auto channel = [&](int c, float tv) {
            const float2_t top = buf_load_f32x2(sp, o_top, c * HsWs * 4u);
            const float2_t bot = buf_load_f32x2(sp, o_bot, c * HsWs * 4u);
            float w = top.x * wa_t;
            w = fmaf(top.y, wb_t, w);
            w = fmaf(bot.x, wa_b, w);
            w = fmaf(bot.y, wb_b, w);
            const float a = fabsf(tv - w);                                               // metrics.py:14-18
            acc = (c == 0) ? a : acc + a;
        };
        if (C > 0) {
#pragma unroll
            for (int c = 0; c < C; ++c) channel(c, tg[c]);
        } else {
            for (int c = 0; c < Cc; ++c) channel(c, buf_load_f32(tp, po, c * HW + ro));
        }
        const float cst = (C == 3) ? div3(acc) : acc / (float)Cc;
        const float e = expf(cst);
        const float v = e * e - 1.0f;                                                    // synthesis.py:127
        // online softmin: with m = min(run_min, v) the old sums scale by exp(m - run_min), the new term by
        // exp(m - v); one of the two factors is 1, the other exp(-|v - run_min|) (exp(-inf) = 0 on the first plane)
        const float f = __builtin_amdgcn_exp2f(fabsf(v - run_min) * -1.4426950408889634f);
        const bool lower = v < run_min;
        const float so = lower ? f : 1.0f, sn = lower ? 1.0f : f;
        den = fmaf(den, so, sn);
        num = fmaf(num, so, sn * (float)k);
        run_min = fminf(run_min, v);
    }
    if (x < G.W) depth[(size_t)n * HW + ro + x] = index_to_depth(num / den, D, sid);
}

// ---------------------------------------------------------------------------------------------------
// a10: edge-aware smoothness
// ---------------------------------------------------------------------------------------------------
// torch.gradient with unit spacing along a line of length L and element stride `st`
__device__ __forceinline__ float grad1d(const float* __restrict__ f, int i, int L, int st) {
    if (L == 1) return 0.f;
    if (i == 0) return f[st] - f[0];
    if (i == L - 1) return f[(L - 1) * st] - f[(L - 2) * st];
    return (f[(i + 1) * st] - f[(i - 1) * st]) * 0.5f;
}

__device__ __forceinline__ float block_sum(float v, float* lds) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += lds[w];
    return r;  // valid on thread 0
}

__global__ __launch_bounds__(kThreads) void smooth_fwd_kernel(const float* __restrict__ depths,
                                                              const float* __restrict__ targets, int S, int C, int H,
                                                              int W, float* __restrict__ partials) {
    __shared__ float lds[kThreads / 64];
    const int n = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * kThreads + threadIdx.x;
    float term = 0.f;
    if (p < HW) {
        const int y = p / W, x = p - y * W;
        const float* D = depths + (size_t)n * HW;
        const float adx = fabsf(grad1d(D + y * W, x, W, 1)), ady = fabsf(grad1d(D + x, y, H, W));
        float acc = 0.f;
        for (int c = 0; c < C; ++c) {
            const float* I = targets + ((size_t)(n / S) * C + c) * HW;
            const float ex = expf(fabsf(grad1d(I + y * W, x, W, 1)));
            const float ey = expf(fabsf(grad1d(I + x, y, H, W)));
            const float t = adx / ex + ady / ey;
            acc = (c == 0) ? t : acc + t;
        }
        term = acc / (float)C;
    }
    const float s = block_sum(term, lds);
    if (threadIdx.x == 0) partials[(size_t)n * gridDim.x + blockIdx.x] = s;
}

constexpr int kSumThreads = 1024;  // one workgroup; fixed summation order (deterministic)

__global__ __launch_bounds__(kSumThreads) void sum_finalize_kernel(const float* __restrict__ partials, int n, double scale,
                                                                   float* __restrict__ out) {
    __shared__ double lds[kSumThreads / 64];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += kSumThreads) a += (double)partials[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kSumThreads / 64; ++w) t += lds[w];
        out[0] = (float)(t * scale);
    }
}

// q = d(term)/d(ddx) at pixel (y,x): upstream * sign(ddx) * sum_c exp(-|dI_c|) / C   (x direction; y alike)
__device__ __forceinline__ void smooth_q(const float* __restrict__ D, const float* __restrict__ T, int C, int H, int W, int y,
                                         int x, float g, float& qx, float& qy) {
    const int HW = H * W;
    const float ddx = grad1d(D + y * W, x, W, 1), ddy = grad1d(D + x, y, H, W);
    float wx = 0.f, wy = 0.f;
    for (int c = 0; c < C; ++c) {
        const float* I = T + (size_t)c * HW;
        wx += 1.0f / expf(fabsf(grad1d(I + y * W, x, W, 1)));
        wy += 1.0f / expf(fabsf(grad1d(I + x, y, H, W)));
    }
    qx = g * wx * (ddx > 0.f ? 1.0f : (ddx < 0.f ? -1.0f : 0.0f));
    qy = g * wy * (ddy > 0.f ? 1.0f : (ddy < 0.f ? -1.0f : 0.0f));
}

// adjoint of torch.gradient along one line: contribution of q[j-1], q[j], q[j+1] to element j
__device__ __forceinline__ float grad1d_adjoint(float qm, float q0, float qp, int j, int L) {
    if (L == 1) return 0.f;
    float g = 0.f;
    if (j >= 1) g += (j - 1 == 0 ? 1.0f : 0.5f) * qm;
    if (j <= L - 2) g -= (j + 1 == L - 1 ? 1.0f : 0.5f) * qp;
    if (j == 0) g -= q0;
    if (j == L - 1) g += q0;
    return g;
}

__global__ __launch_bounds__(kThreads) void smooth_bwd_kernel(const float* __restrict__ depths,
                                                              const float* __restrict__ targets,
                                                              const float* __restrict__ g_value, int N, int S, int C, int H,
                                                              int W, int accumulate, float* __restrict__ g_depths) {
    const int n = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const float* D = depths + (size_t)n * HW;
    const float* T = targets + (size_t)(n / S) * C * HW;
    const float g = g_value[0] / ((float)N * (float)HW) / (float)C;
    float qx0, qy0, qxm = 0.f, qxp = 0.f, qym = 0.f, qyp = 0.f, dummy;
    smooth_q(D, T, C, H, W, y, x, g, qx0, qy0);
    if (x >= 1) smooth_q(D, T, C, H, W, y, x - 1, g, qxm, dummy);
    if (x <= W - 2) smooth_q(D, T, C, H, W, y, x + 1, g, qxp, dummy);
    if (y >= 1) smooth_q(D, T, C, H, W, y - 1, x, g, dummy, qym);
    if (y <= H - 2) smooth_q(D, T, C, H, W, y + 1, x, g, dummy, qyp);
    const float v = grad1d_adjoint(qxm, qx0, qxp, x, W) + grad1d_adjoint(qym, qy0, qyp, y, H);
    g_depths[(size_t)n * HW + p] = accumulate ? g_depths[(size_t)n * HW + p] + v : v;
}

// ---- tiled smoothness for C == 3 ------------------------------------------------------------------
// The generic kernels above evaluate 6 (forward) or 30 (backward) exp + divide pairs per pixel and map.  Here a wave
// owns 64 columns x kSmoothRows rows of ONE target and rolls down the rows with the previous / current / next row of the
// three target channels and of its S depth maps in registers (left / right neighbours are two more coalesced loads per
// row).  The edge weights Wx = sum_c exp(-|dI_c/dx|), Wy alike, are formed once per target pixel and shared by the S
// maps of that target (the reference stacks the target S times, train.py:80-83): a map costs two differences and two
// fma on top.  sum_c |dd|/exp(|dI_c|) = |dd| * sum_c exp(-|dI_c|) is a re-association: the value is a mean over all
// pixels with a 1e-4 relative bar, not a bit-exact map.  torch.gradient's one-sided edges are the same
// difference of clamped neighbours with scale 1 instead of 1/2.
constexpr int kSmoothRows = 8;      // forward: rows per wave
constexpr int kSmoothBwdRows = 16;  // backward: rows per wave (two extra rows of q on top); 8 or 4 for launches that
                                    // would leave the chip mostly empty (smooth_bwd_rows)
constexpr int kSmoothMaxS = 4;      // maps of one target per pass

struct SmoothGeom { int H, W, nbx, S, s0, sn, rows; };  // S maps per target; this launch handles maps s0 .. s0+sn-1 of
                                                           // each; rows: rows per wave (backward)

__device__ __forceinline__ float edge_scale(int i, int L) { return (i == 0 || i == L - 1) ? 1.0f : 0.5f; }
__device__ __forceinline__ float neg_exp(float a) { return __builtin_amdgcn_exp2f(a * -1.4426950408889634f); }
__device__ __forceinline__ float sgn(float v) { return v > 0.f ? 1.0f : (v < 0.f ? -1.0f : 0.0f); }

template <int S>
__global__ __launch_bounds__(kThreads) void smooth_tiled_fwd_kernel(const float* __restrict__ depths,
                                                                    const float* __restrict__ targets, SmoothGeom G,
                                                                    float* __restrict__ partials) {
    const int t = blockIdx.y, HW = G.H * G.W;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int bx = blockIdx.x % G.nbx, by = blockIdx.x / G.nbx;
    const int ya = (by * (kThreads / 64) + wave) * kSmoothRows, yb = min(ya + kSmoothRows, G.H);
    const int x = bx * 64 + lane, xc = min(x, G.W - 1), xl = max(xc - 1, 0), xr = min(xc + 1, G.W - 1);
    const float sx = edge_scale(xc, G.W);
    const float* I = targets + (size_t)t * 3 * HW;
    const float* D = depths + ((size_t)t * G.S + G.s0) * HW;
    float acc = 0.f;
    if (ya < G.H) {
        float ic[3][3], dc[S][3];  // [.][0..2] = rows r-1, r, r+1 (clamped) at column xc
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ic[c][1] = I[c * HW + max(ya - 1, 0) * G.W + xc];
            ic[c][2] = I[c * HW + ya * G.W + xc];
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            dc[s][1] = D[s * HW + max(ya - 1, 0) * G.W + xc];
            dc[s][2] = D[s * HW + ya * G.W + xc];
        }
        for (int r = ya; r < yb; ++r) {
            const int rn = min(r + 1, G.H - 1) * G.W, ro = r * G.W;
            const float sy = edge_scale(r, G.H);
            float wx = 0.f, wy = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                ic[c][0] = ic[c][1];
                ic[c][1] = ic[c][2];
                ic[c][2] = I[c * HW + rn + xc];
                wx += neg_exp(fabsf(I[c * HW + ro + xr] - I[c * HW + ro + xl]) * sx);
                wy += neg_exp(fabsf(ic[c][2] - ic[c][0]) * sy);
            }
#pragma unroll
            for (int s = 0; s < S; ++s) {
                dc[s][0] = dc[s][1];
                dc[s][1] = dc[s][2];
                dc[s][2] = D[s * HW + rn + xc];
                const float adx = fabsf(D[s * HW + ro + xr] - D[s * HW + ro + xl]) * sx;
                const float ady = fabsf(dc[s][2] - dc[s][0]) * sy;
                acc += fmaf(adx, wx, ady * wy);
            }
        }
        if (x >= G.W) acc = 0.f;
    }
    acc = wave_sum(acc);
    if (lane == 0) partials[((size_t)t * gridDim.x + blockIdx.x) * (kThreads / 64) + wave] = acc;
}

// Backward: d value / d depth.  With q = g * W * sign(dd) * scale at every pixel (g = upstream / (N*H*W*3)), the adjoint
// of torch.gradient along a line is q[j-1] - q[j+1], where a neighbour beyond the end of the line is replaced by -q[j].
// A wave covers 62 columns plus one halo lane on either side (the x neighbours of q come from a lane shuffle) and rolls
// over rows ya-1 .. yb: q of row r is formed when row r+1 is in registers, the output row r-1 is written then.
template <int S>
__global__ __launch_bounds__(kThreads) void smooth_tiled_bwd_kernel(const float* __restrict__ depths,
                                                                    const float* __restrict__ targets,
                                                                    const float* __restrict__ g_value, float inv_count,
                                                                    SmoothGeom G, int accumulate,
                                                                    float* __restrict__ g_depths) {
    const int t = blockIdx.y, HW = G.H * G.W;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int bx = blockIdx.x % G.nbx, by = blockIdx.x / G.nbx;
    const int ya = (by * (kThreads / 64) + wave) * G.rows, yb = min(ya + G.rows, G.H);
    if (ya >= G.H) return;
    const int x = bx * 62 - 1 + lane, xc = min(max(x, 0), G.W - 1), xl = max(xc - 1, 0), xr = min(xc + 1, G.W - 1);
    const bool store = lane >= 1 && lane <= 62 && x < G.W;
    const float sx = edge_scale(xc, G.W);
    const float g = g_value[0] * inv_count;
    const float* I = targets + (size_t)t * 3 * HW;
    const float* D = depths + ((size_t)t * G.S + G.s0) * HW;
    float* O = g_depths + ((size_t)t * G.S + G.s0) * HW;
    auto row_of = [&](int r) { return min(max(r, 0), G.H - 1) * G.W; };
    float ic[3][3], dc[S][3], qy[S][2], gx[S];  // qy[.][0], [1] = q_y of rows r-2, r-1; gx = x part of output row r-1
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ic[c][1] = I[c * HW + row_of(ya - 2) + xc];
        ic[c][2] = I[c * HW + row_of(ya - 1) + xc];
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        dc[s][1] = D[s * HW + row_of(ya - 2) + xc];
        dc[s][2] = D[s * HW + row_of(ya - 1) + xc];
        qy[s][0] = qy[s][1] = 0.f;
        gx[s] = 0.f;
    }
    for (int r = ya - 1; r <= yb; ++r) {
        const bool live = r >= 0 && r < G.H;  // wave-uniform
        const int rn = row_of(r + 1), ro = row_of(r);
        const float sy = edge_scale(r, G.H);
        float wx = 0.f, wy = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ic[c][0] = ic[c][1];
            ic[c][1] = ic[c][2];
            ic[c][2] = I[c * HW + rn + xc];
            wx += neg_exp(fabsf(I[c * HW + ro + xr] - I[c * HW + ro + xl]) * sx);
            wy += neg_exp(fabsf(ic[c][2] - ic[c][0]) * sy);
        }
        const int y = r - 1;  // the output row of this iteration
        const bool emit = y >= ya && y < yb;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            dc[s][0] = dc[s][1];
            dc[s][1] = dc[s][2];
            dc[s][2] = D[s * HW + rn + xc];
            float qx = g * wx * sgn(D[s * HW + ro + xr] - D[s * HW + ro + xl]) * sx;
            float qyn = g * wy * sgn(dc[s][2] - dc[s][0]) * sy;
            if (!live) qx = qyn = 0.f;
            const float ql = __shfl_up(qx, 1, 64), qr = __shfl_down(qx, 1, 64);
            const float gxn = (xc >= 1 ? ql : -qx) - (xc <= G.W - 2 ? qr : -qx);
            if (emit && store) {
                const float up = y >= 1 ? qy[s][0] : -qy[s][1];
                const float down = y <= G.H - 2 ? qyn : -qy[s][1];
                float* o = O + s * HW + y * G.W + x;
                const float v = gx[s] + (up - down);
                *o = accumulate ? *o + v : v;
            }
            qy[s][0] = qy[s][1];
            qy[s][1] = qyn;
            gx[s] = gxn;
        }
    }
}

// Forward value and gradient of a10 in one pass (the step that wants both at once, ds_train_step_fwdbwd): the backward
// kernel above with the gradient's upstream factor g = scale / count from the host, accumulating into g_depths, plus the
// forward's sum over the pixels a wave stores (one partial per wave, fixed order in the finalize).
template <int S>
__global__ __launch_bounds__(kThreads) void smooth_tiled_fb_kernel(const float* __restrict__ depths,
                                                                    const float* __restrict__ targets,
                                                                   float g, SmoothGeom G, float* __restrict__ g_depths,
                                                                   float* __restrict__ partials) {
    const int t = blockIdx.y, HW = G.H * G.W;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int bx = blockIdx.x % G.nbx, by = blockIdx.x / G.nbx;
    const int ya = (by * (kThreads / 64) + wave) * G.rows, yb = min(ya + G.rows, G.H);
    float* part = partials + ((size_t)t * gridDim.x + blockIdx.x) * (kThreads / 64) + wave;
    if (ya >= G.H) {
        if (lane == 0) *part = 0.f;
        return;
    }
    float acc = 0.f;   // this wave's part of sum |dD| * w (the forward value; rows ya .. yb-1, columns it stores)
    const int x = bx * 62 - 1 + lane, xc = min(max(x, 0), G.W - 1), xl = max(xc - 1, 0), xr = min(xc + 1, G.W - 1);
    const bool store = lane >= 1 && lane <= 62 && x < G.W;
    const float sx = edge_scale(xc, G.W);
    const float* I = targets + (size_t)t * 3 * HW;
    const float* D = depths + ((size_t)t * G.S + G.s0) * HW;
    float* O = g_depths + ((size_t)t * G.S + G.s0) * HW;
    auto row_of = [&](int r) { return min(max(r, 0), G.H - 1) * G.W; };
    float ic[3][3], dc[S][3], qy[S][2], gx[S];  // qy[.][0], [1] = q_y of rows r-2, r-1; gx = x part of output row r-1
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        ic[c][1] = I[c * HW + row_of(ya - 2) + xc];
        ic[c][2] = I[c * HW + row_of(ya - 1) + xc];
    }
#pragma unroll
    for (int s = 0; s < S; ++s) {
        dc[s][1] = D[s * HW + row_of(ya - 2) + xc];
        dc[s][2] = D[s * HW + row_of(ya - 1) + xc];
        qy[s][0] = qy[s][1] = 0.f;
        gx[s] = 0.f;
    }
    for (int r = ya - 1; r <= yb; ++r) {
        const bool live = r >= 0 && r < G.H;  // wave-uniform
        const int rn = row_of(r + 1), ro = row_of(r);
        const float sy = edge_scale(r, G.H);
        float wx = 0.f, wy = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            ic[c][0] = ic[c][1];
            ic[c][1] = ic[c][2];
            ic[c][2] = I[c * HW + rn + xc];
            wx += neg_exp(fabsf(I[c * HW + ro + xr] - I[c * HW + ro + xl]) * sx);
            wy += neg_exp(fabsf(ic[c][2] - ic[c][0]) * sy);
        }
        const int y = r - 1;  // the output row of this iteration
        const bool emit = y >= ya && y < yb;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            dc[s][0] = dc[s][1];
            dc[s][1] = dc[s][2];
            dc[s][2] = D[s * HW + rn + xc];
            const float ddx = D[s * HW + ro + xr] - D[s * HW + ro + xl], ddy = dc[s][2] - dc[s][0];
            float qx = g * wx * sgn(ddx) * sx;
            float qyn = g * wy * sgn(ddy) * sy;
            if (!live) qx = qyn = 0.f;
            if (store && r >= ya && r < yb) acc += fmaf(fabsf(ddx) * sx, wx, (fabsf(ddy) * sy) * wy);
            const float ql = __shfl_up(qx, 1, 64), qr = __shfl_down(qx, 1, 64);
            const float gxn = (xc >= 1 ? ql : -qx) - (xc <= G.W - 2 ? qr : -qx);
            if (emit && store) {
                const float up = y >= 1 ? qy[s][0] : -qy[s][1];
                const float down = y <= G.H - 2 ? qyn : -qy[s][1];
                float* o = O + s * HW + y * G.W + x;
                const float v = gx[s] + (up - down);
                *o += v;
            }
            qy[s][0] = qy[s][1];
            qy[s][1] = qyn;
            gx[s] = gxn;
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) *part = acc;
}

inline int smooth_fwd_blocks(int H, int W) {
    const int per = (kThreads / 64) * kSmoothRows;
    return ((W + 63) / 64) * ((H + per - 1) / per);
}
inline int smooth_bwd_blocks(int H, int W, int rows) {
    const int per = (kThreads / 64) * rows;
    return ((W + 61) / 62) * ((H + per - 1) / per);
}
// A wave walks its rows one after the other, every step waiting for its loads: with 16 rows per wave 12 targets of
// 192x640 are 396 workgroups -- 27 us of latency on a mostly empty chip (the forward + backward of the whole photometric
// loss take 100).  Shorter walks until the launch has about eight workgroups per CU; the outputs do not depend on it.
inline int smooth_bwd_rows(int B, int H, int W) {
    const long long want = 8ll * ds::cu_count();
    int rows = kSmoothBwdRows;
    while (rows > 4 && (long long)B * smooth_bwd_blocks(H, W, rows) < want) rows >>= 1;
    return rows;
}

// ---------------------------------------------------------------------------------------------------
// a11-a15: all depth metrics in one pass
// ---------------------------------------------------------------------------------------------------
constexpr int kMetPix = 8;  // pixels per thread: two 16-byte loads of each map

// One valid-or-not pixel into the eight running sums.  The four quotients (|d|/g, d^2/g, p/g, g/p) are IEEE divides; for
// operands of ordinary magnitude they come bit-identically from one refined reciprocal per denominator (div_by,
// ds_common.h: 5 instructions each instead of ~12), anything else takes the plain divide.
__device__ __forceinline__ void metrics_pixel(float p, float g, float eps, float (&a)[8]) {
    if (!(g > 0.0f)) return;
    const float df = p - g, sq = df * df, ad = fabsf(df), ap = fabsf(p);
    const float lg = logf(p + eps) - logf(g + eps);   // metrics.py:122
    float q_ad, q_sq, q_pg, q_gp;
    if (g > 1e-15f && g < 1e15f && ap > 1e-15f && ap < 1e15f) {
        const float rg = rcp_refined(g), rp = rcp_refined(p);
        q_ad = div_by(ad, g, rg);
        q_sq = div_by(sq, g, rg);
        q_pg = div_by(p, g, rg);
        q_gp = div_by(g, p, rp);
    } else {
        q_ad = ad / g;
        q_sq = sq / g;
        q_pg = p / g;
        q_gp = g / p;
    }
    a[0] += sq;
    a[1] += lg * lg;
    a[2] += q_ad;
    a[3] += q_sq;
    const float r = fmaxf(q_pg, q_gp);
    a[4] += r < 1.25f ? 1.f : 0.f;
    a[5] += r < 1.5625f ? 1.f : 0.f;
    a[6] += r < 1.953125f ? 1.f : 0.f;
    a[7] += 1.f;
}

__global__ __launch_bounds__(kThreads) void metrics_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                           int HW, int vec, float eps, float* __restrict__ partials) {
    __shared__ float lds[8][kThreads / 64];
    const int b = blockIdx.y;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const float* P = pred + (size_t)b * HW;
    const float* G = gt + (size_t)b * HW;
    const int base = blockIdx.x * (kMetPix * kThreads);
    if (vec && base + kMetPix * kThreads <= HW) {
        // whole block inside the image; vec: every image starts 16-byte aligned (HW % 4 == 0 and aligned bases)
#pragma unroll
        for (int it = 0; it < kMetPix / 4; ++it) {
            const int i = base + (it * kThreads + threadIdx.x) * 4;
            const float4 p = *reinterpret_cast<const float4*>(P + i), g = *reinterpret_cast<const float4*>(G + i);
            metrics_pixel(p.x, g.x, eps, a);
            metrics_pixel(p.y, g.y, eps, a);
            metrics_pixel(p.z, g.z, eps, a);
            metrics_pixel(p.w, g.w, eps, a);
        }
    } else {
#pragma unroll
        for (int it = 0; it < kMetPix; ++it) {
            const int i = base + it * kThreads + threadIdx.x;
            if (i < HW) metrics_pixel(P[i], G[i], eps, a);
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        a[k] = wave_sum(a[k]);
        if (lane == 0) lds[k][wave] = a[k];
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        float s = 0.f;
        for (int w = 0; w < kThreads / 64; ++w) s += lds[threadIdx.x][w];
        partials[((size_t)b * gridDim.x + blockIdx.x) * 8 + threadIdx.x] = s;
    }
}

// partials (B, nblk, 8) -> out (8): per-image finalisation then mean over B (metrics.py:111,125,139,153)
// Two short kernels, fixed summation order (deterministic): one workgroup per image sums the image's block partials in
// double (every thread one or a few independent 32-byte loads) and forms the image's terms; one wave sums over the images.
__global__ __launch_bounds__(kThreads) void metrics_image_kernel(const float* __restrict__ partials, int nblk, double inv_hw,
                                                                 double* __restrict__ per_image,
                                                                 long long* __restrict__ counts) {
    __shared__ double lds[kThreads / 64][8];
    const int b = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 2
    for (int j = threadIdx.x; j < nblk; j += kThreads) {
        const float4* q = reinterpret_cast<const float4*>(partials + ((size_t)b * nblk + j) * 8);
        const float4 lo = q[0], hi = q[1];
        s[0] += (double)lo.x; s[1] += (double)lo.y; s[2] += (double)lo.z; s[3] += (double)lo.w;
        s[4] += (double)hi.x; s[5] += (double)hi.y; s[6] += (double)hi.z; s[7] += (double)hi.w;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o, 64);
        if (lane == 0) lds[wave][k] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const int k = threadIdx.x;
        double t = 0.0;
        for (int w = 0; w < kThreads / 64; ++w) t += lds[w][k];
        per_image[(size_t)b * 8 + k] = k < 2 ? sqrt(t * inv_hw) : (k < 7 ? t * inv_hw : t);
        // the three threshold counts and the valid count are sums of 0/1: every fp32 block partial (<= 2048) and the
        // double sum over them is an exact integer
        if (counts && k >= 4) counts[(size_t)b * 4 + (k - 4)] = (long long)t;
    }
}

__global__ __launch_bounds__(64) void metrics_finalize_kernel(const double* __restrict__ per_image, int B,
                                                              float* __restrict__ out) {
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = threadIdx.x; b < B; b += 64) {
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += per_image[(size_t)b * 8 + k];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc[k] += __shfl_xor(acc[k], o, 64);
    }
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) out[k] = (float)(acc[k] / (double)B);
        out[7] = (float)acc[7];
    }
}

inline bool bad_img(int N, int C, int H, int W) {
    return N <= 0 || C <= 0 || H <= 0 || W <= 0 || N > 65535 || (long long)H * W > (1ll << 30);
}

inline int metric_blocks(int HW) { return (HW + kThreads * kMetPix - 1) / (kThreads * kMetPix); }

inline bool fill_planes(Planes& pl, const double* depths_host, int D) {
    if (!depths_host || D <= 0 || D > kMaxPlanes) return false;
    for (int k = 0; k < D; ++k) pl.d[k] = (float)depths_host[k];
    for (int k = D; k < kMaxPlanes; ++k) pl.d[k] = 0.f;
    return true;
}

// ---------------------------------------------------------------------------------------------------
// Input pipeline to the device: what torchvision's to_tensor does to the resized PIL frame
// (dataset.py:51,110): uint8 HWC -> float32 CHW, value / 255 (true divide).  Four pixels per thread: 12 interleaved
// bytes in, one float4 per channel out.
// ---------------------------------------------------------------------------------------------------
// ground-truth depth (dataset.py:116-120): 16-bit PNG samples -> float32, divided by 2**16 (exact: a power of two)
__global__ __launch_bounds__(kThreads) void ingest_depth_u16_kernel(const uint16_t* __restrict__ in, size_t n,
                                                                    float* __restrict__ out) {
    const size_t i = ((size_t)blockIdx.x * kThreads + threadIdx.x) * 2;
    if (i + 1 < n) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(in + i);
        float2 f;
        f.x = (float)(w & 0xffffu) * (1.0f / 65536.0f);
        f.y = (float)(w >> 16) * (1.0f / 65536.0f);
        *reinterpret_cast<float2*>(out + i) = f;
    } else if (i < n) {
        out[i] = (float)in[i] * (1.0f / 65536.0f);
    }
}

__global__ __launch_bounds__(kThreads) void ingest_u8_kernel(const uint8_t* __restrict__ frames, int HW,
                                                             float* __restrict__ out) {
    const int n = blockIdx.y;
    const int q = blockIdx.x * kThreads + threadIdx.x;  // group of four pixels
    const int p0 = q * 4;
    if (p0 >= HW) return;
    const uint8_t* in = frames + (size_t)n * HW * 3 + (size_t)p0 * 3;
    float* o = out + (size_t)n * 3 * HW + p0;
    if (p0 + 4 <= HW && (HW & 3) == 0) {
        const uint32_t* w = reinterpret_cast<const uint32_t*>(in);  // 12-byte groups of a 4-byte aligned base
        const uint32_t a = w[0], b = w[1], c = w[2];
        const uint8_t v[12] = {(uint8_t)a, (uint8_t)(a >> 8), (uint8_t)(a >> 16), (uint8_t)(a >> 24),
                               (uint8_t)b, (uint8_t)(b >> 8), (uint8_t)(b >> 16), (uint8_t)(b >> 24),
                               (uint8_t)c, (uint8_t)(c >> 8), (uint8_t)(c >> 16), (uint8_t)(c >> 24)};
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            float4 f;
            f.x = (float)v[ch] / 255.0f;
            f.y = (float)v[3 + ch] / 255.0f;
            f.z = (float)v[6 + ch] / 255.0f;
            f.w = (float)v[9 + ch] / 255.0f;
            *reinterpret_cast<float4*>(o + (size_t)ch * HW) = f;
        }
    } else {
        for (int i = 0; i < 4 && p0 + i < HW; ++i)
            for (int ch = 0; ch < 3; ++ch) o[(size_t)ch * HW + i] = (float)in[3 * i + ch] / 255.0f;
    }
}

// ---------------------------------------------------------------------------------------------------
// The resize of the loaders on the device (dataset.py:51,110: tf.resize(img, (192, 640)) on a PIL image = PIL's
// Image.resize(BILINEAR)).  The arithmetic is Pillow's (libImaging/Resample.c): a separable triangle filter whose support
// grows with the down-scale factor, coefficients normalised in double and rounded to 22-bit fixed point, each pass
// accumulated in int32 from 1 << 21 and clipped to 0..255, horizontal pass first with its result rounded to uint8.
// Integer work, so bit-exact: the coefficients are formed on the device in IEEE double with Pillow's operation order
// (this file is compiled without contraction), one thread per output column / row.
// ---------------------------------------------------------------------------------------------------
constexpr int kPilBits = 32 - 8 - 2;
constexpr int kPilMaxTaps = 64;  // down-scaling by up to ~31x

struct PilAxis { int in_size, out_size, ksize; double scale, filterscale, support, ss; };

inline bool pil_axis(PilAxis& a, int in_size, int out_size) {
    a.in_size = in_size; a.out_size = out_size;
    a.scale = (double)in_size / out_size;
    a.filterscale = a.scale < 1.0 ? 1.0 : a.scale;
    a.support = 1.0 * a.filterscale;
    a.ss = 1.0 / a.filterscale;
    a.ksize = (int)ceil(a.support) * 2 + 1;
    return a.ksize <= kPilMaxTaps;
}

// fixed-point coefficients of output sample xx: first tap and tap count; kk[i * stride] receives tap i
__device__ __forceinline__ int pil_coeffs(const PilAxis& a, int xx, int& xmin_out, int* kk, int stride) {
    const double center = (xx + 0.5) * a.scale;
    int xmin = (int)(center - a.support + 0.5), xmax = (int)(center + a.support + 0.5);
    if (xmin < 0) xmin = 0;
    if (xmax > a.in_size) xmax = a.in_size;
    xmax -= xmin;
    if (xmax > a.ksize) xmax = a.ksize;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) {
        double t = (x + xmin - center + 0.5) * a.ss;
        if (t < 0.0) t = -t;
        ww += t < 1.0 ? 1.0 - t : 0.0;
    }
    for (int x = 0; x < xmax; ++x) {
        double t = (x + xmin - center + 0.5) * a.ss;
        if (t < 0.0) t = -t;
        const double w = t < 1.0 ? 1.0 - t : 0.0;
        const double k = ww != 0.0 ? w / ww : w;
        kk[x * stride] = k < 0 ? (int)(-0.5 + k * (double)(1 << kPilBits)) : (int)(0.5 + k * (double)(1 << kPilBits));
    }
    xmin_out = xmin;
    return xmax;
}

__device__ __forceinline__ int pil_clip8(int v) {
    v >>= kPilBits;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: in (N,Hin,Win,3) -> tmp (N,Hin,Wout,3).  A block owns 64 output columns (their coefficients in LDS,
// computed once) and walks down the rows four at a time.
__global__ __launch_bounds__(kThreads) void pil_resize_h_kernel(const uint8_t* __restrict__ in, PilAxis A, int Hin,
                                                                int rows_per_block, uint8_t* __restrict__ tmp) {
    __shared__ int kk[kPilMaxTaps][64];
    __shared__ int first[64], count[64];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int xx = blockIdx.x * 64 + tx, n = blockIdx.z;
    if (ty == 0) {
        int xmin = 0, cnt = 0;
        if (xx < A.out_size) cnt = pil_coeffs(A, xx, xmin, &kk[0][tx], 64);
        first[tx] = xmin;
        count[tx] = cnt;
    }
    __syncthreads();
    if (xx >= A.out_size) return;
    const int xmin = first[tx], cnt = count[tx];
    const int y_end = min(Hin, (int)(blockIdx.y + 1) * rows_per_block);
    for (int y = blockIdx.y * rows_per_block + ty; y < y_end; y += 4) {
        const uint8_t* row = in + ((size_t)n * Hin + y) * A.in_size * 3 + (size_t)xmin * 3;
        int a0 = 1 << (kPilBits - 1), a1 = a0, a2 = a0;
        for (int x = 0; x < cnt; ++x) {
            const int k = kk[x][tx];
            a0 += (int)row[3 * x + 0] * k;
            a1 += (int)row[3 * x + 1] * k;
            a2 += (int)row[3 * x + 2] * k;
        }
        uint8_t* o = tmp + (((size_t)n * Hin + y) * A.out_size + xx) * 3;
        o[0] = (uint8_t)pil_clip8(a0);
        o[1] = (uint8_t)pil_clip8(a1);
        o[2] = (uint8_t)pil_clip8(a2);
    }
}

// vertical pass: tmp (N,Hin,Wout,3) -> out_u8 (N,Hout,Wout,3) and / or out_f32 (N,3,Hout,Wout) = value / 255 (to_tensor).
// One output row per blockIdx.y (its coefficients are the same for every thread), one interleaved byte per thread.
__global__ __launch_bounds__(kThreads) void pil_resize_v_kernel(const uint8_t* __restrict__ tmp, PilAxis A, int Wout,
                                                                uint8_t* __restrict__ out_u8, float* __restrict__ out_f32) {
    __shared__ int kk[kPilMaxTaps];
    __shared__ int first, count;
    const int yy = blockIdx.y, n = blockIdx.z;
    if (threadIdx.x == 0) {
        int ymin = 0;
        count = pil_coeffs(A, yy, ymin, kk, 1);
        first = ymin;
    }
    __syncthreads();
    const int xc = blockIdx.x * kThreads + threadIdx.x;  // interleaved (x, channel) index
    const int rowlen = Wout * 3;
    if (xc >= rowlen) return;
    const uint8_t* col = tmp + ((size_t)n * A.in_size + first) * rowlen + xc;
    int acc = 1 << (kPilBits - 1);
    for (int y = 0; y < count; ++y) acc += (int)col[(size_t)y * rowlen] * kk[y];
    const int v = pil_clip8(acc);
    if (out_u8) out_u8[((size_t)n * A.out_size + yy) * rowlen + xc] = (uint8_t)v;
    if (out_f32) {
        const int x = xc / 3, c = xc - 3 * x;
        out_f32[(((size_t)n * 3 + c) * A.out_size + yy) * Wout + x] = (float)v / 255.0f;
    }
}

// ---------------------------------------------------------------------------------------------------
// the two resizes around the plane sweep (train.py:55 bilinear, train.py:60 bicubic; align_corners=False, size given,
// so the scale is in/out in fp32 as ATen's area_pixel_compute_scale forms it).  One thread per output pixel.
// ---------------------------------------------------------------------------------------------------
struct ResizeGeom { int Hin, Win, Hout, Wout; float sy, sx; };

__global__ __launch_bounds__(kThreads) void resize_bilinear_kernel(const float* __restrict__ in, ResizeGeom G,
                                                                   float* __restrict__ out) {
    const int p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= G.Hout * G.Wout) return;
    const int oy = p / G.Wout, ox = p - oy * G.Wout;
    // source index scale*(dst+0.5)-0.5, negative values clamp to 0; upper neighbour clamps to the last row/column
    const float ry = fmaxf(fmaf(G.sy, (float)oy + 0.5f, -0.5f), 0.f), rx = fmaxf(fmaf(G.sx, (float)ox + 0.5f, -0.5f), 0.f);
    const int y0 = min((int)ry, G.Hin - 1), x0 = min((int)rx, G.Win - 1);
    const int y1 = min(y0 + 1, G.Hin - 1), x1 = min(x0 + 1, G.Win - 1);
    const float ly = fminf(fmaxf(ry - (float)y0, 0.f), 1.f), lx = fminf(fmaxf(rx - (float)x0, 0.f), 1.f);
    const float* pl = in + (size_t)blockIdx.y * G.Hin * G.Win;
    const float top = fmaf(1.f - lx, pl[y0 * G.Win + x0], lx * pl[y0 * G.Win + x1]);
    const float bot = fmaf(1.f - lx, pl[y1 * G.Win + x0], lx * pl[y1 * G.Win + x1]);
    out[(size_t)blockIdx.y * G.Hout * G.Wout + p] = fmaf(1.f - ly, top, ly * bot);
}

// Keys cubic convolution, A = -0.75 (ATen get_cubic_upsample_coefficients)
__device__ __forceinline__ void cubic_coefs(float t, float c[4]) {
    const float A = -0.75f;
    const float x0 = t + 1.f, x3 = 2.f - t, x2 = 1.f - t;
    c[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    c[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    c[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    c[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

__global__ __launch_bounds__(kThreads) void resize_bicubic_kernel(const float* __restrict__ in, ResizeGeom G,
                                                                  float* __restrict__ out) {
    const int p = blockIdx.x * kThreads + threadIdx.x;
    if (p >= G.Hout * G.Wout) return;
    const int oy = p / G.Wout, ox = p - oy * G.Wout;
    const float ry = fmaf(G.sy, (float)oy + 0.5f, -0.5f), rx = fmaf(G.sx, (float)ox + 0.5f, -0.5f);   // not clamped for cubic
    const float fy = floorf(ry), fx = floorf(rx);
    const int iy = (int)fy, ix = (int)fx;
    float cy[4], cx[4];
    cubic_coefs(fminf(fmaxf(ry - fy, 0.f), 1.f), cy);
    cubic_coefs(fminf(fmaxf(rx - fx, 0.f), 1.f), cx);
    int xs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xs[i] = min(max(ix - 1 + i, 0), G.Win - 1);
    const float* pl = in + (size_t)blockIdx.y * G.Hin * G.Win;
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float* row = pl + (size_t)min(max(iy - 1 + j, 0), G.Hin - 1) * G.Win;
        float r = cx[0] * row[xs[0]];
#pragma unroll
        for (int i = 1; i < 4; ++i) r += cx[i] * row[xs[i]];
        acc = (j == 0) ? cy[0] * r : acc + cy[j] * r;
    }
    out[(size_t)blockIdx.y * G.Hout * G.Wout + p] = acc;
}

inline bool resize_geom(ResizeGeom& G, int N, int Hin, int Win, int Hout, int Wout) {
    if (N <= 0 || N > 65535 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || (long long)Hin * Win > (1ll << 30) ||
        (long long)Hout * Wout > (1ll << 30))
        return false;
    G = ResizeGeom{Hin, Win, Hout, Wout, (float)Hin / (float)Hout, (float)Win / (float)Wout};
    return true;
}

// ---------------------------------------------------------------------------------------------------
// ds_depth_metrics_noalloc: the workspace of ds_depth_metrics in a file-scope device array instead of a caller buffer.
// `scratch` is a host-side stand-in for the symbol, not a device address: host code only ever uses the per-device pointer
// hipGetSymbolAddress returns (every GPU the library runs on holds its own copy of the array at its own address).
// ---------------------------------------------------------------------------------------------------
constexpr size_t kScratchBytes = DS_NOALLOC_SCRATCH_BYTES;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];

constexpr int kMaxDevices = 64;
static void* scratch_ptr_of_device[kMaxDevices];  // zero-initialised; filled once per device
static std::mutex scratch_mutex;

// device address of `scratch` on the current device (looked up on the first call per device); rc = DS_OK, DS_ERR_*, or hipError_t
static int scratch_device_ptr(void** out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= kMaxDevices) return DS_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(scratch_mutex);
    if (!scratch_ptr_of_device[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        if (!p) return DS_ERR_WORKSPACE;
        scratch_ptr_of_device[dev] = p;
    }
    *out = scratch_ptr_of_device[dev];
    return DS_OK;
}

// the launches of ds_depth_metrics; `workspace` is a device pointer of at least ds_metrics_workspace_bytes(B,H,W) bytes,
// 16-byte aligned.  Every partial and per-image slot that is read has been written by this call: nothing needs clearing.
static int depth_metrics_launch(const float* pred, const float* gt, int B, int H, int W, float eps, float* out,
                                long long* counts, void* workspace, hipStream_t st) {
    const int nb = metric_blocks(H * W);
    const int vec = ((H * W) & 3) == 0 && (((uintptr_t)pred | (uintptr_t)gt) & 15) == 0;
    float* partials = (float*)workspace;
    double* per_image = (double*)(partials + (size_t)B * nb * 8);  // 32 * B * nb bytes in: 8-byte aligned
    hipLaunchKernelGGL(metrics_kernel, dim3(nb, B), dim3(kThreads), 0, st, pred, gt, H * W, vec, eps, partials);
    DS_CHECK_LAUNCH();
    hipLaunchKernelGGL(metrics_image_kernel, dim3(B), dim3(kThreads), 0, st, partials, nb, 1.0 / ((double)H * W), per_image,
                       counts);
    DS_CHECK_LAUNCH();
    hipLaunchKernelGGL(metrics_finalize_kernel, dim3(1), dim3(64), 0, st, per_image, B, out);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

}  // namespace

extern "C" {

int ds_version(void) { return 101; }

const char* ds_error_string(int code) {
    switch (code) {
        case DS_OK: return "ok";
        case DS_ERR_INVALID_ARG: return "depthsynth: invalid argument (null pointer or dimension out of range)";
        case DS_ERR_UNSUPPORTED: return "depthsynth: unsupported configuration (e.g. more than DS_MAX_SOURCES sources)";
        case DS_ERR_WORKSPACE: return "depthsynth: workspace missing or too small";
        case DS_ERR_SCRATCH_LIMIT: return "depthsynth: input needs more than the DS_NOALLOC_SCRATCH_BYTES built-in scratch; "
                                          "use the entry point that takes a workspace";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "depthsynth: unknown error";
    }
}

int ds_synthesize_from_depths(const float* K, const float* Kinv, const float* pose, const double* depths_host, int D,
                              const float* src, int C, int Hs, int Ws, int H, int W, float* out, void* stream) {
    Planes pl;
    if (!K || !Kinv || !pose || !src || !out || bad_img(D, C, H, W) || Hs < 2 || Ws < 2 || !fill_planes(pl, depths_host, D))
        return DS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(plane_warp_kernel, dim3(pix_blocks(H * W), D), dim3(kThreads), 0, (hipStream_t)stream, K, Kinv, pose, pl,
                       src, C, Hs, Ws, H, W, out);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

int ds_cost_volume(const float* target, const float* synth, int D, int C, int H, int W, float* cost, void* stream) {
    if (!target || !synth || !cost || bad_img(D, C, H, W)) return DS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(cost_volume_kernel, dim3(pix_blocks(H * W), D), dim3(kThreads), 0, (hipStream_t)stream, target, synth, C,
                       H * W, cost);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

int ds_depth_from_costs(const float* cost, int N, int D, int H, int W, int sid, float* depth, void* stream) {
    if (!cost || !depth || bad_img(N, D, H, W)) return DS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(depth_from_costs_kernel, dim3(pix_blocks(H * W), N), dim3(kThreads), 0, (hipStream_t)stream, cost, D,
                       H * W, sid, depth);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

static int plane_sweep_launch(const float* K, const float* Kinv, const float* pose, const double* depths_host, int D,
                              const float* src, const float* targets, int B, int S, int C, int Hs, int Ws, int H, int W,
                              int sid, float* depth, void* stream) {
    Planes pl;
    if (!K || !Kinv || !pose || !src || !targets || !depth || B <= 0 || S <= 0 || (long long)B * S > 65535 ||
        bad_img(B * S, C, H, W) || Hs < 2 || Ws < 2 || (long long)Hs * Ws > (1ll << 30) ||
        (long long)C * H * W * 4 >= (1ll << 32) || (long long)C * Hs * Ws * 4 >= (1ll << 32) ||  // 32-bit byte offsets
        !fill_planes(pl, depths_host, D))
        return DS_ERR_INVALID_ARG;
    const WarpGeom G = warp_geom(Hs, Ws, H, W);
    dim3 grid(G.nbx * ((H + kThreads / 64 - 1) / (kThreads / 64)), B * S), blk(kThreads);
    if (C == 3)
        hipLaunchKernelGGL(plane_sweep_depth_kernel<3>, grid, blk, 0, (hipStream_t)stream, K, Kinv, pose, pl, D, src,
                           targets, S, C, G, sid, depth);
    else
        hipLaunchKernelGGL(plane_sweep_depth_kernel<0>, grid, blk, 0, (hipStream_t)stream, K, Kinv, pose, pl, D, src,
                           targets, S, C, G, sid, depth);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

int ds_plane_sweep_depth(const float* K, const float* Kinv, const float* pose, const double* depths_host, int D,
                         const float* src, const float* target_small, int NS, int C, int Hs, int Ws, int H, int W,
                         int sid, float* depth, void* stream) {
    return plane_sweep_launch(K, Kinv, pose, depths_host, D, src, target_small, 1, NS, C, Hs, Ws, H, W, sid, depth, stream);
}

int ds_plane_sweep_depth_batch(const float* K, const float* Kinv, const float* pose, const double* depths_host, int D,
                               const float* src, const float* targets_small, int B, int S, int C, int Hs, int Ws, int H,
                               int W, int sid, float* depth, void* stream) {
    return plane_sweep_launch(K, Kinv, pose, depths_host, D, src, targets_small, B, S, C, Hs, Ws, H, W, sid, depth, stream);
}

size_t ds_smooth_workspace_bytes(int N, int H, int W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    // one float per workgroup (generic kernel) or per wave (tiled kernel), whichever is more, for every map
    const size_t per_map = (size_t)std::max(pix_blocks(H * W), smooth_fwd_blocks(H, W) * (kThreads / 64));
    return (size_t)N * per_map * sizeof(float);
}

int ds_smooth_batch_fwd(const float* depths, const float* targets, int B, int S, int C, int H, int W, float* value,
                        void* workspace, size_t workspace_bytes, void* stream) {
    if (!depths || !targets || !value || B <= 0 || S <= 0 || (long long)B * S > 65535 || bad_img(B * S, C, H, W))
        return DS_ERR_INVALID_ARG;
    const int N = B * S;
    if (!workspace || workspace_bytes < ds_smooth_workspace_bytes(N, H, W)) return DS_ERR_WORKSPACE;
    float* partials = (float*)workspace;
    hipStream_t st = (hipStream_t)stream;
    if (C != 3) {
        const int nb = pix_blocks(H * W);
        hipLaunchKernelGGL(smooth_fwd_kernel, dim3(nb, N), dim3(kThreads), 0, st, depths, targets, S, C, H, W, partials);
        DS_CHECK_LAUNCH();
        hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(kSumThreads), 0, st, partials, nb * N, 1.0 / ((double)N * H * W),
                           value);
        DS_CHECK_LAUNCH();
        return DS_OK;
    }
    const int nblk = smooth_fwd_blocks(H, W), per_pass = B * nblk * (kThreads / 64);
    int passes = 0;
    for (int s0 = 0; s0 < S; s0 += kSmoothMaxS, ++passes) {
        const SmoothGeom G{H, W, (W + 63) / 64, S, s0, std::min(kSmoothMaxS, S - s0), kSmoothRows};
        float* out = partials + (size_t)passes * per_pass;
        dim3 grid(nblk, B), blk(kThreads);
        switch (G.sn) {
            case 1: hipLaunchKernelGGL(smooth_tiled_fwd_kernel<1>, grid, blk, 0, st, depths, targets, G, out); break;
            case 2: hipLaunchKernelGGL(smooth_tiled_fwd_kernel<2>, grid, blk, 0, st, depths, targets, G, out); break;
            case 3: hipLaunchKernelGGL(smooth_tiled_fwd_kernel<3>, grid, blk, 0, st, depths, targets, G, out); break;
            default: hipLaunchKernelGGL(smooth_tiled_fwd_kernel<4>, grid, blk, 0, st, depths, targets, G, out); break;
        }
        DS_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(sum_finalize_kernel, dim3(1), dim3(kSumThreads), 0, st, partials, passes * per_pass,
                       1.0 / ((double)N * H * W * 3.0), value);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

int ds_smooth_batch_bwd(const float* depths, const float* targets, const float* g_value, int B, int S, int C, int H, int W,
                        int accumulate, float* g_depths, void* stream) {
    if (!depths || !targets || !g_value || !g_depths || B <= 0 || S <= 0 || (long long)B * S > 65535 ||
        bad_img(B * S, C, H, W))
        return DS_ERR_INVALID_ARG;
    const int N = B * S;
    hipStream_t st = (hipStream_t)stream;
    if (C != 3) {
        hipLaunchKernelGGL(smooth_bwd_kernel, dim3(pix_blocks(H * W), N), dim3(kThreads), 0, st, depths, targets, g_value, N, S,
                           C, H, W, accumulate, g_depths);
        DS_CHECK_LAUNCH();
        return DS_OK;
    }
    const float inv_count = (float)(1.0 / ((double)N * H * W * 3.0));
    const int rows = smooth_bwd_rows(B, H, W);
    for (int s0 = 0; s0 < S; s0 += kSmoothMaxS) {
        const SmoothGeom G{H, W, (W + 61) / 62, S, s0, std::min(kSmoothMaxS, S - s0), rows};
        dim3 grid(smooth_bwd_blocks(H, W, rows), B), blk(kThreads);
        switch (G.sn) {
            case 1: hipLaunchKernelGGL(smooth_tiled_bwd_kernel<1>, grid, blk, 0, st, depths, targets, g_value, inv_count, G, accumulate, g_depths); break;
            case 2: hipLaunchKernelGGL(smooth_tiled_bwd_kernel<2>, grid, blk, 0, st, depths, targets, g_value, inv_count, G, accumulate, g_depths); break;
            case 3: hipLaunchKernelGGL(smooth_tiled_bwd_kernel<3>, grid, blk, 0, st, depths, targets, g_value, inv_count, G, accumulate, g_depths); break;
            default: hipLaunchKernelGGL(smooth_tiled_bwd_kernel<4>, grid, blk, 0, st, depths, targets, g_value, inv_count, G, accumulate, g_depths); break;
        }
        DS_CHECK_LAUNCH();
    }
    return DS_OK;
}

}  // extern "C"

size_t ds::smooth_fb_partials_count(int B, int S, int H, int W) {
    if (B <= 0 || S <= 0 || H < 2 || W < 2) return 0;
    const int passes = (S + kSmoothMaxS - 1) / kSmoothMaxS;
    return (size_t)passes * B * smooth_bwd_blocks(H, W, 4) * (kThreads / 64);   // the shortest walk has the most waves
}

int ds::smooth_fb_partials(const float* depths, const float* targets, float scale, int B, int S, int H, int W, float* g_depths,
                           float* partials, int* n_partials, hipStream_t st) {
    if (!depths || !targets || !g_depths || !partials || B <= 0 || S <= 0 || (long long)B * S > 65535 || bad_img(B * S, 3, H, W))
        return DS_ERR_INVALID_ARG;
    const float g = scale * (float)(1.0 / ((double)B * S * H * W * 3.0));
    const int rows = smooth_bwd_rows(B, H, W), nblk = smooth_bwd_blocks(H, W, rows), per_pass = B * nblk * (kThreads / 64);
    int passes = 0;
    for (int s0 = 0; s0 < S; s0 += kSmoothMaxS, ++passes) {
        const SmoothGeom G{H, W, (W + 61) / 62, S, s0, std::min(kSmoothMaxS, S - s0), rows};
        float* out = partials + (size_t)passes * per_pass;
        dim3 grid(nblk, B), blk(kThreads);
        switch (G.sn) {
            case 1: hipLaunchKernelGGL(smooth_tiled_fb_kernel<1>, grid, blk, 0, st, depths, targets, g, G, g_depths, out); break;
            case 2: hipLaunchKernelGGL(smooth_tiled_fb_kernel<2>, grid, blk, 0, st, depths, targets, g, G, g_depths, out); break;
            case 3: hipLaunchKernelGGL(smooth_tiled_fb_kernel<3>, grid, blk, 0, st, depths, targets, g, G, g_depths, out); break;
            default: hipLaunchKernelGGL(smooth_tiled_fb_kernel<4>, grid, blk, 0, st, depths, targets, g, G, g_depths, out); break;
        }
        DS_CHECK_LAUNCH();
    }
    *n_partials = passes * per_pass;
    return DS_OK;
}

extern "C" {

int ds_smooth_fwd(const float* depths, const float* targets, int N, int C, int H, int W, float* value, void* workspace,
                  size_t workspace_bytes, void* stream) {
    return ds_smooth_batch_fwd(depths, targets, N, 1, C, H, W, value, workspace, workspace_bytes, stream);
}

int ds_smooth_bwd(const float* depths, const float* targets, const float* g_value, int N, int C, int H, int W,
                  float* g_depths, void* stream) {
    return ds_smooth_batch_bwd(depths, targets, g_value, N, 1, C, H, W, 0, g_depths, stream);
}

size_t ds_metrics_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    // block partials (8 floats each), then 8 doubles per image
    return (size_t)B * metric_blocks(H * W) * 8 * sizeof(float) + (size_t)B * 8 * sizeof(double);
}

int ds_depth_metrics_eps(const float* pred, const float* gt, int B, int H, int W, float eps, float* out, long long* counts,
                         void* workspace, size_t workspace_bytes, void* stream) {
    if (!pred || !gt || !out || bad_img(B, 1, H, W)) return DS_ERR_INVALID_ARG;
    if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < ds_metrics_workspace_bytes(B, H, W))
        return DS_ERR_WORKSPACE;
    return depth_metrics_launch(pred, gt, B, H, W, eps, out, counts, workspace, (hipStream_t)stream);
}

int ds_depth_metrics(const float* pred, const float* gt, int B, int H, int W, float* out, long long* counts,
                     void* workspace, size_t workspace_bytes, void* stream) {
    return ds_depth_metrics_eps(pred, gt, B, H, W, 1e-6f, out, counts, workspace, workspace_bytes, stream);
}

int ds_depth_metrics_noalloc(const float* pred, const float* gt, int B, int H, int W, float* out, long long* counts,
                             void* stream) {
    if (!pred || !gt || !out || bad_img(B, 1, H, W)) return DS_ERR_INVALID_ARG;
    if (ds_metrics_workspace_bytes(B, H, W) > kScratchBytes) return DS_ERR_SCRATCH_LIMIT;
    void* ws = nullptr;
    const int rc = scratch_device_ptr(&ws);
    if (rc != DS_OK) return rc;
    return depth_metrics_launch(pred, gt, B, H, W, 1e-6f, out, counts, ws, (hipStream_t)stream);
}

int ds_ingest_u8(const uint8_t* frames, int N, int H, int W, float* out, void* stream) {
    if (!frames || !out || N <= 0 || N > 65535 || H <= 0 || W <= 0 || (long long)H * W > (1ll << 30)) return DS_ERR_INVALID_ARG;
    if (((uintptr_t)frames & 3) || ((uintptr_t)out & 15)) return DS_ERR_INVALID_ARG;
    const int HW = H * W, groups = (HW + 3) / 4;
    hipLaunchKernelGGL(ingest_u8_kernel, dim3((groups + kThreads - 1) / kThreads, N), dim3(kThreads), 0, (hipStream_t)stream,
                       frames, HW, out);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

size_t ds_ingest_resize_workspace_bytes(int N, int Hin, int Win, int Hout, int Wout) {
    if (N <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0) return 0;
    return (size_t)N * Hin * Wout * 3;  // the horizontal pass's uint8 result
}

int ds_ingest_resize_u8(const uint8_t* frames, int N, int Hin, int Win, int Hout, int Wout, uint8_t* out_u8, float* out_f32,
                        void* workspace, size_t workspace_bytes, void* stream) {
    if (!frames || (!out_u8 && !out_f32) || N <= 0 || N > 65535 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 ||
        Hout > 65535 || (long long)Hin * Win > (1ll << 30) || (long long)Hout * Wout > (1ll << 30))
        return DS_ERR_INVALID_ARG;
    PilAxis ax, ay;
    if (!pil_axis(ax, Win, Wout) || !pil_axis(ay, Hin, Hout)) return DS_ERR_UNSUPPORTED;  // more than 64 taps
    if (!workspace || workspace_bytes < ds_ingest_resize_workspace_bytes(N, Hin, Win, Hout, Wout)) return DS_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    uint8_t* tmp = (uint8_t*)workspace;
    const int rows_per_block = 32;
    hipLaunchKernelGGL(pil_resize_h_kernel, dim3((Wout + 63) / 64, (Hin + rows_per_block - 1) / rows_per_block, N),
                       dim3(kThreads), 0, st, frames, ax, Hin, rows_per_block, tmp);
    DS_CHECK_LAUNCH();
    hipLaunchKernelGGL(pil_resize_v_kernel, dim3((Wout * 3 + kThreads - 1) / kThreads, Hout, N), dim3(kThreads), 0, st,
                       (const uint8_t*)tmp, ay, Wout, out_u8, out_f32);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

int ds_ingest_depth_u16(const uint16_t* depth, int N, int H, int W, float* out, void* stream) {
    if (!depth || !out || N <= 0 || N > 65535 || H <= 0 || W <= 0 || (long long)H * W > (1ll << 30)) return DS_ERR_INVALID_ARG;
    if (((uintptr_t)depth & 3) || ((uintptr_t)out & 7)) return DS_ERR_INVALID_ARG;
    const size_t n = (size_t)N * H * W, pairs = (n + 1) / 2;
    hipLaunchKernelGGL(ingest_depth_u16_kernel, dim3((unsigned)((pairs + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, depth, n, out);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

int ds_resize_bilinear(const float* in, int N, int Hin, int Win, int Hout, int Wout, float* out, void* stream) {
    ResizeGeom G;
    if (!in || !out || !resize_geom(G, N, Hin, Win, Hout, Wout)) return DS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(pix_blocks(Hout * Wout), N), dim3(kThreads), 0, (hipStream_t)stream, in, G,
                       out);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

int ds_resize_bicubic(const float* in, int N, int Hin, int Win, int Hout, int Wout, float* out, void* stream) {
    ResizeGeom G;
    if (!in || !out || !resize_geom(G, N, Hin, Win, Hout, Wout)) return DS_ERR_INVALID_ARG;
    hipLaunchKernelGGL(resize_bicubic_kernel, dim3(pix_blocks(Hout * Wout), N), dim3(kThreads), 0, (hipStream_t)stream, in, G,
                       out);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

}  // extern "C"
