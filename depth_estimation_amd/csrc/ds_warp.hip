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

#ifndef DS_IMG_EXP
#define DS_IMG_EXP 0   // 32: diagnostic build of img_reduce_kernel (WRONG g_src: every wave's first texels carry its lifetime
                       // in cycles, the point lists and entries it looked at and its hits; read by scripts/img_diag.py)
#endif
#ifndef DS_RED_ROWS
#define DS_RED_ROWS 2
#endif

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
// atomics (~20 cycles per wave instruction) then cost what the global ones did (213 us).  What is done instead: for a smooth
// warp the taps of the 64 pixels of a wave row fall into a window of three source rows x 66 columns that moves down one row
// per pixel row, so
//   * lane x owns source column x + cb of the window; the taps of lanes x-1, x, x+1 that fall into that column are
//     combined in registers (two DPP lane shifts per value), three running sums per channel (rows r0, r0+1, r0+2);
//   * a pixel whose taps do not fit the window (a near pixel among far ones: its disparity differs by more than a texel)
//     is broadcast from its lane and taken over by the two lanes that own its columns if rows and columns are still
//     inside the window, otherwise it leaves the wave on its own ("point"); if more than a few lanes of a row misfit (strong
//     minification, source larger than the grid) the whole row leaves as points.  The result never depends on the path.
//
// Rounds 1-2 sent finished window rows and points to memory with float atomics (one coalesced atomic per texel and channel
// for a row, 4 per channel for a point): 2.0 M requests at the memory side per launch, 66 of the kernel's 127 us, and sums
// whose last bits changed from run to run (the windows of neighbouring waves overlap, border texels collect hundreds of
// terms in varying order).  Round 3: NO atomics of any kind, nothing to clear.  Everything a wave produces goes with plain
// stores into staging that is the wave's own ("tile" = the 64 columns x ROWS rows of pixels a wave walks):
//   * a finished window row ("segment": 66 columns x 3 channels) into the tile's row slot k = source row - rbase, rbase =
//     two rows above the first row the wave finishes, kSegRows slots; the tile's record keeps rbase, a mask of the filled
//     slots and each slot's first column.  A row that does not fit (the window jumped backwards or far ahead, or comes to a
//     slot a second time) leaves as points;
//   * a point (one pixel's tap: PtEntry) is appended to the tile's point list, its position from a ballot prefix inside
//     the wave; the record keeps the count and the row / column range of the list.
// img_reduce_kernel then forms every texel of g_src exactly once -- one thread per texel column of a source row: it looks
// through the tile records of its image for segments of its rows that cover its columns and for point lists whose ranges
// include them, and adds what it finds IN AN ORDER THAT DEPENDS ON THE DATA ONLY (segments by tile index; a list's entries
// batch by batch, inside a batch by the integer bidding described there), so the sum is bit-identical from run to run; the texel is written with a plain store (g_src needs no clearing).  First version of the
// round: per-row directories filled through returning integer atomics and a 64-bit fixed-point sum in the reducer -- correct
// and repeatable, and 4x slower than the float atomics (the atomics' round trips serialise the row loop, the fixed-point
// conversions cost 128 us): profiles/r03_kb2_image_gradient.txt.
constexpr int kSegRows = 16;        // row slots per tile (a wave of 8 pixel rows finishes 10 window rows if the warp is smooth)
constexpr int kSegFloats = 3 * 66;  // one slot: 3 channels x (outer column, 64 owned columns, outer column)
// A tile's record, structure-of-arrays so that the reducer's lanes read the records of 64 consecutive tiles as consecutive
// 16-byte words (as an array of 112-byte structs every lane touched a cache line of its own: 47 of the reducer's 85 us):
//   hdrA[tile] = { rbase: source row of slot 0,  mask: filled slots (bit 31: the point list is not empty),
//                  cbmin, cbmax: range of cb[] over the filled slots }
//   hdrB[tile] = { npt: entries of the point list,  prmin, prmax: the rows they span,
//                  (min + 1) << 16 | (max + 1) of the columns their taps start in }
//   hdrC[tile] = 16 signed bytes: [k] = slot k's window offset cb (its first OWNED column, entry 1 of its 66, is source
//                  column 64 * bx + cb; a window further than 127 columns from its pixels gets no slot)
//   (first these were ints in an array of their own, read by dependent loads after the records: 129 -> 117 us)
// A point = one pixel's whole bilinear tap, 24 bytes: the reducer forms the four weights from the two fractions exactly as
// make_tap does (sy * ex, sy * wx, ny * ex, ny * wx with ex = 1 - wx, sy = 1 - ny) and the 12 products g[c] * w itself, so the
// sums get the bits they would have if pass 1 formed the products (first: 32 bytes per texel ROW, two entries per pixel;
// the change moved no time -- the reducer is latency, not bytes).  A tap on the last column has
// (wx, col) = (1, Ws-2) instead of make_tap's shifted pair: the same texels and products.  A single value for one texel
// (a window row without a slot, the sum of a run) is the entry (value, wx = 0, ny = 0): weight 1 on (row, col), 0 elsewhere.
struct PtEntry {
    unsigned rc;                    // row << 16 | (col + 1): row <= 32767, col in [-1, 65534] (checked on the host)
    float g[3];
    float wx, ny;
};
struct ImgStage {                   // into the caller's workspace (ds_warp_bwd_image_workspace_bytes)
    int4* hdrA;                     // [N][nwy][nbx]
    int4* hdrB;
    int4* hdrC;
    float* seg;                     // [N][nwy][nbx][kSegRows][kSegFloats]
    PtEntry* pt;                    // [N][nwy][nbx][pts_per_tile]
    int nwy;                        // tile rows per image = ceil(H / ROWS)
    int pts_per_tile;               // worst case per pixel row: three finished window rows without a slot (66 each) + 12 misfits
};

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o, 64));
    return v;
}

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
                                                            float* __restrict__ partials, ImgStage st) {
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
    // ---- staging of this wave's tile (IMG): see the comment above ImgStage
    const int wy = by * (kThreads / 64) + wave;
    const size_t tix = ((size_t)n * st.nwy + wy) * G.nbx + bx;
    int rbase = 0, npt = 0;          // uniform
    int cbmin = 0x7fffffff, cbmax = -0x7fffffff - 1;   // uniform
    unsigned smask = 0;              // uniform; rbase is valid once smask != 0 or a row has been turned away
    bool have_base = false;
    int prmin = 0x7fffffff, prmax = -0x7fffffff - 1, pcmin = 0x7fffffff, pcmax = -0x7fffffff - 1;   // per lane
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    // append one point per lane with `has` to the tile's list, in lane order
    auto push_point = [&](bool has, int row, int pcol, float g0, float g1, float g2, float wx, float ny) {
        const unsigned long long m = __ballot(has);
        if (m == 0) return;   // uniform
        if (has) {
            PtEntry* e = st.pt + tix * st.pts_per_tile + npt + __popcll(m & lt_mask);
            e->rc = ((unsigned)row << 16) | (unsigned)(pcol + 1);
            e->g[0] = g0; e->g[1] = g1; e->g[2] = g2;
            e->wx = wx; e->ny = ny;
            prmin = min(prmin, row); prmax = max(prmax, row + 1);   // (an entry reaches the row below as well)
            pcmin = min(pcmin, pcol); pcmax = max(pcmax, pcol);
        }
        npt += __popcll(m);
    };
    // the window's row 0 is finished: stage it, then move the window down one row
    auto emit_shift = [&]() {
        const int r = win.r0;
        const bool nz = win.a[0][0] != 0.0f || win.a[1][0] != 0.0f || win.a[2][0] != 0.0f;
        const bool enz = (lane == 0 || lane == 63) && (win.e[0][0] != 0.0f || win.e[1][0] != 0.0f || win.e[2][0] != 0.0f);
        if (IMG && r >= 0 && r < G.Hs && __ballot(nz || enz) != 0) {   // uniform
            if (!have_base) { rbase = r - 2; have_base = true; }
            const int k = r - rbase;
            if (k >= 0 && k < kSegRows && !((smask >> k) & 1u) && win.cb >= -128 && win.cb <= 127) {
                float* p = st.seg + (tix * kSegRows + k) * kSegFloats;
#pragma unroll
                for (int c = 0; c < 3; ++c) p[c * 66 + 1 + lane] = win.a[c][0];
                if (lane == 0 || lane == 63) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) p[c * 66 + (lane == 0 ? 0 : 65)] = win.e[c][0];
                }
                const int cb0 = bx * 64 + win.cb;
                if (lane == 0) reinterpret_cast<signed char*>(st.hdrC)[tix * 16 + k] = (signed char)win.cb;
                cbmin = min(cbmin, cb0);
                cbmax = max(cbmax, cb0);
                smask |= 1u << k;
            } else {
                // no slot for this row: it leaves as points (a lane's own column, then the two outer columns)
                const int col = x + win.cb;
                push_point(nz, r, col, win.a[0][0], win.a[1][0], win.a[2][0], 0.f, 0.f);
                push_point(enz, r, col + (lane == 0 ? -1 : 1), win.e[0][0], win.e[1][0], win.e[2][0], 0.f, 0.f);
            }
        }
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
            constexpr int kMinRunLanes = 4;   // lanes continuing a run of equal taps before the wave sums runs (below)
            auto misfits = [&](int r0_, int cb_) {
                rel = dx - (cb_ - 1);
                rr = t.y0 - r0_;
                lane_ok = valid && (unsigned)rel <= 1u && (unsigned)rr <= 1u;
                return __popcll(__ballot(valid && !lane_ok));
            };
            bool ok = false;
            int nemit = 0;            // finished window rows to stage before this row's taps are added (uniform)
            int nr0 = 0, ncb = 0;     // the window after that
            if (win.live) {
                if (misfits(win.r0 + 1, win.cb) <= kMaxMisfits) {   // the usual case: the window moves down one row
                    nemit = 1;
                    nr0 = win.r0 + 1;
                    ncb = win.cb;
                    ok = true;
                } else if (misfits(win.r0, win.cb) <= kMaxMisfits) {   // or stays
                    nr0 = win.r0;
                    ncb = win.cb;
                    ok = true;
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
                    nemit = win.live ? 3 : 0;
                    nr0 = br;
                    ncb = bc;
                    win.live = true;
                    ok = true;
                }
            }
#pragma unroll 1
            for (int k = 0; k < nemit; ++k) emit_shift();   // the one place finished rows leave from (inlined once)
            if (ok) {
                win.r0 = nr0;      // (after three emits the window is empty)
                win.cb = ncb;
                misfits(nr0, ncb);
            }
            bool direct = valid && !ok;  // pixels that leave as points
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
                    fromL[s] = lane_left(kr[s]);    // what lane x-1 sends to its right neighbour's column (lane 0: 0)
                    fromR[s] = lane_right(kl[s]);   // what lane x+1 sends to its left neighbour's column (lane 63: 0)
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float gc = lane_ok ? g[c] : 0.f;
                    const float gl = lane_left(gc), gr = lane_right(gc);   // DPP wave shifts: one VALU operation each
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
            {
                // points: every pixel that did not go through the window leaves its tap (PtEntry)
                const bool any = direct && (g[0] != 0.0f || g[1] != 0.0f || g[2] != 0.0f);
                // A run of neighbouring lanes on ONE tap (pixels beyond the left or right border all clamp to the border
                // column; a far plane collapses onto a texel) would reach the reducer as that many entries for one texel,
                // which it then adds one after the other (chunk 0 of a forward-moving camera: 640 k cycles for a wave
                // against a median of 27 k).  Runs of four lanes or more are summed here first, by a segmented scan over
                // the lanes -- a fixed tree, so still the same bits every run -- and the run's last lane leaves the four
                // sums as single values.
                // (the three lane shifts with every lane active: a lane switched off by a short-circuit reads as 0)
                const bool prev_any = lane_left_i((int)any) != 0;
                const int prev_y0 = lane_left_i(t.y0), prev_xl = lane_left_i(t.xl);
                const bool cont = any && lane > 0 && prev_any && prev_y0 == t.y0 && prev_xl == t.xl;
                const unsigned long long cm = __ballot(cont);
                const bool scan = __popcll(cm) >= kMinRunLanes - 1;   // uniform: a run of kMinRunLanes may exist
                bool in_long = false, last = true;
                int head = lane;
                if (scan) {
                    head = 63 - __builtin_clzll(~cm & (lt_mask | (1ull << lane)));   // first lane of this lane's run
                    const unsigned long long ends = ~(cm >> 1) | (1ull << 63);        // lanes whose successor does not continue
                    const int end = lane + __builtin_ctzll(ends >> lane);
                    last = end == lane;
                    in_long = any && end - head + 1 >= kMinRunLanes;
                }
                const float wxe = t.shift ? 1.0f : t.wx;   // (shift: make_tap's pair starts at Ws-2 with the weight on its right texel)
                push_point(any && !in_long, t.y0, t.xl, g[0], g[1], g[2], wxe, t.ny);
                if (scan && __ballot(in_long) != 0) {   // uniform
                    auto run_sums = [&](int row, float wa, float wb) {
                        float v[6];
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            v[2 * c] = in_long ? g[c] * wa : 0.f;
                            v[2 * c + 1] = in_long ? g[c] * wb : 0.f;
                        }
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) {
                            const bool take = lane - o >= head;
#pragma unroll
                            for (int q = 0; q < 6; ++q) {
                                const float u = __shfl_up(v[q], o, 64);
                                v[q] += take ? u : 0.f;
                            }
                        }
                        const bool lead = in_long && last;
                        push_point(lead && (v[0] != 0.0f || v[2] != 0.0f || v[4] != 0.0f), row, t.xl, v[0], v[2], v[4], 0.f, 0.f);
                        push_point(lead && (v[1] != 0.0f || v[3] != 0.0f || v[5] != 0.0f), row, t.xl + 1, v[1], v[3], v[5], 0.f, 0.f);
                    };
                    run_sums(t.y0, t.wa_t, t.wb_t);
                    run_sums(t.y1, t.wa_b, t.wb_b);
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
    if (IMG && yb < G.H) {
        if (win.live) {
#pragma unroll 1
            for (int k = 0; k < 3; ++k) emit_shift();
        }
        prmin = wave_min_i(prmin); prmax = wave_max_i(prmax);
        pcmin = wave_min_i(pcmin); pcmax = wave_max_i(pcmax);
        if (lane == 0) {
            st.hdrA[tix] = make_int4(rbase, (int)(smask | (npt > 0 ? 0x80000000u : 0u)), cbmin, cbmax);
            st.hdrB[tix] = make_int4(npt, prmin, prmax, npt > 0 ? ((pcmin + 1) << 16) | (pcmax + 1) : 0);
        }
    }
    block_reduce12(gM, partials + ((size_t)n * gridDim.x + blockIdx.x) * 12, lds);
}

// ---------------------------------------------------------------------------------------------------
// g_src from the staged segments and points (see ImgStage): one thread per texel column of one source row, all three
// channels, a wave per 64 columns.  Every texel of g_src is written; the order of the additions is fixed.
__device__ __forceinline__ int4 buf_load_i32x4(rsrc_t r, unsigned lane_bytes, unsigned uniform_bytes) {
    return __builtin_bit_cast(int4, __builtin_amdgcn_raw_buffer_load_b128(r, lane_bytes, uniform_bytes, 0));
}
constexpr int kRedRows = DS_RED_ROWS;   // source rows per wave of the reducer
constexpr int kMaxCand = 12;            // segments per (row, 64 columns) kept from the scan (2-4 for a smooth warp; max 6 seen)
constexpr int kCandFirst = 3;           // of those, fetched for all rows at once
constexpr int kPtBatches = 4;           // point-list batches (64 entries each) whose loads are in flight together
// The reducer is bound by the latency of its dependent loads (tile records -> the slot's values; a point list, batch after
// batch), not by bytes: one source row per wave (46 k short waves at 12 x 2 x 192 x 640) took 80 us,
// eight rows per wave handled one after the other 174 us.  So a wave takes 64 columns x kRedRows rows and works in phases that
// each keep all their loads in flight:
//   (1) the records of all tiles of its image, 256 at a time: every lane whose tile holds a segment for one of the rows
//       leaves (tile, slot) in the row's candidate list in LDS (position from a ballot prefix: tile order);
//   (P) the point lists whose row and column ranges meet the wave's texels, kPtBatches x 64 entries at a time, added to the
//       wave's texels in LDS (below);
//   (2) lane f takes entry f of the lists (tile, slot, first column: all from the records); then the values of the first kCandFirst
//       candidates of ALL rows are requested together and added in list order (further candidates, rare, row by row).
__global__ __launch_bounds__(kThreads) void img_reduce_kernel(ImgStage st, int nbx, int Hs, int Ws, float* __restrict__ g_src) {
    constexpr int kFlat = kRedRows * kMaxCand, kE = (kFlat + 63) / 64;
    __shared__ int lds_tk[kThreads / 64][kFlat];                // (tile << 4) | slot, row i's list at i * kMaxCand
    __shared__ int lds_cb[kThreads / 64][kFlat];                // the slot's first owned column
    __shared__ float4 lds_pt[kThreads / 64][kRedRows][64];      // the wave's texels: what the points add (x, y, z) and, in
                                                                // w, the texel's bidding word (below)
    const int ngroups = (Hs + kRedRows - 1) / kRedRows;
    const int n = blockIdx.y / ngroups, r0 = (blockIdx.y - n * ngroups) * kRedRows;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int c0 = blockIdx.x * kThreads + wave * 64, col = c0 + lane;
    if (c0 >= Ws) return;   // wave-uniform
    const int T = st.nwy * nbx;
    const long long t_start = (DS_IMG_EXP & 32) ? (long long)__builtin_amdgcn_s_memtime() : 0;
    const int4* hdrA = st.hdrA + (size_t)n * T;
    const int4* hdrB = st.hdrB + (size_t)n * T;
    const int4* hdrC = st.hdrC + (size_t)n * T;
    const float* segn = st.seg + (size_t)n * T * kSegRows * kSegFloats;
    // buffer loads (descriptor in scalar registers + one 32-bit offset per lane) for everything that is fetched in bulk:
    // with 64-bit addresses the 36 value loads of phase 2 alone held 72 registers and the kernel came to 254 + 52
    const rsrc_t hA = make_rsrc(hdrA), hC = make_rsrc(hdrC), sgr = make_rsrc(segn);
    const rsrc_t ptr_ = make_rsrc(st.pt + (size_t)n * T * st.pts_per_tile);
    const unsigned long long lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    float acc[kRedRows][3];
    int cnt[kRedRows];   // uniform
    int dg_tiles = 0, dg_pts = 0, dg_hits = 0;
    bool pt_used = false;    // uniform
    unsigned epoch = 0;      // uniform: rounds of the point adds so far
#pragma unroll
    for (int i = 0; i < kRedRows; ++i) {
        acc[i][0] = acc[i][1] = acc[i][2] = 0.f;
        cnt[i] = 0;
    }
    constexpr int kU = 4;   // tile records looked at per lane and round: their loads are in flight together
    for (int base = 0; base < T; base += 64 * kU) {
        int4 A[kU], Cq[kU];
        // unconditional loads from clamped indices (a predicated load becomes a branch and a wait: the kU loads would go out
        // one after the other), lanes beyond the last tile get an empty record afterwards
#pragma unroll
        for (int u = 0; u < kU; ++u) A[u] = buf_load_i32x4(hA, (unsigned)min(base + u * 64 + lane, T - 1) * 16u, 0u);   // rbase, mask, cbmin, cbmax
#pragma unroll
        for (int u = 0; u < kU; ++u) Cq[u] = buf_load_i32x4(hC, (unsigned)min(base + u * 64 + lane, T - 1) * 16u, 0u);  // the slots' offsets
#pragma unroll
        for (int u = 0; u < kU; ++u)
            if (base + u * 64 + lane >= T) A[u] = make_int4(0, 0, 0, 0);
        // ---- phase 1: candidate segments of every row, in tile order (no loads in here)
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int t = base + u * 64 + lane;
            const bool cols = A[u].z - 1 <= c0 + 63 && A[u].w + 64 >= c0;   // the tile's segments may reach these columns
            const int klo = r0 - A[u].x;
            const unsigned smk = (unsigned)A[u].y & 0xffffu;
            const unsigned want = !cols || klo >= kSegRows || klo + kRedRows <= 0 ? 0u
                                  : (klo >= 0 ? (smk >> klo) : (smk << -klo)) & ((1u << kRedRows) - 1u);
            if (__ballot(want != 0u) == 0) continue;   // uniform
#pragma unroll
            for (int i = 0; i < kRedRows; ++i) {
                const bool c = (want >> i) & 1u;
                const unsigned long long m = __ballot(c);
                if (m == 0) continue;   // uniform
                const int pos = cnt[i] + __popcll(m & lt_mask);
                if (c && pos < kMaxCand) {
                    const int k = klo + i;
                    const int word = k < 8 ? (k < 4 ? Cq[u].x : Cq[u].y) : (k < 12 ? Cq[u].z : Cq[u].w);
                    lds_tk[wave][i * kMaxCand + pos] = (t << 4) | k;
                    lds_cb[wave][i * kMaxCand + pos] = (t % nbx) * 64 + ((word << (24 - 8 * (k & 3))) >> 24);   // sign-extended byte k
                }
                cnt[i] += __popcll(m);
            }
        }
        // ---- points: lists whose ranges include these rows and columns, tiles in index order, entries in list order
        unsigned long long pmask[kU];   // uniform: tiles of batch u whose list spans rows of this wave
        int nptu[kU];
        {
            const rsrc_t hB = make_rsrc(hdrB);
            int4 Bq[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) Bq[u] = buf_load_i32x4(hB, (unsigned)min(base + u * 64 + lane, T - 1) * 16u, 0u);   // npt, prmin, prmax, -
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                // (w: the columns the list's taps start in, (min + 1) << 16 | (max + 1); a tap reaches one column further)
                pmask[u] = __ballot(A[u].y < 0 && Bq[u].x > 0 && Bq[u].y <= r0 + kRedRows - 1 && r0 <= Bq[u].z &&
                                    (Bq[u].w >> 16) - 1 <= c0 + 63 && (Bq[u].w & 0xffff) >= c0);
                nptu[u] = Bq[u].x;
            }
        }
        // (rolled: unrolled over u the body below existed kU times and the entry registers of all copies were live at once,
        // 254 + 52 registers = one wave per SIMD)
#pragma unroll 1
        for (int u = 0; u < kU; ++u) {
            unsigned long long pm = 0;
            int nptv = 0;
#pragma unroll
            for (int uu = 0; uu < kU; ++uu) {
                if (uu == u) {   // uniform
                    pm = pmask[uu];
                    nptv = nptu[uu];
                }
            }
            const int tb = base + u * 64;
            while (pm) {
                const int m = __builtin_ctzll(pm);
                pm &= pm - 1;
                const int t = tb + m;
                const int npts = __builtin_amdgcn_readlane(nptv, m);
                if (DS_IMG_EXP & 32) { dg_tiles += 1; dg_pts += npts; }
                const unsigned pl = (unsigned)t * (unsigned)st.pts_per_tile * (unsigned)sizeof(PtEntry);   // this tile's list inside the image's
                for (int pb = 0; pb < npts; pb += 64 * kPtBatches) {
                    int4 ea[kPtBatches];       // whole entries, from clamped indices: rc, g[0..2]
                    float2_t eb[kPtBatches];   // wx, ny
#pragma unroll
                    for (int q = 0; q < kPtBatches; ++q) {
                        const unsigned eo = (unsigned)min(pb + q * 64 + lane, npts - 1) * (unsigned)sizeof(PtEntry);
                        ea[q] = buf_load_i32x4(ptr_, eo, pl);
                        eb[q] = buf_load_f32x2(ptr_, eo, pl + 16u);
                    }
#pragma unroll
                    for (int q = 0; q < kPtBatches; ++q) {
                        const int pr = (int)((unsigned)ea[q].x >> 16) - r0, pc = (int)((unsigned)ea[q].x & 0xffffu) - 1;
                        // the tap covers rows pr, pr + 1 and columns pc, pc + 1
                        const bool hit = pb + q * 64 + lane < npts && pr + 1 >= 0 && pr < kRedRows && pc + 1 >= c0 && pc <= c0 + 63;
                        const unsigned long long hits = __ballot(hit);
                        if (DS_IMG_EXP & 32) dg_hits += __popcll(hits);
                        if (hits == 0) continue;   // uniform
                        const float g0 = __builtin_bit_cast(float, ea[q].y), g1 = __builtin_bit_cast(float, ea[q].z),
                                    g2 = __builtin_bit_cast(float, ea[q].w);
                        const float wx = eb[q].x, ny = eb[q].y, ex = 1.0f - wx, sy = 1.0f - ny;   // as make_tap forms them
                        const float wgt[4] = {sy * ex, sy * wx, ny * ex, ny * wx};
                        if (!pt_used) {   // uniform: first entries for this wave's texels
                            pt_used = true;
#pragma unroll
                            for (int i = 0; i < kRedRows; ++i) lds_pt[wave][i][lane] = make_float4(0.f, 0.f, 0.f, 0.f);
                        }
                        // The entries of this batch are added to the wave's texels in LDS, all lanes and all four corners of
                        // their taps at once.  Additions to the SAME texel have to happen one after the other and in an order
                        // that depends on nothing but the data: each round every pending (lane, corner) bids for its texel
                        // with an integer max (order-independent, unlike a float atomic) of (round, corner, 63 - lane);
                        // exactly one bid per texel survives, its owner adds with a plain read-modify-write and leaves, the
                        // others bid again.  The four bids, the four reads and the writes of a round are each issued
                        // together (a corner at a time: four times the LDS round trips; one thread adding hit after hit:
                        // 600 cycles a hit).  Rounds = the largest number of additions to one texel in the batch: two for a
                        // smooth warp (a tap's right texel is its neighbour's left one), more where pixels pile up (pass 1
                        // has summed the long runs already).  A corner with weight 0 adds nothing and is left out.
                        bool pending[4];
                        float4* cell[4];
                        unsigned anyp = 0;   // uniform
#pragma unroll
                        for (int side = 0; side < 4; ++side) {
                            const int ry = pr + (side >> 1), cx = pc + (side & 1) - c0;
                            pending[side] = hit && wgt[side] != 0.0f && ry >= 0 && ry < kRedRows && cx >= 0 && cx <= 63;
                            cell[side] = &lds_pt[wave][min(max(ry, 0), kRedRows - 1)][min(max(cx, 0), 63)];
                            anyp |= __ballot(pending[side]) != 0 ? 1u << side : 0u;
                        }
                        while (anyp) {
                            epoch += 1u;
                            unsigned bid[4];
#pragma unroll
                            for (int side = 0; side < 4; ++side) {
                                bid[side] = (epoch << 8) | ((unsigned)(3 - side) << 6) | (unsigned)(63 - lane);
                                if (pending[side])
                                    __hip_atomic_fetch_max(reinterpret_cast<unsigned*>(&cell[side]->w), bid[side], __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_WAVEFRONT);
                            }
                            asm volatile("" ::: "memory");   // (the bidding word is cell->w: nothing kept across the bids)
                            float4 cur[4];
#pragma unroll
                            for (int side = 0; side < 4; ++side) cur[side] = *cell[side];   // the texel and its bidding word in one read
                            asm volatile("" ::: "memory");
                            anyp = 0;
#pragma unroll
                            for (int side = 0; side < 4; ++side) {
                                const bool win = pending[side] && __builtin_bit_cast(unsigned, cur[side].w) == bid[side];
                                const float w = wgt[side];
                                // (winners own distinct texels, so none of these writes meets another's read above)
                                if (win) *cell[side] = make_float4(cur[side].x + g0 * w, cur[side].y + g1 * w, cur[side].z + g2 * w, cur[side].w);
                                pending[side] = pending[side] && !win;
                                anyp |= __ballot(pending[side]) != 0 ? 1u << side : 0u;
                            }
                            asm volatile("" ::: "memory");
                        }
                    }
                }
            }
        }
    }
    // ---- phase 2: lane f holds entry f of the candidate lists
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int tkf[kE], cbf[kE];
#pragma unroll
    for (int e = 0; e < kE; ++e) {
        const int f = min(e * 64 + lane, kFlat - 1), fi = f / kMaxCand, fq = f - fi * kMaxCand;
        int ci = 0;
#pragma unroll
        for (int i = 0; i < kRedRows; ++i) ci = fi == i ? cnt[i] : ci;
        tkf[e] = fq < ci ? lds_tk[wave][f] : 0;      // (an unused entry reads tile 0, slot 0: memory that exists)
        cbf[e] = fq < ci ? lds_cb[wave][f] : 0;
    }
    auto entry = [&](int f, int& tk, int& cb) {   // f is a compile-time constant wherever this is called
        tk = __builtin_amdgcn_readlane(tkf[f >> 6], f & 63);
        cb = __builtin_amdgcn_readlane(cbf[f >> 6], f & 63);
    };
    {
        float v[kRedRows][kCandFirst][3];
        bool in[kRedRows][kCandFirst];
#pragma unroll
        for (int i = 0; i < kRedRows; ++i) {
#pragma unroll
            for (int q = 0; q < kCandFirst; ++q) {
                int tk, cb;
                entry(i * kMaxCand + q, tk, cb);
                const int j = col - (cb - 1);
                in[i][q] = q < cnt[i] && j >= 0 && j <= 65;
                const unsigned slot = (unsigned)((tk >> 4) * kSegRows + (tk & 15)) * (unsigned)kSegFloats;   // uniform
#pragma unroll
                for (int c = 0; c < 3; ++c) v[i][q][c] = buf_load_f32(sgr, (unsigned)min(max(j, 0), 65) * 4u, slot + c * 66u);
            }
        }
#pragma unroll
        for (int i = 0; i < kRedRows; ++i)
#pragma unroll
            for (int q = 0; q < kCandFirst; ++q)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[i][c] += in[i][q] ? v[i][q][c] : 0.0f;   // (+ 0.0f where nothing was found: exact)
    }
    bool more = false, over = false;
#pragma unroll
    for (int i = 0; i < kRedRows; ++i) {
        more = more || cnt[i] > kCandFirst;
        over = over || cnt[i] > kMaxCand;
    }
    if (more) {   // uniform: the rest of the lists, row by row, three candidates at a time (rolled: this is the rare path)
#pragma unroll 1
        for (int i = 0; i < kRedRows; ++i) {
            int ci = 0;
#pragma unroll
            for (int ii = 0; ii < kRedRows; ++ii) ci = ii == i ? cnt[ii] : ci;
            const int nc = min(ci, kMaxCand);
            float rest[3] = {0.f, 0.f, 0.f};
#pragma unroll 1
            for (int q0 = kCandFirst; q0 < nc; q0 += 3) {
                float v[3][3];
                bool in[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int f = i * kMaxCand + min(q0 + q, nc - 1);
                    int tk = 0, cb = 0;
#pragma unroll
                    for (int e = 0; e < kE; ++e) {
                        if ((f >> 6) == e) {   // uniform
                            tk = __builtin_amdgcn_readlane(tkf[e], f & 63);
                            cb = __builtin_amdgcn_readlane(cbf[e], f & 63);
                        }
                    }
                    const int j = col - (cb - 1);
                    in[q] = q0 + q < nc && j >= 0 && j <= 65;
                    const unsigned slot = (unsigned)((tk >> 4) * kSegRows + (tk & 15)) * (unsigned)kSegFloats;   // uniform
#pragma unroll
                    for (int c = 0; c < 3; ++c) v[q][c] = buf_load_f32(sgr, (unsigned)min(max(j, 0), 65) * 4u, slot + c * 66u);
                }
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int c = 0; c < 3; ++c) rest[c] += in[q] ? v[q][c] : 0.0f;
            }
#pragma unroll
            for (int ii = 0; ii < kRedRows; ++ii) {
                if (ii == i) {   // uniform
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[ii][c] += rest[c];
                }
            }
        }
    }
    // a row with more candidates than the list holds (a strongly minifying warp): those beyond it, found again, in tile order
    if (over) {   // uniform
        for (int i = 0; i < kRedRows; ++i) {
            int seen = 0;
            for (int tb = 0; tb < T; tb += 64) {
                const int t = tb + lane;
                const int4 a = hdrA[min(t, T - 1)];
                const int k = r0 + i - a.x;
                const bool c = t < T && k >= 0 && k < kSegRows && ((((unsigned)a.y & 0xffffu) >> k) & 1u) &&
                               a.z - 1 <= c0 + 63 && a.w + 64 >= c0;
                unsigned long long todo = __ballot(c);
                while (todo) {
                    const int m = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    if (seen++ < kMaxCand) continue;   // already added from the list
                    const int kk = __builtin_amdgcn_readlane(k, m);
                    const int j = col - (((tb + m) % nbx) * 64 + (int)reinterpret_cast<const signed char*>(hdrC)[(tb + m) * 16 + kk] - 1);
                    if (j >= 0 && j <= 65) {
                        const float* p = segn + ((size_t)(tb + m) * kSegRows + kk) * kSegFloats + j;
                        for (int ii = 0; ii < kRedRows; ++ii)
                            if (ii == i)
                                for (int c = 0; c < 3; ++c) acc[ii][c] += p[c * 66];
                    }
                }
            }
        }
    }
    asm volatile("" ::: "memory");
    if (pt_used) {   // uniform
#pragma unroll
        for (int i = 0; i < kRedRows; ++i) {
            const float4 cell = lds_pt[wave][i][lane];
            acc[i][0] += cell.x;
            acc[i][1] += cell.y;
            acc[i][2] += cell.z;
        }
    }
    if (DS_IMG_EXP & 32) {   // diagnostic build: cycles this wave lived (channel 0), point tiles / points scanned / hits
        acc[0][0] = (float)((long long)__builtin_amdgcn_s_memtime() - t_start);
        acc[0][1] = (float)dg_tiles;
        acc[0][2] = (float)dg_pts;
        if (kRedRows > 1) acc[kRedRows > 1 ? 1 : 0][0] = (float)dg_hits;
    }
    if (col < Ws) {
#pragma unroll
        for (int i = 0; i < kRedRows; ++i) {
            if (r0 + i < Hs) {
                float* o = g_src + ((size_t)n * 3 * Hs + r0 + i) * Ws + col;
#pragma unroll
                for (int c = 0; c < 3; ++c) o[(size_t)c * Hs * Ws] = acc[i][c];
            }
        }
    }
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

inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// small launches (fewer workgroups than would give every SIMD two waves) take the short tiles; only the 3-channel kernels
// are instantiated for them (the generic-C kernel keeps kBwdRows, and its grid with it)
inline bool bwd_small(int N, int C, int H, int W) {
    return C == 3 && (long long)warp_tiles(H, W, kBwdRows) * N < 2ll * ds::cu_count();
}

// where the staging of the image gradient lies in a workspace that starts with the pose-gradient partials
struct ImgLayout {
    size_t off_a, off_b, off_cb, off_seg, off_pt, total;
    int nwy, pts_per_tile;
};
inline ImgLayout img_layout(int N, int H, int W) {
    ImgLayout L;
    const int rows = bwd_small(N, 3, H, W) ? kBwdRowsSmall : kBwdRows;
    L.nwy = (H + rows - 1) / rows;
    L.pts_per_tile = 210 * rows + 198;   // per pixel row: three finished window rows without a slot (66 entries each) and 12
                                         // misfits, or 64 pixels as points; three more rows when the tile ends
    const size_t tiles = (size_t)N * L.nwy * ((W + 63) / 64);
    L.off_a = align256(ds_warp_bwd_workspace_bytes(N, H, W));
    L.off_b = align256(L.off_a + tiles * sizeof(int4));
    L.off_cb = align256(L.off_b + tiles * sizeof(int4));
    L.off_seg = align256(L.off_cb + tiles * sizeof(int4));
    L.off_pt = align256(L.off_seg + tiles * kSegRows * kSegFloats * sizeof(float));
    L.total = L.off_pt + tiles * L.pts_per_tile * sizeof(PtEntry);
    return L;
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

size_t ds_warp_bwd_image_workspace_bytes(int N, int C, int Hs, int Ws, int H, int W) {
    if (N <= 0 || C <= 0 || Hs <= 0 || Ws <= 0 || H <= 0 || W <= 0) return 0;
    if (C != 3) return ds_warp_bwd_workspace_bytes(N, H, W);   // other channel counts scatter with float atomics
    return img_layout(N, H, W).total;
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
    const bool stage = g_src && C == 3;   // the image gradient without atomics (see ImgStage)
    if (stage) {
        // a point entry packs (row, column) into 32 bits; the reducer addresses an image's staging with 32-bit byte offsets
        const ImgLayout L = img_layout(N, H, W);
        const long long tiles_per_image = (long long)L.nwy * ((W + 63) / 64);
        if (Hs > 32767 || Ws > 65534 || tiles_per_image * L.pts_per_tile * (long long)sizeof(PtEntry) >= (1ll << 32) ||
            tiles_per_image * kSegRows * kSegFloats * 4ll >= (1ll << 32))
            return DS_ERR_UNSUPPORTED;
    }
    if (stage && (((uintptr_t)workspace & 15) || workspace_bytes < ds_warp_bwd_image_workspace_bytes(N, C, Hs, Ws, H, W)))
        return DS_ERR_WORKSPACE;
    const WarpGeom G = warp_geom(Hs, Ws, H, W);
    const bool small = bwd_small(N, C, H, W);
    const int nblk = warp_tiles(H, W, small ? kBwdRowsSmall : kBwdRows);
    dim3 g(nblk, N), b(kThreads);
    float* partials = (float*)workspace;
    ImgStage st{};
    if (stage) {
        const ImgLayout L = img_layout(N, H, W);
        char* w = (char*)workspace;
        st.hdrA = (int4*)(w + L.off_a);
        st.hdrB = (int4*)(w + L.off_b);
        st.hdrC = (int4*)(w + L.off_cb);
        st.seg = (float*)(w + L.off_seg);
        st.pt = (PtEntry*)(w + L.off_pt);
        st.nwy = L.nwy;
        st.pts_per_tile = L.pts_per_tile;
        if (small)
            hipLaunchKernelGGL((warp_bwd_kernel<3, true, kBwdRowsSmall>), g, b, 0, stream, K, Kinv, pose, depth, src, g_warped, C,
                               G, g_depth, g_src, partials, st);
        else
            hipLaunchKernelGGL((warp_bwd_kernel<3, true>), g, b, 0, stream, K, Kinv, pose, depth, src, g_warped, C, G, g_depth,
                               g_src, partials, st);
        DS_CHECK_LAUNCH();
        hipLaunchKernelGGL(img_reduce_kernel, dim3((Ws + kThreads - 1) / kThreads, (unsigned)(N * ((Hs + kRedRows - 1) / kRedRows))),
                           dim3(kThreads), 0, stream, st, G.nbx, Hs, Ws, g_src);
    } else if (C == 3) {
        if (small)
            hipLaunchKernelGGL((warp_bwd_kernel<3, false, kBwdRowsSmall>), g, b, 0, stream, K, Kinv, pose, depth, src, g_warped,
                               C, G, g_depth, nullptr, partials, st);
        else
            hipLaunchKernelGGL(warp_bwd_kernel<3>, g, b, 0, stream, K, Kinv, pose, depth, src, g_warped, C, G, g_depth, nullptr,
                               partials, st);
    } else {
        hipLaunchKernelGGL(warp_bwd_kernel<0>, g, b, 0, stream, K, Kinv, pose, depth, src, g_warped, C, G, g_depth, g_src,
                           partials, st);
    }
    DS_CHECK_LAUNCH();
    *nblk_out = nblk;
    return DS_OK;
}
