// ds_photo.hip -- SSIM + L1 photometric error, per-pixel min over sources, soft auto-mask ratio,
// forward and backward.
//
// Reference: metrics.reprojection_error (metrics.py:21-51), metrics.reprojection_loss
// (metrics.py:54-72), loss assembly train.py:80-92, and their autograd.
//
// Kernel shape: a wave owns a band of 64 adjacent columns (one column per lane) of ONE colour channel and walks down a
// chunk of rows.  Every plane keeps two rolling partial sums per lane (roll()); the left/right neighbours of the 3x3
// window come from the adjacent lanes through a per-wave LDS row, so the VALU sees only plain adds and every global
// access is a 256-B coalesced row segment.  Halo columns/rows are re-loaded reflect-indexed, which is exactly
// functional.pad(mode="reflect").  The forward's three channel waves meet in a fourth wave through an LDS slab
// (DESIGN.md 4.1); the backward's channel waves are independent.
//
// The 9-tap sums are sequential, row-major, separately rounded, then divided by 9 with a correctly
// rounded divide: bit-identical to torch's avg_pool2d.  VALU-heavy by necessity:
// 5 window sums x 9 taps per (pixel, channel, image pair) cannot be shared between windows without
// changing the rounding order.
#ifndef DS_EXP
#define DS_EXP 0   // experiments (scripts/build_exp.sh), forward kernel, bit mask: 1 = synthetic row values instead of
                   // global loads, 2 = no workgroup barriers, 8 = the loss workgroups of warp_idt_kernel do not warp (wrong
                   // results; timing only), 16 = non-temporal hint on the once-used traffic of warp_idt_kernel (target and
                   // depth loads, warped stores), 32 = on the loads only
#endif
#ifndef DS_ROUNDS
#define DS_ROUNDS 5   // launches with many more workgroups than fit at once: rounds of resident workgroups aimed for (pick_chunks)
#endif
#include "ds_common.h"
#include "ds_warp_dev.h"
#include "ds_internal.h"
#include "depthsynth.h"
#include <math.h>

using namespace ds;

namespace {

constexpr int kWaves = 4;  // wavefronts per block (independent of each other)

struct BandGeom {
    int b, y0, yend, col, cl, lane;
    bool active;
};

// wave -> (image b, row chunk, column band).  USEFUL = 64 - 2*halo columns per band.
template <int HALO>
__device__ __forceinline__ BandGeom band_geom(int B, int H, int W, int nbx, int nby, int RB) {
    BandGeom g;
    // readfirstlane: the wave index is uniform, but only this tells the compiler -- everything derived from it
    // (image, row chunk, row offsets, loop bounds) then lives in SGPRs and the row loop branches on SCC
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    g.lane = threadIdx.x & 63;
    const int wg = xcd_remap(blockIdx.x, gridDim.x) * kWaves + wave;
    g.active = wg < B * nby * nbx;
    const int bx = wg % nbx, t = wg / nbx, by = t % nby;
    g.b = t / nby;
    g.y0 = by * RB;
    g.yend = min(g.y0 + RB, H);
    g.col = bx * (64 - 2 * HALO) - HALO + g.lane;
    g.cl = reflect_idx(g.col, W);
    return g;
}

// =====================================================================================================
// forward: G pred groups of 3 channels each against one target.  Groups 0..S-1 are the warped images
// (min -> out, argmin/pass bits -> aux); with IDT, groups S..2S-1 are the un-warped sources (min -> idt).
// =====================================================================================================
struct PhotoFwdParams {
    const float* preds;     // (B, S_total, 3, H, W); already offset to the first source of this launch
    const float* src;       // IDT only: (B, S, 3, H, W)
    const float* target;
    long long pred_stride_b, tgt_stride_b;
    int B, H, W, nbx, nby, RB;
    int s_offset;           // index of this launch's first source (chunked S / per-source targets)
    float* out;             // (B,H,W)
    float* idt;             // (B,H,W), IDT only
    uint8_t* aux;           // (B,H,W) or NULL
    float* wave_partials;   // (nwaves, 2), IDT only
};

// Rolling 3x3 window sum of one plane, one lane = one column.  Row r arrives as (L, C, R) = the values at columns
// x-1, x, x+1; the call returns the finished sum of the window centred on row r-1 and advances the two partial sums:
//   T = (L+C)+R                      of the newest row            (first row of the window centred on r+1)
//   M = ((T' +L)+C)+R                T' = T of the row before     (first two rows of the window centred on r)
//   w = ((M' +L)+C)+R                M' = M of the row before     (all three rows of the window centred on r-1)
// which is the sequential row-major order of torch's avg_pool2d (the leading 0+ is exact): 8 plain adds per plane and
// pixel, no cross-lane operand.
struct Roll { float T, M; };
__device__ __forceinline__ float roll(Roll& s, float L, float C, float R) {
    const float w = ((s.M + L) + C) + R;
    s.M = ((s.T + L) + C) + R;
    s.T = (L + C) + R;
    return w;
}

struct WarpIdtParams {
    const float* K; const float* Kinv; const float* pose; const float* depth; const float* src; float* warped;
    ds::WarpGeom G;
    int tiles_per_image;   // warp tiles per image
    int n_warp, n_loss;    // workgroups of each kind
    int groups_warp, groups_total;
    int warp_first;        // 1: all warp groups, then all loss groups; 0: interleaved in proportion
};

constexpr int kLdsRow = 68;  // 64 columns + left/right halo column (+2 pad)

// One workgroup = four waves on one band of 64 adjacent columns (all of them useful), walking down a row chunk
// together: three "channel waves", one per colour channel, and one "E wave".
//
// A channel wave holds the NP = 1 + G planes of its channel (target, then the G pred groups).  For every row the
// lane's own values (and, in lanes 0 / 63, the halo columns left / right of the band) go through a per-wave LDS row
// from which each lane reads its left and right neighbours: the stencil's cross-lane traffic rides on the LDS pipe
// and the VALU sees only plain adds (a DPP operand halves the issue rate of the add that carries it).  LDS
// instructions of one wave execute in order and no other wave touches a wave's exchange rows: no barrier for that.
// Splitting the channels over waves keeps a wave near 110 VGPRs, i.e. four workgroups per CU: one gfx950 wave issues
// a VALU instruction every ~4.4 cycles at best and a SIMD can issue one every 2 (scripts/ubench4/5), so a VALU-bound
// kernel wants >= 3 runnable waves per SIMD.  A workgroup of four waves puts one wave on each SIMD of the CU (three-
// wave workgroups left the fourth SIMD idle: -25 %).
//
// The per-channel errors e_c meet in a double-buffered LDS slab of kK rows: a channel wave writes its e_c there and
// joins one s_barrier per kK rows; the E wave forms ((e0+e1)+e2)/3, the min over sources, the identity min, the
// ratio and the sums for the slab the channel waves finished before, while they fill the other one.  The E wave is
// light (~80 of ~1070 VALU instructions per row); the role of wave w rotates with the workgroup index so that the
// light waves spread evenly over the four SIMDs.
constexpr int kFwdWaves = 4;

// warp_idt_kernel's traffic that no other workgroup reads again (see DS_EXP 16 / 32)
__device__ __forceinline__ float ld_once(rsrc_t r, unsigned lane_bytes, unsigned uniform_elems) {
    return (DS_EXP & 48) ? buf_load_f32_nt(r, lane_bytes, uniform_elems) : buf_load_f32(r, lane_bytes, uniform_elems);
}
__device__ __forceinline__ void st_once(float v, rsrc_t r, unsigned lane_bytes, unsigned uniform_bytes) {
    if (DS_EXP & 16) buf_store_f32_nt(v, r, lane_bytes, uniform_bytes);
    else buf_store_f32(v, r, lane_bytes, uniform_bytes);
}

// bid / nblk: this workgroup's index in the list of the launch's loss workgroups (the block index in a plain launch)
// EW: the E wave -- idle nine tenths of the time -- also runs ds_warp_fwd for the band's rows (Q: the warp's arguments),
// software-pipelined around the slab barriers so that no gather is waited for: a unit = 2 rows x S sources is projected
// and its 6 gathers per (row, source) issued, the previous unit is blended and stored while they fly.
// IDM: the identity map is given (P.idt, read): the E wave forms the ratio and the sums from it as with IDT.
template <int S, bool IDT, bool EW = false, bool IDM = false>
__device__ __forceinline__ void photo_fwd_body(const PhotoFwdParams& P, int bid, int nblk, const WarpIdtParams* Q = nullptr) {
    constexpr int G = IDT ? 2 * S : S;
    constexpr int NP = 1 + G;
    constexpr int kK = G <= 4 ? 4 : 2;  // rows per barrier (slab = 2 * kK * 3 * G * 256 B of LDS)
    __shared__ float lds_x[3][NP][kLdsRow];
    __shared__ float lds_e[2][kK][3][G][64];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int role = (wave + (bid >> 3)) & 3;  // 0..2 = channel, 3 = E wave
    const int lane = threadIdx.x & 63;
    const int wg = xcd_remap(bid, nblk);
    const int bx = wg % P.nbx, t = wg / P.nbx, by = t % P.nby, b = t / P.nby;
    const int H = P.H, W = P.W;
    const int y0 = by * P.RB, yend = min(y0 + P.RB, H);
    const int c0 = bx * 64, col = c0 + lane;
    const size_t HW = (size_t)H * W;

    if (role == 3) {
        // ---------------------------------------------------------------------------------- E wave
        const bool lane_out = col < W;
        float sum_ratio = 0.f, sum_out = 0.f;
        const int nrows = yend - y0;
        // EW: the warp of this band's rows (ds_warp_fwd's arithmetic: project_pixel_lean, make_tap, corner pairs)
        constexpr int UR = 2;                 // rows per unit; kK = 2 units per slab
        constexpr int SW = EW ? (S >= 2 ? S - 1 : 1) : 1;   // S >= 2: the last source is warped by the channel waves (below)
        constexpr bool EWX = EW && !(DS_EXP & 8);
        static_assert(!EW || kK == 2 * UR, "two warp units per slab");
        Cam cam[SW];
        float2_t top[UR][SW][3], bot[UR][SW][3];
        float wgt[UR][SW][4], dnx[UR][SW];
        const int xc = min(col, W - 1);
        const unsigned xb = (unsigned)xc * 4u, uHW = (unsigned)HW;
        const int nunits = (nrows + UR - 1) / UR;
        auto load_depth = [&](int u) {
#pragma unroll
            for (int i = 0; i < UR; ++i) {
                const int row = min(y0 + u * UR + i, H - 1);
#pragma unroll
                for (int s = 0; s < SW; ++s)
                    dnx[i][s] = ld_once(make_rsrc(Q->depth + (size_t)(b * S + s) * HW), xb, (unsigned)(row * W));
            }
        };
        auto issue_unit = [&](int u) {
#pragma unroll
            for (int i = 0; i < UR; ++i) {
                const int row = min(y0 + u * UR + i, H - 1);
#pragma unroll
                for (int s = 0; s < SW; ++s) {
                    Proj pr;
                    float rz;
                    Tap tp;
                    project_pixel_lean(cam[s], (float)xc, (float)row, dnx[i][s], pr, rz);
                    make_tap(pr.u, pr.v, Q->G, tp);
                    const rsrc_t sp = make_rsrc(Q->src + (size_t)(b * S + s) * 3 * HW);
#pragma unroll
                    for (int c3 = 0; c3 < 3; ++c3) {
                        top[i][s][c3] = buf_load_f32x2(sp, tp.o_top, c3 * uHW * 4u);
                        bot[i][s][c3] = buf_load_f32x2(sp, tp.o_bot, c3 * uHW * 4u);
                    }
                    wgt[i][s][0] = tp.wa_t; wgt[i][s][1] = tp.wb_t; wgt[i][s][2] = tp.wa_b; wgt[i][s][3] = tp.wb_b;
                }
            }
        };
        auto finish_unit = [&](int u) {
#pragma unroll
            for (int i = 0; i < UR; ++i) {
                const int row = y0 + u * UR + i;
                if (row >= yend) continue;   // uniform
#pragma unroll
                for (int s = 0; s < SW; ++s) {
                    const rsrc_t op = make_rsrc(Q->warped + (size_t)(b * S + s) * 3 * HW);
#pragma unroll
                    for (int c3 = 0; c3 < 3; ++c3) {
                        float acc = top[i][s][c3].x * wgt[i][s][0];
                        acc = fmaf(top[i][s][c3].y, wgt[i][s][1], acc);
                        acc = fmaf(bot[i][s][c3].x, wgt[i][s][2], acc);
                        acc = fmaf(bot[i][s][c3].y, wgt[i][s][3], acc);
                        if (lane_out) st_once(acc, op, (unsigned)col * 4u, (c3 * uHW + (unsigned)(row * W)) * 4u);
                    }
                }
            }
        };
        if (EWX) {
#pragma unroll
            for (int s = 0; s < SW; ++s) load_cam(cam[s], Q->K, Q->Kinv, Q->pose + 12 * (b * S + s));
            load_depth(0);
            issue_unit(0);
            if (nunits > 1) load_depth(1);
        }
        int u = 0;
        for (int k0 = 0; k0 < nrows; k0 += kK) {
            float idv[kK];   // IDM: the slab's identity values, requested before the wait
#pragma unroll
            for (int k = 0; k < kK; ++k)
                idv[k] = IDM ? P.idt[(size_t)b * HW + (size_t)min(y0 + k0 + k, H - 1) * W + min(col, W - 1)] : 0.f;
#if !(DS_EXP & 2)
            __syncthreads();  // the channel waves have filled slab (k0 / kK) & 1
#endif
            if (EWX) {
                finish_unit(u);
                if (u + 1 < nunits) {
                    issue_unit(u + 1);
                    if (u + 2 < nunits) load_depth(u + 2);
                }
            }
            const int buf = (k0 / kK) & 1;
            const int kn = min(kK, nrows - k0);
#pragma unroll
            for (int k = 0; k < kK; ++k) {
                if (k >= kn) break;   // uniform
                float E[G];
#pragma unroll
                for (int g = 0; g < G; ++g)
                    E[g] = div3((lds_e[buf][k][0][g][lane] + lds_e[buf][k][1][g][lane]) + lds_e[buf][k][2][g][lane]);
                const int y = y0 + k0 + k;
                float best = E[0];
                int bi = 0;
#pragma unroll
                for (int s = 1; s < S; ++s)
                    if (E[s] < best) { best = E[s]; bi = s; }
                const size_t o = (size_t)b * HW + (size_t)y * W + col;
                if (lane_out) {
                    const unsigned a = (unsigned)(bi + P.s_offset);
                    bool keep = true;
                    if (P.s_offset > 0) keep = best < P.out[o];  // later chunk: strictly smaller wins (first index on ties)
                    if (keep) {
                        P.out[o] = best;
                        if (P.aux) P.aux[o] = (uint8_t)a;
                    }
                    if (IDT) {
                        float bid = E[S];
#pragma unroll
                        for (int s = 1; s < S; ++s) bid = fminf(bid, E[S + s]);
                        P.idt[o] = bid;
                        sum_ratio += best / expf(bid + 1e-6f);  // train.py:88
                        sum_out += best;                        // train.py:92
                    }
                    if (IDM) {
                        sum_ratio += best / expf(idv[k] + 1e-6f);
                        sum_out += best;
                    }
                }
            }
            if (EWX) {
                if (u + 1 < nunits) finish_unit(u + 1);
                if (u + 2 < nunits) {
                    issue_unit(u + 2);
                    if (u + 3 < nunits) load_depth(u + 3);
                }
                u += 2;
            }
        }
        if (IDT || IDM) {
            sum_ratio = wave_sum(sum_ratio);
            sum_out = wave_sum(sum_out);
            if (lane == 0) {
                P.wave_partials[2 * wg + 0] = sum_ratio;
                P.wave_partials[2 * wg + 1] = sum_out;
            }
        }
        return;
    }

    // ------------------------------------------------------------------------------------ channel wave
    const int c = role;
    const unsigned uHW = (unsigned)HW;
    // per-lane byte offsets inside a row: own column, and the band's halo column (left for the lower half of the
    // wave, right for the upper half; only lanes 0 and 63 store theirs)
    const unsigned cl = (unsigned)reflect_idx(col, W) * 4u;
    const unsigned ch = (unsigned)reflect_idx(lane < 32 ? c0 - 1 : c0 + 64, W) * 4u;
    const bool edge = lane == 0 || lane == 63;
    const int pad_slot = lane == 0 ? 0 : 65;
    const rsrc_t tp = make_rsrc(P.target + (size_t)b * P.tgt_stride_b);
    const rsrc_t pp = make_rsrc(P.preds + (size_t)b * P.pred_stride_b);
    const rsrc_t sp = make_rsrc(IDT ? P.src + (size_t)b * (S * 3 * HW) : P.preds);
    float (*lds)[kLdsRow] = lds_x[c];

    // global loads of row r of this wave's channel: plane 0 = target, 1..S = preds, S+1..2S = un-warped sources
    auto issue = [&](int r, float (&v)[NP], float (&h)[NP]) {
        const unsigned ro = (unsigned)(reflect_idx(r, H) * W);
#if DS_EXP & 1
        const float base = 0.3f + 0.001f * (float)(r & 7) + 0.002f * (float)lane;
#pragma unroll
        for (int p = 0; p < NP; ++p) { v[p] = base * (1.0f + 0.1f * p); h[p] = v[p]; }
        (void)ro;
        return;
#endif
        v[0] = EW ? ld_once(tp, cl, c * uHW + ro) : buf_load_f32(tp, cl, c * uHW + ro);
        h[0] = EW ? ld_once(tp, ch, c * uHW + ro) : buf_load_f32(tp, ch, c * uHW + ro);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            v[1 + s] = buf_load_f32(pp, cl, (s * 3 + c) * uHW + ro);
            h[1 + s] = buf_load_f32(pp, ch, (s * 3 + c) * uHW + ro);
            if (IDT) {
                v[1 + S + s] = buf_load_f32(sp, cl, (s * 3 + c) * uHW + ro);
                h[1 + S + s] = buf_load_f32(sp, ch, (s * 3 + c) * uHW + ro);
            }
        }
    };
    // neighbour exchange of one row.  The wavefront-scope fences emit no instruction; they keep the compiler from
    // moving this thread's LDS reads above the writes of its neighbour lanes (different addresses, same wave).
    auto exchange = [&](const float (&v)[NP], const float (&h)[NP], float (&L)[NP], float (&R)[NP]) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < NP; ++p) lds[p][lane + 1] = v[p];
        if (edge) {
#pragma unroll
            for (int p = 0; p < NP; ++p) lds[p][pad_slot] = h[p];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            L[p] = lds[p][lane];
            R[p] = lds[p][lane + 2];
        }
    };

    Roll Sy{0.f, 0.f}, Syy{0.f, 0.f}, Sx[G], Sxx[G], Sxy[G];
    float Yp = 0.f, Xp[G];  // centre values of the previous row (the L1 term of the window being finished)
#pragma unroll
    for (int g = 0; g < G; ++g) {
        Sx[g] = Sxx[g] = Sxy[g] = Roll{0.f, 0.f};
        Xp[g] = 0.f;
    }

    // EW, S >= 2: the E wave warps sources 0..S-2; the last source's rows are shared out over the three channel waves (row
    // k of the chunk to wave k % 3) -- with all S sources the E wave had more instructions per slab than a channel wave
    // and set the workgroup's pace.  One (row, source) chain per wave in flight for three row steps: project, tap, 6
    // gathers at step k; blend and store at step k + 3.
    constexpr bool EWC = EW && S >= 2 && !(DS_EXP & 8);
    constexpr int s1 = S - 1;
    Cam camw;
    float2_t wtop[3], wbot[3];
    float wwg[4], wdn = 1.0f;
    int wpend = -1, kmod = 0;
    const int xcw = min(col, W - 1);
    if (EWC) {
        load_cam(camw, Q->K, Q->Kinv, Q->pose + 12 * (b * S + s1));
        if (y0 + c < yend) wdn = ld_once(make_rsrc(Q->depth + (size_t)(b * S + s1) * HW), (unsigned)xcw * 4u, (unsigned)((y0 + c) * W));
    }
    auto warp_finish = [&]() {
        const rsrc_t op = make_rsrc(Q->warped + (size_t)(b * S + s1) * 3 * HW);
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
            float acc = wtop[c3].x * wwg[0];
            acc = fmaf(wtop[c3].y, wwg[1], acc);
            acc = fmaf(wbot[c3].x, wwg[2], acc);
            acc = fmaf(wbot[c3].y, wwg[3], acc);
            if (col < W) st_once(acc, op, (unsigned)col * 4u, (c3 * uHW + (unsigned)(wpend * W)) * 4u);
        }
    };
    auto warp_issue = [&](int row) {
        Proj pr;
        float rz;
        Tap tpw;
        project_pixel_lean(camw, (float)xcw, (float)row, wdn, pr, rz);
        make_tap(pr.u, pr.v, Q->G, tpw);
        const rsrc_t spw = make_rsrc(Q->src + (size_t)(b * S + s1) * 3 * HW);
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
            wtop[c3] = buf_load_f32x2(spw, tpw.o_top, c3 * uHW * 4u);
            wbot[c3] = buf_load_f32x2(spw, tpw.o_bot, c3 * uHW * 4u);
        }
        wwg[0] = tpw.wa_t; wwg[1] = tpw.wb_t; wwg[2] = tpw.wa_b; wwg[3] = tpw.wb_b;
        wpend = row;
        if (row + 3 < yend)
            wdn = ld_once(make_rsrc(Q->depth + (size_t)(b * S + s1) * HW), (unsigned)xcw * 4u, (unsigned)((row + 3) * W));
    };

    float pend[NP], pendh[NP], cL[NP], cC[NP], cR[NP];
    issue(y0 - 1, pend, pendh);
    exchange(pend, pendh, cL, cR);
#pragma unroll
    for (int p = 0; p < NP; ++p) cC[p] = pend[p];
    issue(y0, pend, pendh);

    // rows y0-1 and y0 only feed the partial sums; from row y0+1 on, row r finishes the windows centred on r-1
    for (int r = y0 - 1; r <= yend; ++r) {
        float nL[NP], nC[NP], nR[NP];
        exchange(pend, pendh, nL, nR);  // row r+1
#pragma unroll
        for (int p = 0; p < NP; ++p) nC[p] = pend[p];
        issue(r + 2, pend, pendh);      // two rows ahead (clamped past the chunk, then unused)

        const float yL = cL[0], yC = cC[0], yR = cR[0];
        const float ay = roll(Sy, yL, yC, yR);
        const float ayy = roll(Syy, yL * yL, yC * yC, yR * yR);
        float ax[G], axx[G], axy[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float xL = cL[1 + g], xC = cC[1 + g], xR = cR[1 + g];
            ax[g] = roll(Sx[g], xL, xC, xR);
            axx[g] = roll(Sxx[g], xL * xL, xC * xC, xR * xR);
            axy[g] = roll(Sxy[g], xL * yL, xC * yC, xR * yR);
        }
        if (r > y0) {
            const int k = r - y0 - 1;  // index of the finished output row inside the chunk
            if (EWC) {
                if (kmod == c) {   // uniform
                    if (wpend >= 0) warp_finish();
                    warp_issue(y0 + k);
                }
                kmod = kmod == 2 ? 0 : kmod + 1;
            }
            TgtStats ts;
            target_stats(ay, ayy, ts);
            float (*slab)[G][64] = lds_e[(k / kK) & 1][k % kK];
#pragma unroll
            for (int g = 0; g < G; ++g) {
                SsimOut so;
                ssim_from_sums(ax[g], axx[g], axy[g], ts, so);
                slab[c][g][lane] = 0.85f * clamp01(so.v) + 0.15f * fabsf(Yp - Xp[g]);
            }
#if !(DS_EXP & 2)
            if (k % kK == kK - 1 || r == yend) __syncthreads();  // slab full: hand it to the E wave
#endif
        }
        Yp = yC;
#pragma unroll
        for (int g = 0; g < G; ++g) Xp[g] = cC[1 + g];
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            cL[p] = nL[p];
            cC[p] = nC[p];
            cR[p] = nR[p];
        }
    }
    if (EWC && wpend >= 0) warp_finish();
}

template <int S, bool IDT>
__global__ __launch_bounds__(64 * kFwdWaves, (IDT ? 2 * S : S) > 4 ? 3 : 4) void photo_fwd_kernel(PhotoFwdParams P) {
    photo_fwd_body<S, IDT>(P, (int)blockIdx.x, (int)gridDim.x);
}

// the warped groups only, identity map given (ds_loss_step_fwd after warp_idt_kernel)
template <int S>
__global__ __launch_bounds__(64 * kFwdWaves, 4) void photo_fwd_idm_kernel(PhotoFwdParams P) {
    photo_fwd_body<S, false, false, true>(P, (int)blockIdx.x, (int)gridDim.x);
}

// -----------------------------------------------------------------------------------------------------------------
// The warp and the identity loss (photo_fwd_body on the un-warped sources) in one launch.  Neither depends on the other --
// the identity term knows nothing of depth or pose.
//   EW (S <= 3): every workgroup is a loss workgroup and its waves warp the band's rows themselves, the gathers issued a
//   step ahead of their use (photo_fwd_body).  B=12 S=2 192x640: 35 us against 18 (warp) + 23 (identity loss) as two
//   launches; the gathers hit the cache lines the same workgroup streams (profiles/r02_kb3_warp_identity_one_launch.txt).
//   !EW (S = 4): two kinds of workgroups, the warp's tiles (ds_warp_fwd) and the loss workgroups, in groups of eight (one
//   per XCD, so that a workgroup's XCD is the one its index inside its own list implies).  Two launches = 100 %: B=12
//   192x640 interleaved in proportion 102 %, warp groups first 97 %, loss groups first 99 %; B=64 320x1024 97.4 % /
//   100.2 % / 98.1 % -- sharing the CUs gains little beyond the missing kernel boundary.  Warp first for small launches,
//   interleaved for large ones.

template <int S, bool EW>
__global__ __launch_bounds__(64 * kFwdWaves, 4) void warp_idt_kernel(WarpIdtParams Q, PhotoFwdParams P) {
    if (EW) {   // every workgroup is a loss workgroup whose waves warp the band's rows
        photo_fwd_body<S, false, true>(P, (int)blockIdx.x, (int)gridDim.x, &Q);
        return;
    }
    const int grp = (int)(blockIdx.x >> 3), x = (int)(blockIdx.x & 7);
    // number of warp groups among the first g groups: floor(g * groups_warp / groups_total)
    int before, upto;
    if (Q.warp_first) {
        before = grp < Q.groups_warp ? grp : Q.groups_warp;
        upto = grp < Q.groups_warp ? grp + 1 : Q.groups_warp;
    } else {
        before = (int)(((long long)grp * Q.groups_warp) / Q.groups_total);
        upto = (int)(((long long)(grp + 1) * Q.groups_warp) / Q.groups_total);
    }
    if (upto > before) {
        const int t = before * 8 + x;
        if (t >= Q.n_warp) return;
        const int n = t / Q.tiles_per_image, tile = t - n * Q.tiles_per_image;
        warp_fwd_tile<3>(Q.K, Q.Kinv, Q.pose, Q.depth, Q.src, 3, Q.G, Q.warped, n, tile);
    } else {
        const int bid = (grp - before) * 8 + x;
        if (bid >= Q.n_loss) return;   // whole workgroup: no barrier is left waiting
        photo_fwd_body<S, false>(P, bid, Q.n_loss);
    }
}

// per-image sums of the wave partials, fixed order, double accumulation -> means (B,2)
__global__ void loss_finalize_kernel(const float* __restrict__ wave_partials, int waves_per_image, double inv_hw,
                                     float* __restrict__ sums) {
    loss_finalize_wave(wave_partials, blockIdx.x, waves_per_image, inv_hw, sums, threadIdx.x);
}

// =====================================================================================================
// plain SSIM map of independent planes (metrics.reprojection_error), same rolling scheme
// =====================================================================================================
__global__ __launch_bounds__(64 * kWaves) void ssim_fwd_kernel(const float* __restrict__ preds,
                                                               const float* __restrict__ targets, int nplanes, int H,
                                                               int W, int nbx, int nby, int RB, float* __restrict__ out) {
    const BandGeom g = band_geom<1>(nplanes, H, W, nbx, nby, RB);
    if (!g.active) return;
    const unsigned cl = (unsigned)g.cl * 4u;  // lane's column byte offset: the only per-lane part of a load address
    const size_t HW = (size_t)H * W;
    const float* xp = preds + (size_t)g.b * HW;
    const float* yp = targets + (size_t)g.b * HW;
    const bool lane_out = g.lane >= 1 && g.lane <= 62 && g.col < W;
    float X[3], Y[3];
    auto place = [&](int slot, int r) {  // slot is a compile-time constant at every call site after unrolling
        const size_t ro = (size_t)reflect_idx(r, H) * W;
        X[slot] = (xp + ro)[cl >> 2];
        Y[slot] = (yp + ro)[cl >> 2];
    };
    auto emit = [&](int y, int T, int M, int Bt) {
        float ax, axx, axy, ay, ayy;
        win_target(Y[T], Y[M], Y[Bt], ay, ayy);
        win_pair(X[T], X[M], X[Bt], Y[T], Y[M], Y[Bt], ax, axx, axy);
        TgtStats ts;
        target_stats(ay, ayy, ts);
        SsimOut so;
        ssim_from_sums(ax, axx, axy, ts, so);
        if (lane_out) out[(size_t)g.b * HW + (size_t)y * W + g.col] = clamp01(so.v);
    };
    place(0, g.y0 - 1);
    place(1, g.y0);
    for (int y = g.y0; y < g.yend; y += 3) {
        place(2, y + 1);
        emit(y, 0, 1, 2);
        if (y + 1 >= g.yend) break;
        place(0, y + 2);
        emit(y + 1, 1, 2, 0);
        if (y + 2 >= g.yend) break;
        place(1, y + 3);
        emit(y + 2, 2, 0, 1);
    }
}

// =====================================================================================================
// backward.  Window (q) -> tap coefficients: d loss / d x_tap = (A_q + Bq * x_tap + Cq * y_tap), already
// divided by 9; a pixel sums them over the windows that contain it (reflect-pad adjoint = multiplicity 2
// for the border windows seen from the second row/column).
// =====================================================================================================
struct Coef { float A, Bc, Cc; };

// g_v = d loss / d v of this window & channel (0 when the clamp blocked it)
__device__ __forceinline__ Coef ssim_coefs(const SsimOut& so, const TgtStats& ts, float g_v) {
    const float rdd = so.rdd;  // within 1 ulp of 1/dd: the gradient needs no more
    const float g_n = -0.5f * g_v * rdd;
    const float g_dd = -g_n * so.n * rdd;
    const float mx2 = so.mx * so.mx;
    const float a = fmaf(2.0f, so.mx * ts.my, kC1), bb = fmaf(2.0f, so.sxy, kC2);
    const float cc = (mx2 + ts.my2) + kC1, ee = (so.sx + ts.sy_raw) + kC2;
    const float g_a = g_n * bb, g_b = g_n * a, g_c = g_dd * ee, g_sx = g_dd * cc;
    const float g_sxy = 2.0f * g_b;
    // g_mx = 2 my g_a + 2 mx g_c - 2 mx g_sx - my g_sxy
    const float g_mx = fmaf(2.0f * ts.my, g_a, 2.0f * so.mx * (g_c - g_sx)) - ts.my * g_sxy;
    const float ninth = 1.0f / 9.0f;
    Coef k;
    k.A = g_mx * ninth;
    k.Bc = 2.0f * g_sx * ninth;
    k.Cc = g_sxy * ninth;
    return k;
}

struct PhotoBwdParams {
    const float* preds;     // (B, S_total, 3, H, W) offset to this launch's first source
    const float* target;
    long long pred_stride_b, tgt_stride_b;
    const float* g_out;     // !FUSED: (B,H,W) upstream gradient of the min map
    const float* idt;       // FUSED: (B,H,W)
    const float* g_sums;    // FUSED: (B) upstream gradient of mean(out/exp(idt+1e-6))
    const uint8_t* aux;     // (B,H,W)
    int B, H, W, nbx, nby, RB, s_offset;
    float inv_hw;
    float* g_preds;         // (B, S_total, 3, H, W) offset like preds
};

// One wave = one colour channel of one band of 64 adjacent columns (62 of them outputs: a pixel's gradient needs the
// windows one column to either side) walking down a row chunk (two lead-in rows that only feed the window sums, one
// more window row above and below the chunk's outputs).  The channels are independent in the backward, so the waves
// of a workgroup are four unrelated units: no barrier.  Two neighbour exchanges per row ride on the LDS pipe (a
// wave's LDS instructions execute in order; no other wave touches its rows):
//   * the raw values of the next row (as in the forward): L, C, R of every plane for the rolling window sums;
//   * the window's tap coefficients, masked per source: the horizontal part of the 3x3 spread.
// The vertical part of the spread keeps the horizontally spread coefficients of the last two window rows.
// The forward's intermediates (window sums -> SSIM -> clamp decision) are recomputed with the forward's own code and
// rounding; only the argmin source of a window gets gradient, so its sums are selected before the SSIM formula.
template <int S, bool FUSED>
__global__ __launch_bounds__(64 * kWaves, S >= 4 ? 3 : 4) void photo_bwd_kernel(PhotoBwdParams P) {
    constexpr int NP = 1 + S;  // raw planes of this channel: target, then the S preds
    constexpr int NK = 3 * S;  // masked tap coefficients: (A, B, C) per source
    __shared__ float lds_raw[kWaves][NP][kLdsRow];
    __shared__ float lds_k[kWaves][NK][kLdsRow];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int unit = xcd_remap(blockIdx.x, gridDim.x) * kWaves + wave;
    if (unit >= P.B * P.nby * P.nbx * 3) return;
    const int c = unit % 3, u1 = unit / 3, bx = u1 % P.nbx, u2 = u1 / P.nbx, by = u2 % P.nby, b = u2 / P.nby;
    const int H = P.H, W = P.W;
    const int y0 = by * P.RB, yend = min(y0 + P.RB, H);
    const int c0 = bx * 62, col = c0 - 1 + lane;  // lane l holds column c0-1+l; outputs are lanes 1..62
    const size_t HW = (size_t)H * W;
    const unsigned uHW = (unsigned)HW;
    const unsigned cl = (unsigned)reflect_idx(col, W) * 4u;
    const unsigned ch = (unsigned)reflect_idx(lane < 32 ? c0 - 2 : c0 + 63, W) * 4u;
    const bool edge = lane == 0 || lane == 63;
    const int pad_slot = lane == 0 ? 0 : 65;
    const bool col_exists = col >= 0 && col < W;
    const bool lane_out = lane >= 1 && lane <= 62 && col < W;
    const float wL = col == 1 ? 2.0f : 1.0f, wR = col == W - 2 ? 2.0f : 1.0f;  // reflect-pad adjoint, see above
    const rsrc_t tp = make_rsrc(P.target + (size_t)b * P.tgt_stride_b);
    const rsrc_t pp = make_rsrc(P.preds + (size_t)b * P.pred_stride_b);
    const rsrc_t wp = make_rsrc((FUSED ? P.idt : P.g_out) + (size_t)b * HW);
    const rsrc_t ap = make_rsrc(P.aux + (size_t)b * HW);
    const rsrc_t gp = make_rsrc(P.g_preds + (size_t)b * P.pred_stride_b);
    const float gscale = FUSED ? (P.g_sums ? P.g_sums[b] * P.inv_hw : P.inv_hw) : 1.0f;
    float (*lr)[kLdsRow] = lds_raw[wave];
    float (*lk)[kLdsRow] = lds_k[wave];

    struct Row { float v[NP], h[NP], gw; int am; };
    auto issue = [&](int r, Row& o) {
        const unsigned ro = (unsigned)(reflect_idx(r, H) * W);
        o.v[0] = buf_load_f32(tp, cl, c * uHW + ro);
        o.h[0] = buf_load_f32(tp, ch, c * uHW + ro);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            o.v[1 + s] = buf_load_f32(pp, cl, (s * 3 + c) * uHW + ro);
            o.h[1 + s] = buf_load_f32(pp, ch, (s * 3 + c) * uHW + ro);
        }
        o.gw = buf_load_f32(wp, cl, ro);
        o.am = buf_load_u8(ap, cl >> 2, ro);
    };
    auto exchange_raw = [&](const Row& o, float (&L)[NP], float (&R)[NP]) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < NP; ++p) lr[p][lane + 1] = o.v[p];
        if (edge) {
#pragma unroll
            for (int p = 0; p < NP; ++p) lr[p][pad_slot] = o.h[p];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            L[p] = lr[p][lane];
            R[p] = lr[p][lane + 2];
        }
    };

    Roll Sy{0.f, 0.f}, Syy{0.f, 0.f}, Sx[S], Sxx[S], Sxy[S];
#pragma unroll
    for (int s = 0; s < S; ++s) Sx[s] = Sxx[s] = Sxy[s] = Roll{0.f, 0.f};
    // per-lane values of the two rows before the newest one: [0] = row r-1 (window row), [1] = row r-2 (output row)
    float pv[2][NP], pgw[2] = {0.f, 0.f};
    int pam[2] = {0, 0};
    float Hs[2][NK];  // horizontally spread coefficients of window rows j-1 ([0]) and j-2 ([1])
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int p = 0; p < NP; ++p) pv[i][p] = 0.f;
#pragma unroll
        for (int k = 0; k < NK; ++k) Hs[i][k] = 0.f;
    }

    Row pend;
    float cL[NP], cC[NP], cR[NP], cgw;
    int cam;
    issue(y0 - 2, pend);
    exchange_raw(pend, cL, cR);
#pragma unroll
    for (int p = 0; p < NP; ++p) cC[p] = pend.v[p];
    cgw = pend.gw;
    cam = pend.am;
    issue(y0 - 1, pend);

    // row r arrives: window row j = r-1 completes (from r = y0 on), output row t = r-2 completes (from r = y0+2 on)
    for (int r = y0 - 2; r <= yend + 1; ++r) {
        float nL[NP], nC[NP], nR[NP];
        exchange_raw(pend, nL, nR);  // row r+1
#pragma unroll
        for (int p = 0; p < NP; ++p) nC[p] = pend.v[p];
        const float ngw_raw = pend.gw;
        const int nam = pend.am;
        issue(r + 2, pend);

        const float yL = cL[0], yC = cC[0], yR = cR[0];
        const float ay = roll(Sy, yL, yC, yR);
        const float ayy = roll(Syy, yL * yL, yC * yC, yR * yR);
        float ax[S], axx[S], axy[S];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const float xL = cL[1 + s], xC = cC[1 + s], xR = cR[1 + s];
            ax[s] = roll(Sx[s], xL, xC, xR);
            axx[s] = roll(Sxx[s], xL * xL, xC * xC, xR * xR);
            axy[s] = roll(Sxy[s], xL * yL, xC * yC, xR * yR);
        }
        // gradient weight of the newest row's pixels: d loss / d out(p)
        const float gw_new = FUSED ? gscale * __builtin_amdgcn_exp2f((cgw + 1e-6f) * -1.4426950408889634f) : cgw;

        if (r >= y0) {
            // ---- window row j = r-1: coefficients of its taps, masked per source, spread horizontally
            const int j = r - 1;
            const bool win_exists = col_exists && j >= 0 && j < H;
            const int idx = (pam[0] & 0x1f) - P.s_offset;  // argmin source of this window, relative to this launch
            TgtStats ts;
            target_stats(ay, ayy, ts);
            float bx_ = ax[0], bxx = axx[0], bxy = axy[0];
#pragma unroll
            for (int s = 1; s < S; ++s) {
                const bool is = idx == s;
                bx_ = is ? ax[s] : bx_;
                bxx = is ? axx[s] : bxx;
                bxy = is ? axy[s] : bxy;
            }
            SsimOut so;
            ssim_from_sums(bx_, bxx, bxy, ts, so);
            // clamp passes the gradient iff 0 <= v <= 1 (so.v is recomputed bit-identically to the forward)
            const bool live = win_exists && idx >= 0 && idx < S && so.v >= 0.0f && so.v <= 1.0f;
            const Coef k = ssim_coefs(so, ts, live ? pgw[0] * (0.85f / 3.0f) : 0.0f);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const bool is = idx == s;
                lk[3 * s + 0][lane + 1] = is ? k.A : 0.0f;
                lk[3 * s + 1][lane + 1] = is ? k.Bc : 0.0f;
                lk[3 * s + 2][lane + 1] = is ? k.Cc : 0.0f;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float Hn[NK];
#pragma unroll
            for (int q = 0; q < NK; ++q) {
                const float kc = (idx == q / 3) ? (q % 3 == 0 ? k.A : (q % 3 == 1 ? k.Bc : k.Cc)) : 0.0f;
                Hn[q] = fmaf(wR, lk[q][lane + 2], fmaf(wL, lk[q][lane], kc));
            }
            // ---- output row t = r-2: windows t-1 (Hs[1]), t (Hs[0]), t+1 (Hn), reflect-pad adjoint weights
            const int t = r - 2;
            if (t >= y0) {
                const float wU = t == 1 ? 2.0f : 1.0f, wD = t == H - 2 ? 2.0f : 1.0f;
                const float ge = pgw[1] * (1.0f / 3.0f);
                const float Y = pv[1][0];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float gA = fmaf(wD, Hn[3 * s + 0], fmaf(wU, Hs[1][3 * s + 0], Hs[0][3 * s + 0]));
                    const float gB = fmaf(wD, Hn[3 * s + 1], fmaf(wU, Hs[1][3 * s + 1], Hs[0][3 * s + 1]));
                    const float gC = fmaf(wD, Hn[3 * s + 2], fmaf(wU, Hs[1][3 * s + 2], Hs[0][3 * s + 2]));
                    const float X = pv[1][1 + s];
                    float gg = fmaf(Y, gC, fmaf(X, gB, gA));
                    const float diff = Y - X;
                    const float sgn = diff > 0.f ? -1.0f : (diff < 0.f ? 1.0f : 0.0f);
                    if ((pam[1] & 0x1f) == s + P.s_offset) gg = fmaf(0.15f * ge, sgn, gg);
                    if (lane_out) buf_store_f32(gg, gp, (unsigned)col * 4u, ((s * 3 + c) * uHW + (unsigned)(t * W)) * 4u);
                }
            }
#pragma unroll
            for (int q = 0; q < NK; ++q) {
                Hs[1][q] = Hs[0][q];
                Hs[0][q] = Hn[q];
            }
        }
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            pv[1][p] = pv[0][p];
            pv[0][p] = cC[p];
            cL[p] = nL[p];
            cC[p] = nC[p];
            cR[p] = nR[p];
        }
        pgw[1] = pgw[0];
        pgw[0] = gw_new;
        pam[1] = pam[0];
        pam[0] = cam;
        cgw = ngw_raw;
        cam = nam;
    }
}

// =====================================================================================================
// forward AND backward of the warped groups in one kernel (ds_fused_loss_fb), given the identity map.
//
// The backward above recomputes the forward's window sums and its SSIM (60 % of its instructions) because the forward
// kernel had finished before it started.  Here one workgroup does both for a band of 62 output columns x a row chunk:
// three channel waves (one colour channel each, all S warped groups) and one light output wave.
//   * A channel wave runs the forward of its channel exactly like photo_fwd_kernel (rolling sums, neighbours through its
//     own LDS row), leaves e_c in a double-buffered LDS slab of kK rows and keeps the SSIM intermediates of those rows
//     in registers;
//   * after the slab's barrier every channel wave reads all three channels' e_c back, forms E = ((e0+e1)+e2)/3, the min
//     and argmin over the sources itself (redundantly: ~20 instructions against a second hand-over through LDS), takes
//     d loss / d out from the identity map, and runs the backward of its channel for those rows: coefficients of the
//     argmin source, 3x3 spread (horizontally through a second LDS row, vertically over the last two rows), d loss / d
//     warped for the row above;
//   * the output wave reads the same slabs, writes out / argmin and the partial sums of the two means.
// Same arithmetic, in the same order, as photo_fwd_kernel + photo_bwd_kernel: identical results (tests).
// =====================================================================================================
struct PhotoFbParams {
    const float* preds;     // (B, S, 3, H, W) warped
    const float* target;    // (B, 3, H, W)
    const float* idt;       // (B, H, W) identity loss map (ds_reprojection_loss_fwd on the un-warped sources)
    const float* g_sums;    // (B) upstream gradient of mean(out/exp(idt+1e-6)), or NULL = 1
    int B, H, W, nbx, nby, RB;
    float inv_hw;
    float* out;             // (B,H,W)
    uint8_t* aux;           // (B,H,W)
    float* wave_partials;   // (nblocks, 2)
    float* g_preds;         // (B, S, 3, H, W); may be NULL with WARP (d loss / d warped then never reaches memory)
    // WARP: the fourth wave also runs the warp's autograd on the rows the channel waves finish (ds_warp_bwd's arithmetic)
    const float* K; const float* Kinv; const float* pose;   // (4,4), (4,4), (B*S,3,4)
    const float* depth;     // (B, S, H, W)
    const float* src;       // (B, S, 3, H, W)
    float* g_depth;         // (B, S, H, W)
    float* pose_partials;   // (B*S, nbx*nby, 12)
    ds::WarpGeom G;
};

template <int S, bool WARP>
__global__ __launch_bounds__(64 * kFwdWaves, S <= 2 ? 4 : 3) void photo_fb_kernel(PhotoFbParams P) {
    constexpr int NP = 1 + S;  // raw planes of a channel: target, then the S warped groups
    constexpr int NK = 3 * S;  // masked tap coefficients: (A, B, C) per source
    constexpr int kK = 2;      // window rows per barrier
    __shared__ float lds_raw[3][NP][kLdsRow];
    __shared__ float lds_k[3][NK][kLdsRow];
    __shared__ float lds_e[2][kK][3][S][64];
    __shared__ float lds_p[3][kK][NP + 1][64];  // per channel wave: centre values and identity loss of the pending rows
    __shared__ float lds_g[WARP ? 2 : 1][kK][3][WARP ? S : 1][WARP ? 64 : 1];  // WARP: rows of d loss / d warped for the 4th wave
    __shared__ float lds_s[2][WARP ? 64 : 1];                                   // WARP: channel wave 0's per-lane loss sums
    __shared__ float lds_w[WARP ? kK : 1][WARP ? S : 1][7][WARP ? 64 : 1];      // WARP: the 4th wave's per-chain state across the barrier
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int role = (wave + (int)(blockIdx.x >> 3)) & 3;  // 0..2 = channel, 3 = output wave
    const int lane = threadIdx.x & 63;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int bx = wg % P.nbx, tt = wg / P.nbx, by = tt % P.nby, b = tt / P.nby;
    const int H = P.H, W = P.W;
    const int y0 = by * P.RB, yend = min(y0 + P.RB, H);
    const int c0 = bx * 62, col = c0 - 1 + lane;   // lane l holds column c0-1+l; outputs are lanes 1..62
    const size_t HW = (size_t)H * W;
    const int nwin = yend - y0 + 2;                // window rows y0-1 .. yend
    const int nslab = (nwin + kK - 1) / kK;        // every wave of the workgroup joins exactly nslab barriers
    const bool lane_out = lane >= 1 && lane <= 62 && col < W;

    if (role == 3) {
        // ---------------------------------------------------------------------------------- output (and warp) wave
        const rsrc_t ip = make_rsrc(P.idt + (size_t)b * HW);
        const unsigned cl = (unsigned)reflect_idx(col, W) * 4u;
        float sum_ratio = 0.f, sum_out = 0.f;
        // WARP: the autograd of the warp for the rows of d loss / d warped the channel waves finished one barrier ago --
        // per pixel exactly ds_warp_bwd's arithmetic (project_pixel_lean, make_tap, corner pairs, grid-sample border
        // rule, chain back to depth and pose); the rows come through lds_g instead of memory.
        const int xc = min(max(col, 0), W - 1);
        const unsigned xb = (unsigned)xc * 4u;
        const unsigned HsWs = (unsigned)HW;
        Cam cam[WARP ? S : 1];
        float gM[WARP ? S : 1][12];
        if (WARP) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                load_cam(cam[s], P.K, P.Kinv, P.pose + 12 * (b * S + s));
#pragma unroll
                for (int i = 0; i < 12; ++i) gM[s][i] = 0.f;
            }
        }
        const float sxg = WARP ? (P.G.w1 * 0.5f) * (2.0f * P.G.rw1) : 0.f, syg = WARP ? (P.G.h1 * 0.5f) * (2.0f * P.G.rh1) : 0.f;
        const int niter = WARP ? nslab + 1 : nslab;
        // WARP: the gathers of a slab's rows do not depend on the channel waves -- only the rows of d loss / d warped do.  So
        // the projection, the taps and the 6 gathers of all kK x S (row, source) chains are issued BEFORE the barrier and
        // are back when the channel waves arrive; after it only the arithmetic is left.  (One chain at a time after the
        // barrier, as round 2 first had it, made this wave the workgroup's critical path: four dependent memory round
        // trips per slab.)  The depths are requested one slab earlier still.
        float dep[kK][S];
        if (WARP) {
#pragma unroll
            for (int k = 0; k < kK; ++k)
#pragma unroll
                for (int s = 0; s < S; ++s) dep[k][s] = 1.0f;
        }
        for (int sl = 0; sl < niter; ++sl) {
            // identity values of this slab's rows, requested before the wait
            float iv[kK];
#pragma unroll
            for (int k = 0; k < kK; ++k)
                iv[k] = WARP ? 0.f : buf_load_f32(ip, cl, (unsigned)(reflect_idx(y0 - 1 + sl * kK + k, H) * W));
            float2_t top_[WARP ? kK : 1][WARP ? S : 1][3], bot_[WARP ? kK : 1][WARP ? S : 1][3];
            if (WARP) {
                if (sl >= 1) {
#pragma unroll
                    for (int k = 0; k < kK; ++k) {
                        const int t = y0 - 2 + (sl - 1) * kK + k;   // output row of window row y0-1 + (sl-1)*kK + k
                        if (t < y0 || t >= yend) continue;          // uniform
#pragma unroll
                        for (int s = 0; s < S; ++s) {
                            Proj pr;
                            float rz;
                            Tap tp;
                            project_pixel_lean(cam[s], (float)xc, (float)t, dep[k][s], pr, rz);
                            make_tap(pr.u, pr.v, P.G, tp);
                            const rsrc_t sp = make_rsrc(P.src + (size_t)(b * S + s) * 3 * HW);
#pragma unroll
                            for (int ch3 = 0; ch3 < 3; ++ch3) {
                                top_[k][s][ch3] = buf_load_f32x2(sp, tp.o_top, ch3 * HsWs * 4u);
                                bot_[k][s][ch3] = buf_load_f32x2(sp, tp.o_bot, ch3 * HsWs * 4u);
                            }
                            // what the arithmetic after the barrier needs of this chain goes through LDS (own lane's slot, no
                            // synchronisation): in registers the four chains' state spills (53 VGPRs)
                            const int flags = (tp.shift ? 1 : 0) | ((tp.ixu > 0.0f && tp.ixu < P.G.w1) ? 2 : 0) |
                                              ((tp.iyu > 0.0f && tp.iyu < P.G.h1) ? 4 : 0) | ((fabsf(pr.q[2]) < 1e-6f) ? 8 : 0);
                            lds_w[k][s][0][lane] = tp.wx;
                            lds_w[k][s][1][lane] = tp.ny;
                            lds_w[k][s][2][lane] = __int_as_float(flags);
                            lds_w[k][s][3][lane] = pr.u;
                            lds_w[k][s][4][lane] = pr.v;
                            lds_w[k][s][5][lane] = rz;
                            lds_w[k][s][6][lane] = dep[k][s];
                        }
                    }
                }
                // depths of the rows the NEXT iteration differentiates
#pragma unroll
                for (int k = 0; k < kK; ++k) {
                    const int t = min(max(y0 - 2 + sl * kK + k, 0), H - 1);
#pragma unroll
                    for (int s = 0; s < S; ++s)
                        dep[k][s] = buf_load_f32(make_rsrc(P.depth + (size_t)(b * S + s) * HW), xb, (unsigned)(t * W));
                }
            }
            __syncthreads();  // the channel waves have filled slab sl & 1 (and the rows of d loss / d warped before it)
            const int buf = sl & 1;
            if (!WARP && sl < nslab) {   // WARP: channel wave 0 writes out / argmin and keeps the sums
#pragma unroll
                for (int k = 0; k < kK; ++k) {
                    const int y = y0 - 1 + sl * kK + k;
                    float E[S];
#pragma unroll
                    for (int g = 0; g < S; ++g)
                        E[g] = div3((lds_e[buf][k][0][g][lane] + lds_e[buf][k][1][g][lane]) + lds_e[buf][k][2][g][lane]);
                    float best = E[0];
                    int bi = 0;
#pragma unroll
                    for (int g = 1; g < S; ++g)
                        if (E[g] < best) { best = E[g]; bi = g; }
                    if (lane_out && y >= y0 && y < yend) {
                        const size_t o = (size_t)b * HW + (size_t)y * W + col;
                        P.out[o] = best;
                        P.aux[o] = (uint8_t)bi;
                        sum_ratio += best / expf(iv[k] + 1e-6f);  // train.py:88
                        sum_out += best;                          // train.py:92
                    }
                }
            }
            if (WARP && sl >= 1) {
                const int gbuf = (sl - 1) & 1;
                float fxl = (float)xc;
                asm volatile("" : "+v"(fxl));   // a value of its own: the ray is formed again, not kept across the barrier
#pragma unroll
                for (int k = 0; k < kK; ++k) {
                    const int t = y0 - 2 + (sl - 1) * kK + k;
                    if (t < y0 || t >= yend) continue;          // uniform
#pragma unroll
                    for (int s = 0; s < S; ++s) {
                        const float wx = lds_w[k][s][0][lane], ny = lds_w[k][s][1][lane];
                        const int flags = __float_as_int(lds_w[k][s][2][lane]);
                        const float pu = lds_w[k][s][3][lane], pv = lds_w[k][s][4][lane], rz = lds_w[k][s][5][lane];
                        const float d = lds_w[k][s][6][lane];
                        const float ex = 1.0f - wx, sy = 1.0f - ny;   // as make_tap forms them
                        const bool shift = flags & 1;
                        float pc[3], pP[3];                            // as project_pixel_lean forms them
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            float acc = fxl * cam[s].Kinv[3 * j + 0];
                            acc = fmaf((float)t, cam[s].Kinv[3 * j + 1], acc);
                            acc = fmaf(1.0f, cam[s].Kinv[3 * j + 2], acc);
                            pc[j] = acc;
                            pP[j] = acc * d;
                        }
                        const float2_t* top = top_[k][s];
                        const float2_t* bot = bot_[k][s];
                        float gix = 0.f, giy = 0.f;
#pragma unroll
                        for (int ch3 = 0; ch3 < 3; ++ch3) {
                            const float gc = lane_out ? lds_g[gbuf][k][ch3][s][lane] : 0.0f;
                            const float2_t a = top[ch3], bb = bot[ch3];
                            const float nw = shift ? a.y : a.x, ne = a.y, sw = shift ? bb.y : bb.x, se = bb.y;
                            gix = fmaf(gc, fmaf(ne - nw, sy, (se - sw) * ny), gix);
                            giy = fmaf(gc, fmaf(sw - nw, ex, (se - ne) * wx), giy);
                        }
                        // torch grid_sampler_2d backward: zero on or outside the border of the un-clipped coordinate
                        gix = (flags & 2) ? gix * sxg : 0.0f;
                        giy = (flags & 4) ? giy * syg : 0.0f;
                        if (!lane_out) gix = giy = 0.0f;   // halo lanes and lanes beyond the image
                        const float gq0 = gix * rz, gq1 = giy * rz;
                        const float gq2 = (flags & 8) ? 0.0f : -(fmaf(pu, gix, pv * giy)) * rz;
                        float gT[3], gP[3];
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            gT[q] = fmaf(cam[s].K[6 + q], gq2, fmaf(cam[s].K[3 + q], gq1, cam[s].K[q] * gq0));
#pragma unroll
                        for (int q = 0; q < 3; ++q)
                            gP[q] = fmaf(cam[s].M[8 + q], gT[2], fmaf(cam[s].M[4 + q], gT[1], cam[s].M[q] * gT[0]));
                        const float gd = fmaf(pc[2], gP[2], fmaf(pc[1], gP[1], pc[0] * gP[0]));
#pragma unroll
                        for (int j2 = 0; j2 < 3; ++j2) {
                            gM[s][4 * j2 + 0] = fmaf(gT[j2], pP[0], gM[s][4 * j2 + 0]);
                            gM[s][4 * j2 + 1] = fmaf(gT[j2], pP[1], gM[s][4 * j2 + 1]);
                            gM[s][4 * j2 + 2] = fmaf(gT[j2], pP[2], gM[s][4 * j2 + 2]);
                            gM[s][4 * j2 + 3] += gT[j2];
                        }
                        if (lane_out) P.g_depth[(size_t)(b * S + s) * HW + (size_t)t * W + col] = gd;
                    }
                }
            }
        }
        if (!WARP) {
            sum_ratio = wave_sum(sum_ratio);
            sum_out = wave_sum(sum_out);
            if (lane == 0) {
                P.wave_partials[2 * wg + 0] = sum_ratio;
                P.wave_partials[2 * wg + 1] = sum_out;
            }
        }
        if (WARP) {
            const int per_image = P.nbx * P.nby, slot = by * P.nbx + bx;
#pragma unroll
            for (int s = 0; s < S; ++s) {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    const float v = wave_sum(gM[s][i]);
                    if (lane == 0) P.pose_partials[((size_t)(b * S + s) * per_image + slot) * 12 + i] = v;
                }
            }
        }
        return;
    }

    // ------------------------------------------------------------------------------------ channel wave
    const int c = role;
    const unsigned uHW = (unsigned)HW;
    const unsigned cl = (unsigned)reflect_idx(col, W) * 4u;
    const unsigned ch = (unsigned)reflect_idx(lane < 32 ? c0 - 2 : c0 + 63, W) * 4u;
    const bool edge = lane == 0 || lane == 63;
    const int pad_slot = lane == 0 ? 0 : 65;
    const bool col_exists = col >= 0 && col < W;
    const float wL = col == 1 ? 2.0f : 1.0f, wR = col == W - 2 ? 2.0f : 1.0f;  // reflect-pad adjoint
    const rsrc_t tp = make_rsrc(P.target + (size_t)b * 3 * HW);
    const rsrc_t pp = make_rsrc(P.preds + (size_t)b * (S * 3 * HW));
    const rsrc_t ip = make_rsrc(P.idt + (size_t)b * HW);
    const bool store_g = P.g_preds != nullptr;   // uniform
    const rsrc_t gp = make_rsrc(store_g ? P.g_preds + (size_t)b * (S * 3 * HW) : P.out);
    const float gscale = P.g_sums ? P.g_sums[b] * P.inv_hw : P.inv_hw;
    float (*lr)[kLdsRow] = lds_raw[c];
    float (*lk)[kLdsRow] = lds_k[c];

    struct Row { float v[NP], h[NP], iv; };
    auto issue = [&](int r, Row& o) {
        const unsigned ro = (unsigned)(reflect_idx(r, H) * W);
        o.v[0] = buf_load_f32(tp, cl, c * uHW + ro);
        o.h[0] = buf_load_f32(tp, ch, c * uHW + ro);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            o.v[1 + s] = buf_load_f32(pp, cl, (s * 3 + c) * uHW + ro);
            o.h[1 + s] = buf_load_f32(pp, ch, (s * 3 + c) * uHW + ro);
        }
        o.iv = buf_load_f32(ip, cl, ro);
    };
    auto exchange_raw = [&](const Row& o, float (&L)[NP], float (&R)[NP]) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < NP; ++p) lr[p][lane + 1] = o.v[p];
        if (edge) {
#pragma unroll
            for (int p = 0; p < NP; ++p) lr[p][pad_slot] = o.h[p];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            L[p] = lr[p][lane];
            R[p] = lr[p][lane + 2];
        }
    };

    Roll Sy{0.f, 0.f}, Syy{0.f, 0.f}, Sx[S], Sxx[S], Sxy[S];
#pragma unroll
    for (int s = 0; s < S; ++s) Sx[s] = Sxx[s] = Sxy[s] = Roll{0.f, 0.f};

    // what a finished window row keeps until its slab's barrier has passed
    struct Pend {
        float ax[S], axx[S], axy[S];  // window sums of every source (the SSIM of the argmin source is formed again from
                                      // them after the barrier: 40 instructions against 2 x 7 more registers, which spill)
        TgtStats ts;
    };  // the row's centre values (L1 term, tap values of the output row) and identity loss wait in lds_p: registers are
        // short (128 with four waves per SIMD), the LDS pipe is not
    Pend pd[kK];
    // backward state carried from window row to window row
    float Hs[2][NK];     // horizontally spread coefficients of window rows j-1 ([0]) and j-2 ([1])
    float praw[NP];      // centre values, gradient weight and argmin of window row j-1 (= the output row)
    float pgw = 0.f;
    int pam = 0;
#pragma unroll
    for (int q = 0; q < NK; ++q) Hs[0][q] = Hs[1][q] = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) praw[p] = 0.f;

    Row pend;
    float cL[NP], cC[NP], cR[NP], civ;
    float pvC[NP], piv = 0.f;   // centre values / identity loss of the row before the newest one
#pragma unroll
    for (int p = 0; p < NP; ++p) pvC[p] = 0.f;

    float ay, ayy, ax[S], axx[S], axy[S];  // sums of the window row a raw row has just finished

    issue(y0 - 2, pend);
    {
        float l0[NP], r0[NP];
        exchange_raw(pend, l0, r0);
#pragma unroll
        for (int p = 0; p < NP; ++p) { cL[p] = l0[p]; cC[p] = pend.v[p]; cR[p] = r0[p]; }
        civ = pend.iv;
    }
    issue(y0 - 1, pend);

    // One raw row: the neighbours of row r+1 are exchanged and row r+2 requested, the rolling sums take row r (cL, cC, cR),
    // which finishes the window centred on row r-1 (sums in ay .. axy, centre values in wraw, identity loss in wiv).
#define DS_FB_RAW_ROW(r_)                                                            \
    {                                                                                \
        float nL[NP], nC[NP], nR[NP];                                                \
        exchange_raw(pend, nL, nR);                                                  \
        _Pragma("unroll") for (int p = 0; p < NP; ++p) nC[p] = pend.v[p];            \
        const float niv = pend.iv;                                                   \
        issue((r_) + 2, pend);                                                       \
        const float yL = cL[0], yC = cC[0], yR = cR[0];                              \
        ay = roll(Sy, yL, yC, yR);                                                   \
        ayy = roll(Syy, yL * yL, yC * yC, yR * yR);                                  \
        _Pragma("unroll") for (int s = 0; s < S; ++s) {                              \
            const float xL = cL[1 + s], xC = cC[1 + s], xR = cR[1 + s];              \
            ax[s] = roll(Sx[s], xL, xC, xR);                                         \
            axx[s] = roll(Sxx[s], xL * xL, xC * xC, xR * xR);                        \
            axy[s] = roll(Sxy[s], xL * yL, xC * yC, xR * yR);                        \
        }                                                                            \
        _Pragma("unroll") for (int p = 0; p < NP; ++p) {                             \
            wraw[p] = pvC[p];                                                        \
            pvC[p] = cC[p];                                                          \
            cL[p] = nL[p];                                                           \
            cC[p] = nC[p];                                                           \
            cR[p] = nR[p];                                                           \
        }                                                                            \
        wiv = piv;                                                                   \
        piv = civ;                                                                   \
        civ = niv;                                                                   \
    }

    float wraw[NP], wiv;  // centre values / identity loss of the window row a raw row has just finished
    // WARP, channel wave 0: the sums the output wave keeps otherwise -- in LDS (two registers more and the kernel spills)
    if (WARP && c == 0) lds_s[0][lane] = lds_s[1][lane] = 0.f;
    // lead-in: rows y0-2 and y0-1 only feed the partial sums
    DS_FB_RAW_ROW(y0 - 2)
    DS_FB_RAW_ROW(y0 - 1)

    for (int sl = 0; sl < nslab; ++sl) {
        const int buf = sl & 1;
        // ---- forward of kK window rows: j = y0-1 + sl*kK + k, finished by raw row j+1
#pragma unroll
        for (int k = 0; k < kK; ++k) {
            const int j = y0 - 1 + sl * kK + k;
            DS_FB_RAW_ROW(j + 1)
            Pend& q = pd[k];
            target_stats(ay, ayy, q.ts);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                SsimOut so;
                ssim_from_sums(ax[s], axx[s], axy[s], q.ts, so);
                q.ax[s] = ax[s]; q.axx[s] = axx[s]; q.axy[s] = axy[s];
                lds_e[buf][k][c][s][lane] = 0.85f * clamp01(so.v) + 0.15f * fabsf(wraw[0] - wraw[1 + s]);
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) lds_p[c][k][p][lane] = wraw[p];
            lds_p[c][k][NP][lane] = wiv;
        }
        __syncthreads();  // all three channels of these rows are in the slab
        // ---- backward of the same window rows
#pragma unroll
        for (int k = 0; k < kK; ++k) {
            const int j = y0 - 1 + sl * kK + k;
            const Pend& q = pd[k];
            float E[S];
#pragma unroll
            for (int g = 0; g < S; ++g)
                E[g] = div3((lds_e[buf][k][0][g][lane] + lds_e[buf][k][1][g][lane]) + lds_e[buf][k][2][g][lane]);
            int bi = 0;
            float best = E[0];
#pragma unroll
            for (int g = 1; g < S; ++g)
                if (E[g] < best) { best = E[g]; bi = g; }
            const float ivk = lds_p[c][k][NP][lane];
            if (WARP && c == 0 && lane_out && j >= y0 && j < yend) {   // the output wave's part (see there)
                const size_t o = (size_t)b * HW + (size_t)j * W + col;
                P.out[o] = best;
                P.aux[o] = (uint8_t)bi;
                lds_s[0][lane] += best / expf(ivk + 1e-6f);  // train.py:88
                lds_s[1][lane] += best;                      // train.py:92
            }
            // gradient weight of this row's pixels: d loss / d out(p)
            const float gw = gscale * __builtin_amdgcn_exp2f((ivk + 1e-6f) * -1.4426950408889634f);
            const bool win_exists = col_exists && j >= 0 && j < H;
            float bx_ = q.ax[0], bxx = q.axx[0], bxy = q.axy[0];
#pragma unroll
            for (int s = 1; s < S; ++s) {
                const bool is = bi == s;
                bx_ = is ? q.ax[s] : bx_;
                bxx = is ? q.axx[s] : bxx;
                bxy = is ? q.axy[s] : bxy;
            }
            SsimOut so;
            ssim_from_sums(bx_, bxx, bxy, q.ts, so);
            // clamp passes the gradient iff 0 <= v <= 1
            const bool live = win_exists && so.v >= 0.0f && so.v <= 1.0f;
            const Coef kc = ssim_coefs(so, q.ts, live ? gw * (0.85f / 3.0f) : 0.0f);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const bool is = bi == s;
                lk[3 * s + 0][lane + 1] = is ? kc.A : 0.0f;
                lk[3 * s + 1][lane + 1] = is ? kc.Bc : 0.0f;
                lk[3 * s + 2][lane + 1] = is ? kc.Cc : 0.0f;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float Hn[NK];
#pragma unroll
            for (int t = 0; t < NK; ++t) {
                const float own = (bi == t / 3) ? (t % 3 == 0 ? kc.A : (t % 3 == 1 ? kc.Bc : kc.Cc)) : 0.0f;
                Hn[t] = fmaf(wR, lk[t][lane + 2], fmaf(wL, lk[t][lane], own));
            }
            // ---- output row t = j-1: windows t-1 (Hs[1]), t (Hs[0]), t+1 (Hn), reflect-pad adjoint weights
            const int t = j - 1;
            if (t >= y0 && t < yend) {
                const float wU = t == 1 ? 2.0f : 1.0f, wD = t == H - 2 ? 2.0f : 1.0f;
                const float ge = pgw * (1.0f / 3.0f);
                const float Y = praw[0];
#pragma unroll
                for (int s = 0; s < S; ++s) {
                    const float gA = fmaf(wD, Hn[3 * s + 0], fmaf(wU, Hs[1][3 * s + 0], Hs[0][3 * s + 0]));
                    const float gB = fmaf(wD, Hn[3 * s + 1], fmaf(wU, Hs[1][3 * s + 1], Hs[0][3 * s + 1]));
                    const float gC = fmaf(wD, Hn[3 * s + 2], fmaf(wU, Hs[1][3 * s + 2], Hs[0][3 * s + 2]));
                    const float X = praw[1 + s];
                    float gg = fmaf(Y, gC, fmaf(X, gB, gA));
                    const float diff = Y - X;
                    const float sgn = diff > 0.f ? -1.0f : (diff < 0.f ? 1.0f : 0.0f);
                    if (pam == s) gg = fmaf(0.15f * ge, sgn, gg);
                    if (WARP) lds_g[buf][k][c][s][lane] = gg;
                    if (store_g && lane_out)
                        buf_store_f32(gg, gp, (unsigned)col * 4u, ((s * 3 + c) * uHW + (unsigned)(t * W)) * 4u);
                }
            }
#pragma unroll
            for (int t2 = 0; t2 < NK; ++t2) {
                Hs[1][t2] = Hs[0][t2];
                Hs[0][t2] = Hn[t2];
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) praw[p] = lds_p[c][k][p][lane];
            pgw = gw;
            pam = bi;
        }
    }
#undef DS_FB_RAW_ROW
    if (WARP) {
        __syncthreads();  // the fourth wave's last iteration: it takes the rows of the last slab after this barrier
        if (c == 0) {
            const float r0 = wave_sum(lds_s[0][lane]), r1 = wave_sum(lds_s[1][lane]);
            if (lane == 0) {
                P.wave_partials[2 * wg + 0] = r0;
                P.wave_partials[2 * wg + 1] = r1;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// SSIM-only backward (autograd of metrics.reprojection_error w.r.t. preds), one plane per "image"
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * kWaves) void ssim_bwd_kernel(const float* __restrict__ preds,
                                                               const float* __restrict__ targets,
                                                               const float* __restrict__ g_out, int nplanes, int H, int W,
                                                               int nbx, int nby, int RB, float* __restrict__ g_preds) {
    const BandGeom g = band_geom<2>(nplanes, H, W, nbx, nby, RB);
    if (!g.active) return;
    const unsigned cl = (unsigned)g.cl * 4u;  // lane's column byte offset: the only per-lane part of a load address
    const size_t HW = (size_t)H * W;
    const float* xp = preds + (size_t)g.b * HW;
    const float* yp = targets + (size_t)g.b * HW;
    const float* wp = g_out + (size_t)g.b * HW;
    float* gp = g_preds + (size_t)g.b * HW + g.col;
    const bool col_exists = g.col >= 0 && g.col < W;
    const bool lane_out = g.lane >= 2 && g.lane <= 61 && g.col < W;
    const float wL = g.col == 1 ? 2.0f : 1.0f, wR = g.col == W - 2 ? 2.0f : 1.0f;
    float X[3], Y[3], GW[3], ACC[3][3];
    auto place = [&](int slot, int r) {
        const size_t ro = (size_t)reflect_idx(r, H) * W;
        X[slot] = (xp + ro)[cl >> 2];
        Y[slot] = (yp + ro)[cl >> 2];
        GW[slot] = (wp + ro)[cl >> 2];
    };
    auto window = [&](int T, int M, int Bt, bool exists, float wUn, float wDp) {
        float ax, axx, axy, ay, ayy;
        win_target(Y[T], Y[M], Y[Bt], ay, ayy);
        win_pair(X[T], X[M], X[Bt], Y[T], Y[M], Y[Bt], ax, axx, axy);
        TgtStats ts;
        target_stats(ay, ayy, ts);
        SsimOut so;
        ssim_from_sums(ax, axx, axy, ts, so);
        const float gv = (exists && so.v >= 0.0f && so.v <= 1.0f) ? GW[M] : 0.0f;
        const Coef k = ssim_coefs(so, ts, gv);
        float h[3];
        spread3(k.A, k.Bc, k.Cc, wL, wR, h[0], h[1], h[2]);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            ACC[Bt][i] = wUn * h[i];
            ACC[M][i] += h[i];
            ACC[T][i] = fmaf(wDp, h[i], ACC[T][i]);
        }
    };
    auto emit = [&](int t, int slot) {
        if (lane_out) gp[(size_t)t * W] = fmaf(Y[slot], ACC[slot][2], fmaf(X[slot], ACC[slot][1], ACC[slot][0]));
    };
    auto wU = [&](int t) { return t == 1 ? 2.0f : 1.0f; };
    auto wD = [&](int t) { return t == H - 2 ? 2.0f : 1.0f; };
    place(0, g.y0 - 2);
    place(1, g.y0 - 1);
#pragma unroll
    for (int i = 0; i < 3; ++i) ACC[0][i] = ACC[1][i] = 0.f;
    const int rlast = g.yend + 1;
    for (int r = g.y0; r <= rlast; r += 3) {
        place(2, r);
        window(0, 1, 2, col_exists && r - 1 >= 0 && r - 1 < H, wU(r), wD(r - 2));
        if (r - 2 >= g.y0) emit(r - 2, 0);
        if (r + 1 > rlast) break;
        place(0, r + 1);
        window(1, 2, 0, col_exists && r >= 0 && r < H, wU(r + 1), wD(r - 1));
        if (r - 1 >= g.y0) emit(r - 1, 1);
        if (r + 2 > rlast) break;
        place(1, r + 2);
        window(2, 0, 1, col_exists && r + 1 >= 0 && r + 1 < H, wU(r + 2), wD(r));
        emit(r, 2);
    }
}

// ---------------------------------------------------------------------------------------------------
// host-side launch geometry
// ---------------------------------------------------------------------------------------------------
struct Geom { int nbx, nby, RB, nblocks, nwaves; };

inline int simd_count() { return 4 * ds::cu_count(); }  // SIMDs of the current device (4 per CU)

// Rows per wave.  The kernels are VALU-bound, so a launch takes about (waves on the busiest SIMD) x (rows a wave
// walks, halo rows included); a SIMD needs two resident waves to issue back to back.  Pick the row chunk that
// minimises that product: small batches get chunks sized so that the waves divide evenly over the SIMDs, large
// ones get long chunks (less halo re-computation).  Pure function of (B,H,W,halo): the workspace size, the
// partial-sum layout and both launches of a step derive the same geometry from it.
inline Geom make_geom(int B, int H, int W, int halo, int useful, double halo_row_cost) {
    Geom g;
    g.nbx = (W + useful - 1) / useful;
    const double simds = (double)simd_count();
    const long long bands = (long long)B * g.nbx;
    double best = 1e300;
    int best_nby = 1;
    for (int nby = 1; nby <= (H + 3) / 4; ++nby) {
        const int RB = (H + nby - 1) / nby;
        if (nby > 1 && (nby - 1) * RB >= H) continue;  // same chunking as a smaller nby
        const double w = (double)(bands * nby) / simds;
        const double rounds = w <= 1.0 ? 1.85 : (w <= 4.0 ? ceil(w - 1e-9) : w + 0.5);
        const double cost = rounds * (RB + 2 * halo * halo_row_cost);
        if (cost < best - 1e-9) { best = cost; best_nby = nby; }
    }
    g.nby = best_nby;
    g.RB = (H + g.nby - 1) / g.nby;
    g.nwaves = B * g.nbx * g.nby;
    g.nblocks = (g.nwaves + kWaves - 1) / kWaves;
    return g;
}

// Forward loss kernels.  The channel-split LDS-exchange kernel (one 4-wave workgroup per band of 64 columns and row
// chunk) is used wherever a wave's partial sums fit the register file without spilling (G = groups per launch <= 6:
// 111 VGPRs at G = 4, 154 at G = 6); G = 8 (S = 4 with the identity groups) runs as two four-group launches
// (fused_loss_fwd_partials).
inline bool fwd_uses_lds(int S, bool idt) { return (idt ? 2 * S : S) <= 6; }

// Row chunking of the channel-split kernel: one workgroup per (band, row chunk); a CU holds four of them (VGPR
// limit), the chip 4 * CUs at a time.  A chunk costs its rows plus about one row of lead-in, so chunks should be long;
// on the other hand the launch balances itself only through workgroups that start when others finish.  Measured
// (gpurun_out/kb14.txt): 320x1024 B=64 0.561 / 0.529 / 0.510 / 0.506 ms with 1 / 2 / 4 / 8 chunks per band, 192x640
// B=12 flat between 8 and 24 chunks.  Rule: about eight rounds of workgroups, but chunks of at least 12 rows.
// Pure function of (B,H,W): workspace size and launch derive the same geometry.
// Small launches (fewer than ~4 rounds even with the shortest chunks) are a different regime: the tail decides.  1920
// workgroups on 1024 slots run as one full round and one at 7/8 fill; 960 workgroups with chunks twice as long finish
// sooner (B=12, 192x640: 50.3 -> 47.7 us; gpurun_out/r02_kb1_nby_sweep.txt), and a launch a little over one round is the
// worst case (+10 %).  So: if the launch cannot reach four rounds, use whole rounds -- the largest chunk count whose
// workgroups fill k resident rounds exactly or stay just below.
// H, min_rows: the image height and the shortest chunk of the regular regime.  A launch that cannot fill even one round
// with chunks of min_rows is a third regime (pyramid levels, small batches): the chip is mostly empty, the run time is
// the length of one workgroup's walk (about a microsecond per row), so chunks go down to 4 rows -- the lead-in rows a
// short chunk repeats cost nothing there (24x80 x 12 targets: 27 -> 9 us for the fused loss kernel).
inline long long pick_chunks(long long per_chunk, long long capacity, int H, int min_rows) {
    int max_nby = H / min_rows > 1 ? H / min_rows : 1;
    long long nby = (DS_ROUNDS * capacity + per_chunk - 1) / per_chunk;   // about DS_ROUNDS rounds
    if (nby <= max_nby) return nby;
    if (per_chunk * max_nby < capacity) {
        max_nby = H / 4 > 1 ? H / 4 : 1;
        nby = capacity / per_chunk;
        if (nby > max_nby) nby = max_nby;
        return nby < 1 ? 1 : nby;
    }
    const double rounds_at_max = (double)per_chunk * max_nby / (double)capacity;
    if (rounds_at_max >= 4.0) return max_nby;
    long long k = (long long)rounds_at_max;
    if (k < 1) k = 1;
    nby = k * capacity / per_chunk;
    if (nby > max_nby) nby = max_nby;
    return nby < 1 ? 1 : nby;
}

inline Geom make_fwd_lds_geom(int B, int H, int W, int groups, int per_cu_given = 0) {
    Geom g;
    g.nbx = (W + 63) / 64;
    // workgroups resident at once: one wave per SIMD each, as many per CU as the VGPR count of the variant allows
    // (111 VGPRs at G = 4: four; 68 at G = 2: seven)
    const long long per_cu = per_cu_given > 0 ? per_cu_given :
                             groups <= 1 ? 8 : (groups == 2 ? 7 : (groups == 3 ? 5 : (groups == 4 ? 4 : 3)));
    const long long capacity = (long long)simd_count() / 4 * per_cu;
    const long long bands = (long long)B * g.nbx;
    const long long nby = pick_chunks(bands, capacity, H, 12);
    g.RB = (H + (int)nby - 1) / (int)nby;
    g.nby = (H + g.RB - 1) / g.RB;
    g.nblocks = B * g.nbx * g.nby;
    g.nwaves = g.nblocks;  // one partial-sum slot per workgroup
    return g;
}
inline Geom make_fwd_geom(int B, int H, int W, int S, bool idt) {
    return make_fwd_lds_geom(B, H, W, idt && fwd_uses_lds(S, true) ? 2 * S : S);   // (S = 4 with identity groups: two S-group launches)
}

inline bool bad_dims(int B, int S, int H, int W) {
    return B <= 0 || S <= 0 || H < 2 || W < 2 || (long long)H * W > (1ll << 30) || (long long)B * H * W > (1ll << 31) ||
           (long long)DS_MAX_SOURCES * 3 * H * W * 4 >= (1ll << 32);  // 32-bit buffer offsets inside one launch's planes of an image
}

// P carries everything but the geometry, which depends on the kernel variant S selects
template <bool IDT>
int launch_photo_fwd(int S, PhotoFwdParams P, Geom* geom_out, hipStream_t st) {
    if (S < 1 || S > DS_MAX_SOURCES) return DS_ERR_UNSUPPORTED;
    const Geom gm = make_fwd_geom(P.B, P.H, P.W, S, IDT);
    P.nbx = gm.nbx; P.nby = gm.nby; P.RB = gm.RB;
    if (geom_out) *geom_out = gm;
    dim3 g(gm.nblocks), b(64 * kWaves);
    if (fwd_uses_lds(S, IDT)) {
        b = dim3(64 * kFwdWaves);
        switch (IDT ? 2 * S : S) {  // instantiate only the variants fwd_uses_lds admits
            case 1: hipLaunchKernelGGL((photo_fwd_kernel<1, false>), g, b, 0, st, P); break;
            case 2: if (IDT) hipLaunchKernelGGL((photo_fwd_kernel<1, true>), g, b, 0, st, P);
                    else hipLaunchKernelGGL((photo_fwd_kernel<2, false>), g, b, 0, st, P);
                    break;
            case 3: hipLaunchKernelGGL((photo_fwd_kernel<3, false>), g, b, 0, st, P); break;
            case 4: if (IDT) hipLaunchKernelGGL((photo_fwd_kernel<2, true>), g, b, 0, st, P);
                    else hipLaunchKernelGGL((photo_fwd_kernel<4, false>), g, b, 0, st, P);
                    break;
            case 6: hipLaunchKernelGGL((photo_fwd_kernel<3, true>), g, b, 0, st, P); break;
        }
    } else {
        return DS_ERR_UNSUPPORTED;   // more than six groups (S = 4 with the identity groups): two launches, fused_loss_fwd_partials
    }
    DS_CHECK_LAUNCH();
    return DS_OK;
}

// Backward loss kernel: units of (band of 62 output columns, row chunk, channel), four to a workgroup.  Chunks cost
// about three rows of lead-in/lead-out, so at least 16 rows; otherwise about eight rounds of workgroups (see
// make_fwd_lds_geom).  nblocks counts workgroups, nwaves units.
inline Geom make_bwd_geom(int B, int H, int W) {
    Geom g;
    g.nbx = (W + 61) / 62;
    const long long capacity = (long long)simd_count() * 4;  // units resident at once (four waves per SIMD)
    const long long per_chunk = (long long)B * g.nbx * 3;
    const long long nby = pick_chunks(per_chunk, capacity, H, 16);
    g.RB = (H + (int)nby - 1) / (int)nby;
    g.nby = (H + g.RB - 1) / g.RB;
    g.nwaves = B * g.nbx * g.nby * 3;
    g.nblocks = (g.nwaves + kWaves - 1) / kWaves;
    return g;
}

// Fused forward+backward kernel: one workgroup per (band of 62 output columns, row chunk); 4 workgroups per CU.
inline Geom make_fb_geom(int B, int H, int W, int S = 2) {
    Geom g;
    g.nbx = (W + 61) / 62;
    const long long capacity = S <= 2 ? simd_count() : simd_count() / 4 * 3;  // workgroups resident at once: four per CU (three at S = 3)
    const long long per_chunk = (long long)B * g.nbx;
    const long long nby = pick_chunks(per_chunk, capacity, H, 16);
    g.RB = (H + (int)nby - 1) / (int)nby;
    g.nby = (H + g.RB - 1) / g.RB;
    g.nblocks = B * g.nbx * g.nby;
    g.nwaves = g.nblocks;  // one partial-sum slot per workgroup
    return g;
}

template <bool FUSED>
int launch_photo_bwd(int S, const PhotoBwdParams& P, int nblocks, hipStream_t st) {
    dim3 g(nblocks), b(64 * kWaves);
    switch (S) {
        case 1: hipLaunchKernelGGL((photo_bwd_kernel<1, FUSED>), g, b, 0, st, P); break;
        case 2: hipLaunchKernelGGL((photo_bwd_kernel<2, FUSED>), g, b, 0, st, P); break;
        case 3: hipLaunchKernelGGL((photo_bwd_kernel<3, FUSED>), g, b, 0, st, P); break;
        case 4: hipLaunchKernelGGL((photo_bwd_kernel<4, FUSED>), g, b, 0, st, P); break;
        default: return DS_ERR_UNSUPPORTED;
    }
    DS_CHECK_LAUNCH();
    return DS_OK;
}

}  // namespace

extern "C" {

int ds_ssim_fwd(const float* preds, const float* targets, int nplanes, int H, int W, float* out, void* stream) {
    if (!preds || !targets || !out || bad_dims(nplanes, 1, H, W)) return DS_ERR_INVALID_ARG;
    const Geom g = make_geom(nplanes, H, W, 1, 62, 1.0);
    hipLaunchKernelGGL(ssim_fwd_kernel, dim3(g.nblocks), dim3(64 * kWaves), 0, (hipStream_t)stream, preds, targets, nplanes,
                       H, W, g.nbx, g.nby, g.RB, out);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

int ds_ssim_bwd(const float* preds, const float* targets, const float* g_out, int nplanes, int H, int W, float* g_preds,
                void* stream) {
    if (!preds || !targets || !g_out || !g_preds || bad_dims(nplanes, 1, H, W)) return DS_ERR_INVALID_ARG;
    const Geom g = make_geom(nplanes, H, W, 2, 60, 1.0);
    hipLaunchKernelGGL(ssim_bwd_kernel, dim3(g.nblocks), dim3(64 * kWaves), 0, (hipStream_t)stream, preds, targets, g_out,
                       nplanes, H, W, g.nbx, g.nby, g.RB, g_preds);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

int ds_reprojection_loss_fwd(const float* preds, const float* targets, long long tgt_stride_b, long long tgt_stride_s,
                             int B, int S, int H, int W, float* out, uint8_t* aux, void* stream) {
    if (!preds || !targets || !out || bad_dims(B, S, H, W) || S > 32) return DS_ERR_INVALID_ARG;
    const long long HW = (long long)H * W;
    PhotoFwdParams P{};
    P.pred_stride_b = (long long)S * 3 * HW;
    P.tgt_stride_b = tgt_stride_b;
    P.B = B; P.H = H; P.W = W;
    P.out = out; P.aux = aux;
    // one launch per chunk of sources that share a target; per-source targets go one source at a time
    const int chunk = tgt_stride_s == 0 ? DS_MAX_SOURCES : 1;
    for (int s0 = 0; s0 < S; s0 += chunk) {
        const int n = (S - s0) < chunk ? (S - s0) : chunk;
        P.preds = preds + (long long)s0 * 3 * HW;
        P.target = targets + (long long)s0 * tgt_stride_s;
        P.s_offset = s0;
        int rc = launch_photo_fwd<false>(n, P, nullptr, (hipStream_t)stream);
        if (rc != DS_OK) return rc;
    }
    return DS_OK;
}

int ds_reprojection_loss_bwd(const float* preds, const float* targets, long long tgt_stride_b, long long tgt_stride_s,
                             const float* g_out, const uint8_t* aux, int B, int S, int H, int W, float* g_preds,
                             void* stream) {
    if (!preds || !targets || !g_out || !aux || !g_preds || bad_dims(B, S, H, W) || S > 32) return DS_ERR_INVALID_ARG;
    const Geom g = make_bwd_geom(B, H, W);
    const long long HW = (long long)H * W;
    PhotoBwdParams P{};
    P.pred_stride_b = (long long)S * 3 * HW;
    P.tgt_stride_b = tgt_stride_b;
    P.g_out = g_out; P.aux = aux;
    P.B = B; P.H = H; P.W = W; P.nbx = g.nbx; P.nby = g.nby; P.RB = g.RB;
    P.inv_hw = 1.0f;
    const int chunk = tgt_stride_s == 0 ? DS_MAX_SOURCES : 1;
    for (int s0 = 0; s0 < S; s0 += chunk) {
        const int n = (S - s0) < chunk ? (S - s0) : chunk;
        P.preds = preds + (long long)s0 * 3 * HW;
        P.g_preds = g_preds + (long long)s0 * 3 * HW;
        P.target = targets + (long long)s0 * tgt_stride_s;
        P.s_offset = s0;
        int rc = launch_photo_bwd<false>(n, P, g.nblocks, (hipStream_t)stream);
        if (rc != DS_OK) return rc;
    }
    return DS_OK;
}

size_t ds_fused_loss_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H < 2 || W < 2) return 0;
    // enough for every forward kernel variant (the entry point has no S)
    size_t n = 0;
    for (int S = 1; S <= DS_MAX_SOURCES; ++S) {
        const size_t k = (size_t)make_fwd_geom(B, H, W, S, true).nwaves;
        if (k > n) n = k;
        const size_t k2 = (size_t)make_fwd_geom(B, H, W, S, false).nwaves;   // identity map given (photo_fwd_idm_kernel)
        if (k2 > n) n = k2;
    }
    return n * 2 * sizeof(float);
}

int ds_fused_loss_fwd(const float* warped, const float* src, const float* target, int B, int S, int H, int W,
                      float* out_loss, float* idt_loss, uint8_t* aux, float* sums, void* workspace,
                      size_t workspace_bytes, void* stream) {
    if (!sums) return DS_ERR_INVALID_ARG;
    int wpi = 0;
    int rc = ds::fused_loss_fwd_partials(warped, src, target, B, S, H, W, out_loss, idt_loss, aux, workspace, workspace_bytes,
                                         &wpi, (hipStream_t)stream);
    if (rc != DS_OK) return rc;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, (const float*)workspace, wpi,
                       1.0 / ((double)H * W), sums);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

size_t ds_fused_loss_fb_workspace_bytes(int B, int H, int W) { return ds::fused_loss_fb_workspace_bytes(B, H, W); }

int ds_fused_loss_fb(const float* warped, const float* target, const float* idt_loss, const float* g_sums, int B, int S, int H,
                     int W, float* out_loss, uint8_t* aux, float* sums, float* g_warped, void* workspace,
                     size_t workspace_bytes, void* stream) {
    if (!sums) return DS_ERR_INVALID_ARG;
    int wpi = 0;
    int rc = ds::fused_loss_fb_partials(warped, target, idt_loss, g_sums, B, S, H, W, out_loss, aux, g_warped, workspace,
                                        workspace_bytes, &wpi, (hipStream_t)stream);
    if (rc != DS_OK) return rc;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, (const float*)workspace, wpi,
                       1.0 / ((double)H * W), sums);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

int ds_fused_loss_bwd(const float* warped, const float* target, const float* idt_loss, const uint8_t* aux,
                      const float* g_sums, int B, int S, int H, int W, float* g_warped, void* stream) {
    // g_sums may be NULL: unit upstream gradient for every target
    if (!warped || !target || !idt_loss || !aux || !g_warped || bad_dims(B, S, H, W)) return DS_ERR_INVALID_ARG;
    if (S > DS_MAX_SOURCES) return DS_ERR_UNSUPPORTED;
    const Geom g = make_bwd_geom(B, H, W);
    const long long HW = (long long)H * W;
    PhotoBwdParams P{};
    P.preds = warped; P.target = target; P.idt = idt_loss; P.g_sums = g_sums; P.aux = aux;
    P.pred_stride_b = (long long)S * 3 * HW;
    P.tgt_stride_b = 3 * HW;
    P.B = B; P.H = H; P.W = W; P.nbx = g.nbx; P.nby = g.nby; P.RB = g.RB;
    P.inv_hw = (float)(1.0 / (double)HW);
    P.g_preds = g_warped;
    return launch_photo_bwd<true>(S, P, g.nblocks, (hipStream_t)stream);
}

}  // extern "C"

int ds::fused_loss_fwd_idm_partials(const float* warped, const float* target, const float* idt_loss, int B, int S, int H, int W,
                                    float* out_loss, uint8_t* aux, void* workspace, size_t workspace_bytes,
                                    int* waves_per_image, hipStream_t stream) {
    if (!warped || !target || !idt_loss || !out_loss || !aux || bad_dims(B, S, H, W)) return DS_ERR_INVALID_ARG;
    if (S > DS_MAX_SOURCES) return DS_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < ds_fused_loss_workspace_bytes(B, H, W)) return DS_ERR_WORKSPACE;
    const long long HW = (long long)H * W;
    PhotoFwdParams P{};
    P.preds = warped; P.target = target;
    P.pred_stride_b = (long long)S * 3 * HW;
    P.tgt_stride_b = 3 * HW;
    P.B = B; P.H = H; P.W = W;
    P.out = out_loss; P.idt = const_cast<float*>(idt_loss); P.aux = aux; P.wave_partials = (float*)workspace;
    const Geom g = make_fwd_geom(B, H, W, S, false);
    P.nbx = g.nbx; P.nby = g.nby; P.RB = g.RB;
    const dim3 grid(g.nblocks), block(64 * kFwdWaves);
    switch (S) {
        case 1: hipLaunchKernelGGL((photo_fwd_idm_kernel<1>), grid, block, 0, stream, P); break;
        case 2: hipLaunchKernelGGL((photo_fwd_idm_kernel<2>), grid, block, 0, stream, P); break;
        case 3: hipLaunchKernelGGL((photo_fwd_idm_kernel<3>), grid, block, 0, stream, P); break;
        default: hipLaunchKernelGGL((photo_fwd_idm_kernel<4>), grid, block, 0, stream, P); break;
    }
    DS_CHECK_LAUNCH();
    *waves_per_image = g.nbx * g.nby;
    return DS_OK;
}

int ds::warp_and_identity(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                          const float* target, int B, int S, int H, int W, float* warped, float* idt_loss,
                          hipStream_t stream) {
    if (!K || !Kinv || !pose || !depth || !src || !target || !warped || !idt_loss || bad_dims(B, S, H, W))
        return DS_ERR_INVALID_ARG;
    if (S > DS_MAX_SOURCES) return DS_ERR_UNSUPPORTED;
    const long long HW = (long long)H * W;
    PhotoFwdParams P{};
    P.preds = src; P.target = target;
    P.pred_stride_b = (long long)S * 3 * HW;
    P.tgt_stride_b = 3 * HW;
    P.B = B; P.H = H; P.W = W;
    P.out = idt_loss; P.aux = nullptr;
    // S <= 3: every workgroup is a loss workgroup whose waves also warp (eight / five / four per CU at S = 1 / 2 / 3);
    // S = 4: warp workgroups and loss workgroups side by side in one launch (three sources in the E wave spill)
    // (the chunking is flat in the assumed residency between 4 and 8 per CU, profiles/r02_kb3_warp_identity_one_launch.txt)
    const bool ew = S <= 3;
    const Geom gm = ew ? make_fwd_lds_geom(B, H, W, S, S == 1 ? 8 : (S == 2 ? 5 : 4)) : make_fwd_geom(B, H, W, S, false);
    P.nbx = gm.nbx; P.nby = gm.nby; P.RB = gm.RB;
    WarpIdtParams Q{};
    Q.K = K; Q.Kinv = Kinv; Q.pose = pose; Q.depth = depth; Q.src = src; Q.warped = warped;
    Q.G = warp_geom(H, W, H, W);
    const int rows_per_tile = 4 * kFwdRows;
    Q.tiles_per_image = Q.G.nbx * ((H + rows_per_tile - 1) / rows_per_tile);
    Q.n_warp = B * S * Q.tiles_per_image;
    Q.n_loss = gm.nblocks;
    Q.groups_warp = (Q.n_warp + 7) / 8;
    Q.groups_total = Q.groups_warp + (Q.n_loss + 7) / 8;
    Q.warp_first = (long long)Q.n_loss <= 7ll * simd_count() ? 1 : 0;   // fewer than ~4 resident rounds of loss workgroups
    const dim3 block(64 * kFwdWaves);
    if (ew) {
        const dim3 grid(Q.n_loss);
        switch (S) {
            case 1: hipLaunchKernelGGL((warp_idt_kernel<1, true>), grid, block, 0, stream, Q, P); break;
            case 2: hipLaunchKernelGGL((warp_idt_kernel<2, true>), grid, block, 0, stream, Q, P); break;
            default: hipLaunchKernelGGL((warp_idt_kernel<3, true>), grid, block, 0, stream, Q, P); break;
        }
    } else {
        const dim3 grid(8 * Q.groups_total);
        switch (S) {
            case 1: hipLaunchKernelGGL((warp_idt_kernel<1, false>), grid, block, 0, stream, Q, P); break;
            case 2: hipLaunchKernelGGL((warp_idt_kernel<2, false>), grid, block, 0, stream, Q, P); break;
            case 3: hipLaunchKernelGGL((warp_idt_kernel<3, false>), grid, block, 0, stream, Q, P); break;
            default: hipLaunchKernelGGL((warp_idt_kernel<4, false>), grid, block, 0, stream, Q, P); break;
        }
    }
    DS_CHECK_LAUNCH();
    return DS_OK;
}

bool ds::fused_loss_fb_supported(int S) { return S >= 1 && S <= 2; }
bool ds::fused_loss_fbw_supported(int S) { return S >= 1 && S <= 3; }

size_t ds::fused_loss_fb_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H < 2 || W < 2) return 0;
    return (size_t)make_fb_geom(B, H, W).nwaves * 2 * sizeof(float);
}

int ds::fused_loss_fb_partials(const float* warped, const float* target, const float* idt_loss, const float* g_sums, int B,
                               int S, int H, int W, float* out_loss, uint8_t* aux, float* g_warped, void* workspace,
                               size_t workspace_bytes, int* waves_per_image, hipStream_t stream) {
    if (!warped || !target || !idt_loss || !out_loss || !aux || !g_warped || bad_dims(B, S, H, W)) return DS_ERR_INVALID_ARG;
    if (!fused_loss_fb_supported(S)) return DS_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < fused_loss_fb_workspace_bytes(B, H, W)) return DS_ERR_WORKSPACE;
    const Geom g = make_fb_geom(B, H, W);
    PhotoFbParams P{};
    P.preds = warped; P.target = target; P.idt = idt_loss; P.g_sums = g_sums;
    P.B = B; P.H = H; P.W = W; P.nbx = g.nbx; P.nby = g.nby; P.RB = g.RB;
    P.inv_hw = (float)(1.0 / ((double)H * W));
    P.out = out_loss; P.aux = aux; P.wave_partials = (float*)workspace; P.g_preds = g_warped;
    const dim3 grid(g.nblocks), block(64 * kFwdWaves);
    if (S == 1) hipLaunchKernelGGL((photo_fb_kernel<1, false>), grid, block, 0, stream, P);
    else hipLaunchKernelGGL((photo_fb_kernel<2, false>), grid, block, 0, stream, P);
    DS_CHECK_LAUNCH();
    *waves_per_image = g.nbx * g.nby;
    return DS_OK;
}

size_t ds::fused_loss_fbw_workspace_bytes(int B, int S, int H, int W) {
    if (B <= 0 || S <= 0 || H < 2 || W < 2) return 0;
    const Geom g = make_fb_geom(B, H, W, S);
    // per-workgroup loss partials, then per-workgroup pose partials of every (target, source)
    return (size_t)g.nwaves * 2 * sizeof(float) + (size_t)g.nwaves * S * 12 * sizeof(float);
}

int ds::fused_loss_fbw_partials(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                                const float* warped, const float* target, const float* idt_loss, const float* g_sums, int B,
                                int S, int H, int W, float* out_loss, uint8_t* aux, float* g_warped, float* g_depth,
                                void* workspace, size_t workspace_bytes, int* waves_per_image, float** pose_partials,
                                hipStream_t stream) {
    if (!K || !Kinv || !pose || !depth || !src || !warped || !target || !idt_loss || !out_loss || !aux || !g_depth ||
        bad_dims(B, S, H, W) || (long long)B * S > 65535)
        return DS_ERR_INVALID_ARG;
    if (!fused_loss_fbw_supported(S)) return DS_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < fused_loss_fbw_workspace_bytes(B, S, H, W)) return DS_ERR_WORKSPACE;
    const Geom g = make_fb_geom(B, H, W, S);
    PhotoFbParams P{};
    P.preds = warped; P.target = target; P.idt = idt_loss; P.g_sums = g_sums;
    P.B = B; P.H = H; P.W = W; P.nbx = g.nbx; P.nby = g.nby; P.RB = g.RB;
    P.inv_hw = (float)(1.0 / ((double)H * W));
    P.out = out_loss; P.aux = aux; P.wave_partials = (float*)workspace; P.g_preds = g_warped;
    P.K = K; P.Kinv = Kinv; P.pose = pose; P.depth = depth; P.src = src; P.g_depth = g_depth;
    P.pose_partials = (float*)workspace + (size_t)g.nwaves * 2;
    P.G = warp_geom(H, W, H, W);
    const dim3 grid(g.nblocks), block(64 * kFwdWaves);
    if (S == 1) hipLaunchKernelGGL((photo_fb_kernel<1, true>), grid, block, 0, stream, P);
    else if (S == 2) hipLaunchKernelGGL((photo_fb_kernel<2, true>), grid, block, 0, stream, P);
    else hipLaunchKernelGGL((photo_fb_kernel<3, true>), grid, block, 0, stream, P);
    DS_CHECK_LAUNCH();
    *waves_per_image = g.nbx * g.nby;
    *pose_partials = P.pose_partials;
    return DS_OK;
}

int ds::fused_loss_fwd_partials(const float* warped, const float* src, const float* target, int B, int S, int H, int W,
                                float* out_loss, float* idt_loss, uint8_t* aux, void* workspace, size_t workspace_bytes,
                                int* waves_per_image, hipStream_t stream) {
    if (!warped || !src || !target || !out_loss || !idt_loss || !aux || bad_dims(B, S, H, W)) return DS_ERR_INVALID_ARG;
    if (S > DS_MAX_SOURCES) return DS_ERR_UNSUPPORTED;
    if (!workspace || workspace_bytes < ds_fused_loss_workspace_bytes(B, H, W)) return DS_ERR_WORKSPACE;
    const long long HW = (long long)H * W;
    PhotoFwdParams P{};
    P.preds = warped; P.src = src; P.target = target;
    P.pred_stride_b = (long long)S * 3 * HW;
    P.tgt_stride_b = 3 * HW;
    P.B = B; P.H = H; P.W = W;
    P.out = out_loss; P.idt = idt_loss; P.aux = aux; P.wave_partials = (float*)workspace;
    Geom g;
    if (!fwd_uses_lds(S, true)) {
        // S = 4: eight groups do not fit one workgroup's registers and LDS.  The identity map first (the un-warped sources as
        // the "predictions" of a four-group launch, min over them -> idt_loss), then the warped groups against it -- the
        // same arithmetic in two launches.  (Rounds 1-2 had a register-lean DPP kernel for this case alone, 25 % slower per
        // pixel; the reference never runs S = 4, constants.py:6.)
        P.preds = src; P.src = nullptr; P.out = idt_loss; P.idt = nullptr; P.aux = nullptr; P.wave_partials = nullptr;
        int rc = launch_photo_fwd<false>(S, P, &g, stream);
        if (rc != DS_OK) return rc;
        return fused_loss_fwd_idm_partials(warped, target, idt_loss, B, S, H, W, out_loss, aux, workspace, workspace_bytes,
                                           waves_per_image, stream);
    }
    int rc = launch_photo_fwd<true>(S, P, &g, stream);
    if (rc != DS_OK) return rc;
    *waves_per_image = g.nbx * g.nby;
    return DS_OK;
}
