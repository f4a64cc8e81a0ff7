/*
 * ds_oracle.c -- CPU restatement of the view-synthesis / reprojection-loss hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity oracle: tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may call it; the product (depth_estimation_amd/) never does.
 *
 * It restates, per pixel and in plain C, what the reference's PyTorch composite computes
 * (reference = banaga-benjamin/Depth_Estimation, file:line cited at every function), with the
 * fp32 operation order that torch 2.10 CPU uses, so that the forward results are reproduced
 * bit-for-bit where the golden vectors under tests/golden/ (captured by tools/gen_golden.py from
 * the reference itself) say so.  Parity status: PINNED by those vectors (tests/test_oracle_golden.py).
 *
 * Numeric contract:
 *   - the three small matmuls of reproject_from_depth are k-ordered fmaf chains;
 *   - everything in the photometric part is separately rounded (build with -ffp-contract=off);
 *   - avg_pool2d is a row-major sequential 9-tap sum starting from 0, then a true divide by 9;
 *   - scalar means are accumulated in double (the reference's fp32 pairwise sums agree to ~1e-7).
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -mfma -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define DSO_API __attribute__((visibility("default")))

static const float kC1 = 0.0001f;      /* float32(0.01**2), metrics.py:46 */
static const float kC2 = 0.0009f;      /* float32(0.03**2), metrics.py:46 */

DSO_API int dso_version(void) { return 1; }

DSO_API void dso_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

DSO_API int dso_get_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* a1  synthesis.construct_pose_batch  (synthesis.py:6-33)                                     */
/* ------------------------------------------------------------------------------------------ */
/* vec (N,6) -> pose (N,3,4).  theta = ||r||, u = r/(theta+1e-6), U = skew(u),
 * R = I + sin(theta) U + (1-cos(theta)) U.U  (synthesis.py:19-30), pose = [R | t] (:32). */
DSO_API void dso_pose_from_vec(const float* vec, int N, float* pose) {
    for (int n = 0; n < N; ++n) {
        const float* r = vec + 6 * n;
        float th = sqrtf((r[0] * r[0] + r[1] * r[1]) + r[2] * r[2]);
        float den = th + 1e-6f;
        float u0 = r[0] / den, u1 = r[1] / den, u2 = r[2] / den;
        float U[9] = {0.f, -u2, u1, u2, 0.f, -u0, -u1, u0, 0.f};
        float UU[9];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                float acc = U[3 * i] * U[j];
                acc = fmaf(U[3 * i + 1], U[3 + j], acc);
                acc = fmaf(U[3 * i + 2], U[6 + j], acc);
                UU[3 * i + j] = acc;
            }
        float s = sinf(th), c1 = 1.0f - cosf(th);
        float* P = pose + 12 * n;
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) {
                float eye = (i == j) ? 1.0f : 0.0f;
                P[4 * i + j] = (eye + s * U[3 * i + j]) + c1 * UU[3 * i + j];
            }
            P[4 * i + 3] = r[3 + i];
        }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a2  synthesis.reproject_from_depth  (synthesis.py:83-116)                                   */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    float c[3];   /* Kinv3 . [x,y,1]        synthesis.py:100 */
    float P[3];   /* c * depth              synthesis.py:105 */
    float T[3];   /* pose . [P,1]           synthesis.py:108-109 */
    float q[3];   /* K3 . T                 synthesis.py:112 */
    float z;      /* |q2|<1e-6 ? 1e-6 : q2  synthesis.py:113 */
    float u, v;   /* q0/z, q1/z             synthesis.py:114 */
} dso_proj_t;

/* K, Kinv: row-major 4x4 (only [:3,:3] is read, as the reference does); M: 3x4 pose. */
static inline void project_pixel(const float* K, const float* Kinv, const float* M, float fx, float fy, float d,
                                 dso_proj_t* o) {
    for (int j = 0; j < 3; ++j) {
        float acc = fx * Kinv[4 * j + 0];
        acc = fmaf(fy, Kinv[4 * j + 1], acc);
        acc = fmaf(1.0f, Kinv[4 * j + 2], acc);
        o->c[j] = acc;
    }
    for (int j = 0; j < 3; ++j) o->P[j] = o->c[j] * d;
    for (int j = 0; j < 3; ++j) {
        float acc = o->P[0] * M[4 * j + 0];
        acc = fmaf(o->P[1], M[4 * j + 1], acc);
        acc = fmaf(o->P[2], M[4 * j + 2], acc);
        acc = fmaf(1.0f, M[4 * j + 3], acc);
        o->T[j] = acc;
    }
    for (int j = 0; j < 3; ++j) {
        float acc = o->T[0] * K[4 * j + 0];
        acc = fmaf(o->T[1], K[4 * j + 1], acc);
        acc = fmaf(o->T[2], K[4 * j + 2], acc);
        o->q[j] = acc;
    }
    o->z = (fabsf(o->q[2]) < 1e-6f) ? 1e-6f : o->q[2];
    o->u = o->q[0] / o->z;
    o->v = o->q[1] / o->z;
}

/* depth (H,W) -> coords (H,W,2) in pixel units of the H x W grid, x first. */
DSO_API void dso_reproject(const float* K, const float* Kinv, const float* pose, const float* depth, int H, int W,
                           float* coords) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            dso_proj_t p;
            project_pixel(K, Kinv, pose, (float)x, (float)y, depth[(size_t)y * W + x], &p);
            coords[((size_t)y * W + x) * 2 + 0] = p.u;
            coords[((size_t)y * W + x) * 2 + 1] = p.v;
        }
}

/* ------------------------------------------------------------------------------------------ */
/* a3  synthesis.synthesize  (synthesis.py:36-54)                                              */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    float gx, gy;          /* normalised coords written back in place (synthesis.py:50-51) */
    float ixu, iyu;        /* un-normalised, un-clipped source coords */
    int x0, y0;            /* north-west corner */
    float w_nw, w_ne, w_sw, w_se;
    float wx, ex, ny, sy;  /* distances: wx = ix-x0, ex = 1-wx, ny = iy-y0, sy = 1-ny */
} dso_samp_t;

/* normalise with SOURCE dims (synthesis.py:49-51), un-normalise as grid_sample(align_corners=True)
 * does, clip to the border (padding_mode="border"), bilinear weights. */
static inline void sample_setup(float u, float v, int Hs, int Ws, dso_samp_t* s) {
    s->gx = (u / (float)(Ws - 1)) * 2.0f - 1.0f;
    s->gy = (v / (float)(Hs - 1)) * 2.0f - 1.0f;
    float ix = (s->gx + 1.0f) * ((float)(Ws - 1) / 2.0f);
    float iy = (s->gy + 1.0f) * ((float)(Hs - 1) / 2.0f);
    s->ixu = ix;
    s->iyu = iy;
    /* min(max_val, max(in, 0)) -- a NaN coordinate is not a supported input */
    ix = fminf((float)(Ws - 1), fmaxf(ix, 0.0f));
    iy = fminf((float)(Hs - 1), fmaxf(iy, 0.0f));
    float xf = floorf(ix), yf = floorf(iy);
    s->x0 = (int)xf;
    s->y0 = (int)yf;
    s->wx = ix - xf;
    s->ex = 1.0f - s->wx;
    s->ny = iy - yf;
    s->sy = 1.0f - s->ny;
    s->w_nw = s->sy * s->ex;
    s->w_ne = s->sy * s->wx;
    s->w_sw = s->ny * s->ex;
    s->w_se = s->ny * s->wx;
}

/* The four corner products are combined as an fmaf chain in the order nw, ne, sw, se: this is the form
 * that reproduces torch 2.10's CPU grid_sampler_2d bit-for-bit on every golden case (a separately
 * rounded sum differs by 1 ulp on ~23 % of pixels; tests/test_oracle_golden.py pins it). */
static inline float bilinear_combine(float nw, float ne, float sw, float se, const dso_samp_t* s) {
    float acc = nw * s->w_nw;
    acc = fmaf(ne, s->w_ne, acc);
    acc = fmaf(sw, s->w_sw, acc);
    return fmaf(se, s->w_se, acc);
}

static inline float sample_plane(const float* plane, int Hs, int Ws, const dso_samp_t* s) {
    int x0 = s->x0, y0 = s->y0, x1 = x0 + 1, y1 = y0 + 1;
    int xin = x1 < Ws, yin = y1 < Hs; /* x0,y0 are always in range after the clip */
    float nw = plane[(size_t)y0 * Ws + x0];
    float ne = xin ? plane[(size_t)y0 * Ws + x1] : 0.0f;
    float sw = yin ? plane[(size_t)y1 * Ws + x0] : 0.0f;
    float se = (xin && yin) ? plane[(size_t)y1 * Ws + x1] : 0.0f;
    return bilinear_combine(nw, ne, sw, se, s);
}

/* src (C,Hs,Ws); grid (H,W,2) holds pixel coords on entry and NORMALISED coords on return
 * (the reference mutates its argument, synthesis.py:50-51); out (C,H,W). */
DSO_API void dso_grid_sample(const float* src, int C, int Hs, int Ws, float* grid, int H, int W, float* out) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t p = (size_t)y * W + x;
            dso_samp_t s;
            sample_setup(grid[2 * p], grid[2 * p + 1], Hs, Ws, &s);
            grid[2 * p] = s.gx;
            grid[2 * p + 1] = s.gy;
            for (int c = 0; c < C; ++c)
                out[(size_t)c * H * W + p] = sample_plane(src + (size_t)c * Hs * Ws, Hs, Ws, &s);
        }
}

/* fused a2+a3 for one (source image, depth map, pose) triple: depth (H,W) -> warped (C,H,W). */
DSO_API void dso_warp(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                      int C, int Hs, int Ws, int H, int W, float* warped) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t p = (size_t)y * W + x;
            dso_proj_t pr;
            dso_samp_t s;
            project_pixel(K, Kinv, pose, (float)x, (float)y, depth[p], &pr);
            sample_setup(pr.u, pr.v, Hs, Ws, &s);
            for (int c = 0; c < C; ++c)
                warped[(size_t)c * H * W + p] = sample_plane(src + (size_t)c * Hs * Ws, Hs, Ws, &s);
        }
}

/* ------------------------------------------------------------------------------------------ */
/* a7  metrics.reprojection_error  (metrics.py:21-51): SSIM loss per plane                     */
/* ------------------------------------------------------------------------------------------ */
static inline int reflect(int i, int L) { return i < 0 ? -i : (i > L - 1 ? 2 * (L - 1) - i : i); }

typedef struct { float mx, my, sx, sy, sxy, n, dd, v; } dso_ssim_t;

/* window statistics at (y,x) of planes X (pred) and Y (target), reflect-padded by 1 (metrics.py:33-43). */
static inline void ssim_stats(const float* X, const float* Y, int H, int W, int y, int x, dso_ssim_t* o) {
    float ax = 0.f, ay = 0.f, axx = 0.f, ayy = 0.f, axy = 0.f;
    for (int dy = -1; dy <= 1; ++dy) {
        int yy = reflect(y + dy, H);
        for (int dx = -1; dx <= 1; ++dx) {
            int xx = reflect(x + dx, W);
            float a = X[(size_t)yy * W + xx], b = Y[(size_t)yy * W + xx];
            ax = ax + a;
            ay = ay + b;
            axx = axx + a * a;
            ayy = ayy + b * b;
            axy = axy + a * b;
        }
    }
    o->mx = ax / 9.0f;
    o->my = ay / 9.0f;
    o->sx = axx / 9.0f - o->mx * o->mx;
    o->sy = ayy / 9.0f - o->my * o->my;
    o->sxy = axy / 9.0f - o->mx * o->my;
    o->n = ((2.0f * o->mx) * o->my + kC1) * (2.0f * o->sxy + kC2);              /* metrics.py:46 */
    o->dd = ((o->mx * o->mx + o->my * o->my) + kC1) * ((o->sx + o->sy) + kC2);   /* metrics.py:47 */
    o->v = (1.0f - o->n / o->dd) / 2.0f;                                        /* metrics.py:50 */
}

static inline float clamp01(float v) { return fminf(fmaxf(v, 0.0f), 1.0f); }

/* preds, targets, out: nplanes x (H,W).  target_plane_stride = 0 broadcasts one target. */
DSO_API void dso_ssim(const float* preds, const float* targets, int nplanes, int H, int W, float* out) {
    for (int n = 0; n < nplanes; ++n) {
        const float* X = preds + (size_t)n * H * W;
        const float* Y = targets + (size_t)n * H * W;
        float* O = out + (size_t)n * H * W;
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                dso_ssim_t s;
                ssim_stats(X, Y, H, W, y, x, &s);
                O[(size_t)y * W + x] = clamp01(s.v);
            }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* a8  metrics.reprojection_loss  (metrics.py:54-72)                                           */
/* ------------------------------------------------------------------------------------------ */
/* per-pixel photometric error of ONE (pred,target) image pair, mean over C=3 channels:
 * E = ((e0+e1)+e2)/3, e_c = 0.85*ssim_c + 0.15*|t-p|  (metrics.py:62-69) */
static inline float photo_pixel(const float* pred, const float* tgt, int C, int H, int W, int y, int x) {
    float acc = 0.f;
    size_t p = (size_t)y * W + x;
    for (int c = 0; c < C; ++c) {
        const float* X = pred + (size_t)c * H * W;
        const float* Y = tgt + (size_t)c * H * W;
        dso_ssim_t s;
        ssim_stats(X, Y, H, W, y, x, &s);
        float e = 0.85f * clamp01(s.v) + 0.15f * fabsf(Y[p] - X[p]);
        acc = (c == 0) ? e : acc + e;
    }
    return acc / (float)C;
}

/* preds (S,C,H,W); target (C,H,W) shared by the S preds (train.py:80 stacks one target S times);
 * out (H,W) = min over S (first index on ties), argmin (H,W) u8, per_src (S,H,W) optional. */
DSO_API void dso_reprojection_loss(const float* preds, const float* target, int S, int C, int H, int W, float* out,
                                   uint8_t* argmin, float* per_src) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t p = (size_t)y * W + x;
            float best = 0.f;
            int bi = 0;
            for (int s = 0; s < S; ++s) {
                float E = photo_pixel(preds + (size_t)s * C * H * W, target, C, H, W, y, x);
                if (per_src) per_src[(size_t)s * H * W + p] = E;
                if (s == 0 || E < best) { best = E; bi = s; }
            }
            out[p] = best;
            if (argmin) argmin[p] = (uint8_t)bi;
        }
}

/* ------------------------------------------------------------------------------------------ */
/* a9  loss assembly  (train.py:80-92) for ONE target and its S sources                        */
/* ------------------------------------------------------------------------------------------ */
/* warped,src (S,C,H,W), target (C,H,W) -> out_loss, idt_loss, ratio (H,W), argmin (H,W),
 * scalars[0] = mean(ratio) (train.py:88), scalars[1] = mean(out_loss) (train.py:92). */
DSO_API void dso_loss_assembly(const float* warped, const float* src, const float* target, int S, int C, int H, int W,
                               float* out_loss, float* idt_loss, float* ratio, uint8_t* argmin, double* scalars) {
    dso_reprojection_loss(warped, target, S, C, H, W, out_loss, argmin, NULL);
    dso_reprojection_loss(src, target, S, C, H, W, idt_loss, NULL, NULL);
    double sr = 0.0, so = 0.0;
    for (size_t p = 0; p < (size_t)H * W; ++p) {
        float r = out_loss[p] / expf(idt_loss[p] + 1e-6f);
        if (ratio) ratio[p] = r;
        sr += r;
        so += out_loss[p];
    }
    scalars[0] = sr / ((double)H * W);
    scalars[1] = so / ((double)H * W);
}

/* ------------------------------------------------------------------------------------------ */
/* backward of a9/a8/a7 w.r.t. the warped images (autograd of metrics.py:21-72 + train.py:88) */
/* ------------------------------------------------------------------------------------------ */
/* g_scale = d(total loss)/d(mean ratio) (1 for train.py:88).  g_warped (S,C,H,W) is overwritten.
 * Scatter formulation that mirrors autograd: every window spreads g/9 to its 9 (reflected) taps. */
DSO_API void dso_loss_assembly_bwd(const float* warped, const float* target, const float* idt_loss,
                                   const uint8_t* argmin, int S, int C, int H, int W, double g_scale,
                                   float* g_warped) {
    size_t HW = (size_t)H * W;
    double* acc = (double*)calloc((size_t)S * C * HW, sizeof(double));
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t p = (size_t)y * W + x;
            int s = argmin[p];
            double g_out = g_scale / (double)HW / (double)expf(idt_loss[p] + 1e-6f);
            double g_e = g_out / (double)C;
            for (int c = 0; c < C; ++c) {
                const float* X = warped + ((size_t)s * C + c) * HW;
                const float* Y = target + (size_t)c * HW;
                double* G = acc + ((size_t)s * C + c) * HW;
                dso_ssim_t st;
                ssim_stats(X, Y, H, W, y, x, &st);
                /* L1 term: d|t-p|/dp = -sign(t-p), 0 at 0 */
                float diff = Y[p] - X[p];
                G[p] += 0.15 * g_e * (diff > 0.f ? -1.0 : (diff < 0.f ? 1.0 : 0.0));
                /* clamp passes the gradient on the closed interval [0,1] */
                if (!(st.v >= 0.0f && st.v <= 1.0f)) continue;
                double g_v = 0.85 * g_e;
                double a = (double)((2.0f * st.mx) * st.my + kC1), b = (double)(2.0f * st.sxy + kC2);
                double cc = (double)((st.mx * st.mx + st.my * st.my) + kC1), e = (double)((st.sx + st.sy) + kC2);
                double dd = (double)st.dd, n = (double)st.n;
                double g_n = -g_v / (2.0 * dd);
                double g_dd = g_v * n / (2.0 * dd * dd);
                double g_a = g_n * b, g_b = g_n * a, g_c = g_dd * e, g_ee = g_dd * cc;
                double g_sxy = 2.0 * g_b, g_sx = g_ee;
                double g_mx = g_a * 2.0 * st.my + g_c * 2.0 * st.mx - 2.0 * st.mx * g_sx - (double)st.my * g_sxy;
                for (int dy = -1; dy <= 1; ++dy) {
                    int yy = reflect(y + dy, H);
                    for (int dx = -1; dx <= 1; ++dx) {
                        int xx = reflect(x + dx, W);
                        size_t t = (size_t)yy * W + xx;
                        G[t] += (g_mx + g_sx * 2.0 * (double)X[t] + g_sxy * (double)Y[t]) / 9.0;
                    }
                }
            }
        }
    for (size_t i = 0; i < (size_t)S * C * HW; ++i) g_warped[i] = (float)acc[i];
    free(acc);
}

/* ------------------------------------------------------------------------------------------ */
/* backward of a3+a2: g_warped -> g_depth, g_pose, g_src  (autograd of synthesis.py:36-54,83-116) */
/* ------------------------------------------------------------------------------------------ */
/* One (source image, depth map, pose) triple.  g_depth (H,W) overwritten; g_pose (3,4) overwritten;
 * g_src (C,Hs,Ws) ACCUMULATED into when non-NULL (callers zero it). */
DSO_API void dso_warp_bwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                          const float* g_warped, int C, int Hs, int Ws, int H, int W, float* g_depth,
                          float* g_pose, float* g_src) {
    double gM[12];
    memset(gM, 0, sizeof gM);
    double* gS = g_src ? (double*)calloc((size_t)C * Hs * Ws, sizeof(double)) : NULL;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t p = (size_t)y * W + x;
            dso_proj_t pr;
            dso_samp_t s;
            project_pixel(K, Kinv, pose, (float)x, (float)y, depth[p], &pr);
            sample_setup(pr.u, pr.v, Hs, Ws, &s);
            int x0 = s.x0, y0 = s.y0, x1 = x0 + 1, y1 = y0 + 1;
            int xin = x1 < Ws, yin = y1 < Hs;
            double gix = 0.0, giy = 0.0;
            for (int c = 0; c < C; ++c) {
                const float* pl = src + (size_t)c * Hs * Ws;
                double g = (double)g_warped[(size_t)c * H * W + p];
                double nw = pl[(size_t)y0 * Ws + x0];
                double ne = xin ? pl[(size_t)y0 * Ws + x1] : 0.0;
                double sw = yin ? pl[(size_t)y1 * Ws + x0] : 0.0;
                double se = (xin && yin) ? pl[(size_t)y1 * Ws + x1] : 0.0;
                gix += g * ((ne - nw) * (double)s.sy + (se - sw) * (double)s.ny);
                giy += g * ((sw - nw) * (double)s.ex + (se - ne) * (double)s.wx);
                if (gS) {
                    double* gp = gS + (size_t)c * Hs * Ws;
                    gp[(size_t)y0 * Ws + x0] += g * (double)s.w_nw;
                    if (xin) gp[(size_t)y0 * Ws + x1] += g * (double)s.w_ne;
                    if (yin) gp[(size_t)y1 * Ws + x0] += g * (double)s.w_sw;
                    if (xin && yin) gp[(size_t)y1 * Ws + x1] += g * (double)s.w_se;
                }
            }
            /* border clip: the grid gradient is zero when the un-clipped coordinate is on or outside the
             * border (torch clip_coordinates_set_grad); un-normalise then normalise: (Ws-1)/2 * 2/(Ws-1) */
            if (!(s.ixu > 0.0f && s.ixu < (float)(Ws - 1))) gix = 0.0;
            if (!(s.iyu > 0.0f && s.iyu < (float)(Hs - 1))) giy = 0.0;
            double g_u = gix * ((double)(Ws - 1) / 2.0) * (2.0 / (double)(Ws - 1));
            double g_v = giy * ((double)(Hs - 1) / 2.0) * (2.0 / (double)(Hs - 1));
            double z = pr.z;
            double g_q[3];
            g_q[0] = g_u / z;
            g_q[1] = g_v / z;
            g_q[2] = (fabsf(pr.q[2]) < 1e-6f) ? 0.0 : -((double)pr.u * g_u + (double)pr.v * g_v) / z;
            double g_T[3], g_P[3];
            for (int k = 0; k < 3; ++k)
                g_T[k] = (double)K[0 + k] * g_q[0] + (double)K[4 + k] * g_q[1] + (double)K[8 + k] * g_q[2];
            for (int i = 0; i < 3; ++i)
                g_P[i] = (double)pose[i] * g_T[0] + (double)pose[4 + i] * g_T[1] + (double)pose[8 + i] * g_T[2];
            g_depth[p] = (float)((double)pr.c[0] * g_P[0] + (double)pr.c[1] * g_P[1] + (double)pr.c[2] * g_P[2]);
            for (int j = 0; j < 3; ++j) {
                gM[4 * j + 0] += g_T[j] * (double)pr.P[0];
                gM[4 * j + 1] += g_T[j] * (double)pr.P[1];
                gM[4 * j + 2] += g_T[j] * (double)pr.P[2];
                gM[4 * j + 3] += g_T[j];
            }
        }
    for (int i = 0; i < 12; ++i) g_pose[i] = (float)gM[i];
    if (gS) {
        for (size_t i = 0; i < (size_t)C * Hs * Ws; ++i) g_src[i] += (float)gS[i];
        free(gS);
    }
}

/* backward of a3 alone: g_out (C,H,W) -> g_grid (H,W,2) w.r.t. the PIXEL coords that were passed to
 * synthesize(), and g_src accumulated.  grid holds pixel coords (not yet normalised). */
DSO_API void dso_grid_sample_bwd(const float* src, int C, int Hs, int Ws, const float* grid_px, int H, int W,
                                 const float* g_out, float* g_grid, float* g_src) {
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            size_t p = (size_t)y * W + x;
            dso_samp_t s;
            sample_setup(grid_px[2 * p], grid_px[2 * p + 1], Hs, Ws, &s);
            int x0 = s.x0, y0 = s.y0, x1 = x0 + 1, y1 = y0 + 1;
            int xin = x1 < Ws, yin = y1 < Hs;
            double gix = 0.0, giy = 0.0;
            for (int c = 0; c < C; ++c) {
                const float* pl = src + (size_t)c * Hs * Ws;
                double g = (double)g_out[(size_t)c * H * W + p];
                double nw = pl[(size_t)y0 * Ws + x0];
                double ne = xin ? pl[(size_t)y0 * Ws + x1] : 0.0;
                double sw = yin ? pl[(size_t)y1 * Ws + x0] : 0.0;
                double se = (xin && yin) ? pl[(size_t)y1 * Ws + x1] : 0.0;
                gix += g * ((ne - nw) * (double)s.sy + (se - sw) * (double)s.ny);
                giy += g * ((sw - nw) * (double)s.ex + (se - ne) * (double)s.wx);
                if (g_src) {
                    float* gp = g_src + (size_t)c * Hs * Ws;
                    gp[(size_t)y0 * Ws + x0] += (float)(g * (double)s.w_nw);
                    if (xin) gp[(size_t)y0 * Ws + x1] += (float)(g * (double)s.w_ne);
                    if (yin) gp[(size_t)y1 * Ws + x0] += (float)(g * (double)s.w_sw);
                    if (xin && yin) gp[(size_t)y1 * Ws + x1] += (float)(g * (double)s.w_se);
                }
            }
            if (!(s.ixu > 0.0f && s.ixu < (float)(Ws - 1))) gix = 0.0;
            if (!(s.iyu > 0.0f && s.iyu < (float)(Hs - 1))) giy = 0.0;
            g_grid[2 * p + 0] = (float)(gix * ((double)(Ws - 1) / 2.0) * (2.0 / (double)(Ws - 1)));
            g_grid[2 * p + 1] = (float)(giy * ((double)(Hs - 1) / 2.0) * (2.0 / (double)(Hs - 1)));
        }
}

/* ------------------------------------------------------------------------------------------ */
/* a10  metrics.regularization_term  (metrics.py:75-97)                                        */
/* ------------------------------------------------------------------------------------------ */
/* torch.gradient, spacing 1: central difference inside, one-sided at the two edges (metrics.py:83-84) */
static inline float grad1d(const float* f, int i, int L, size_t stride) {
    if (L == 1) return 0.f;
    if (i == 0) return f[stride] - f[0];
    if (i == L - 1) return f[(size_t)(L - 1) * stride] - f[(size_t)(L - 2) * stride];
    return (f[(size_t)(i + 1) * stride] - f[(size_t)(i - 1) * stride]) / 2.0f;
}

/* depths (N,1,H,W) (already divided by MAX_DEPTH by the caller, train.py:83), targets (N,C,H,W).
 * returns the scalar; g_depths (N,1,H,W), when non-NULL, receives d(term)/d(depths) * g_scale. */
DSO_API double dso_regularization_term(const float* depths, const float* targets, int N, int C, int H, int W,
                                       double g_scale, float* g_depths) {
    size_t HW = (size_t)H * W;
    double total = 0.0;
    double* acc = g_depths ? (double*)calloc((size_t)N * HW, sizeof(double)) : NULL;
    for (int n = 0; n < N; ++n) {
        const float* D = depths + (size_t)n * HW;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                size_t p = (size_t)y * W + x;
                float ddx = grad1d(D + (size_t)y * W, x, W, 1);
                float ddy = grad1d(D + x, y, H, (size_t)W);
                float adx = fabsf(ddx), ady = fabsf(ddy);
                float accum = 0.f;
                double wx = 0.0, wy = 0.0;
                for (int c = 0; c < C; ++c) {
                    const float* I = targets + ((size_t)n * C + c) * HW;
                    float ex = expf(fabsf(grad1d(I + (size_t)y * W, x, W, 1)));
                    float ey = expf(fabsf(grad1d(I + x, y, H, (size_t)W)));
                    float t = adx / ex + ady / ey;
                    accum = (c == 0) ? t : accum + t;
                    wx += 1.0 / (double)ex;
                    wy += 1.0 / (double)ey;
                }
                total += (double)(accum / (float)C);
                if (acc) {
                    double g = g_scale / ((double)N * (double)HW) / (double)C;
                    double gdx = g * wx * (ddx > 0.f ? 1.0 : (ddx < 0.f ? -1.0 : 0.0));
                    double gdy = g * wy * (ddy > 0.f ? 1.0 : (ddy < 0.f ? -1.0 : 0.0));
                    double* A = acc + (size_t)n * HW;
                    if (W > 1) {
                        if (x == 0) { A[p + 1] += gdx; A[p] -= gdx; }
                        else if (x == W - 1) { A[p] += gdx; A[p - 1] -= gdx; }
                        else { A[p + 1] += gdx / 2.0; A[p - 1] -= gdx / 2.0; }
                    }
                    if (H > 1) {
                        if (y == 0) { A[p + W] += gdy; A[p] -= gdy; }
                        else if (y == H - 1) { A[p] += gdy; A[p - W] -= gdy; }
                        else { A[p + W] += gdy / 2.0; A[p - W] -= gdy / 2.0; }
                    }
                }
            }
    }
    if (acc) {
        for (size_t i = 0; i < (size_t)N * HW; ++i) g_depths[i] = (float)acc[i];
        free(acc);
    }
    return total / ((double)N * (double)HW);
}

/* ------------------------------------------------------------------------------------------ */
/* a11-a15  depth metrics  (metrics.py:100-153; test.py:78-81)                                 */
/* ------------------------------------------------------------------------------------------ */
/* pred, gt (B,1,H,W) flattened to (B, HW).  out[0..3] = rmse, rmsle, abs_rel, sq_rel exactly as the
 * reference defines them (masked-out pixels count as zeros in a mean over ALL H*W pixels, then mean
 * over B).  out[4..6] = delta<1.25, <1.25^2, <1.25^3 with the same all-pixel denominator (NOT in the
 * reference: parity unpinned), out[7] = number of valid (gt>0) pixels summed over B. */
/* per image: numbers of valid pixels under the three thresholds, number of valid pixels (integers; (B,4)) */
DSO_API void dso_depth_metric_counts(const float* pred, const float* gt, int B, size_t HW, long long* counts) {
    for (int b = 0; b < B; ++b) {
        long long c[4] = {0, 0, 0, 0};
        for (size_t i = 0; i < HW; ++i) {
            float p = pred[(size_t)b * HW + i], g = gt[(size_t)b * HW + i];
            if (!(g > 0.0f)) continue;
            float r = fmaxf(p / g, g / p);
            c[0] += r < 1.25f;
            c[1] += r < 1.5625f;
            c[2] += r < 1.953125f;
            c[3] += 1;
        }
        for (int k = 0; k < 4; ++k) counts[(size_t)b * 4 + k] = c[k];
    }
}

/* eps: the offset inside the two logarithms of rmsle (metrics.py:114,122; a Python float added to fp32 tensors, i.e.
 * rounded to fp32 first); the reference's default is 1e-6 */
DSO_API void dso_depth_metrics_eps(const float* pred, const float* gt, int B, size_t HW, float eps, double* out) {
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = 0; b < B; ++b) {
        double s_sq = 0, s_lg = 0, s_ar = 0, s_sr = 0, d1 = 0, d2 = 0, d3 = 0, nv = 0;
        for (size_t i = 0; i < HW; ++i) {
            float p = pred[(size_t)b * HW + i], g = gt[(size_t)b * HW + i];
            if (!(g > 0.0f)) continue;
            float df = p - g;
            float sq = df * df;
            float lg = logf(p + eps) - logf(g + eps);
            s_sq += (double)sq;
            s_lg += (double)(lg * lg);
            s_ar += (double)(fabsf(df) / g);
            s_sr += (double)(sq / g);
            float r = fmaxf(p / g, g / p);
            d1 += r < 1.25f;
            d2 += r < 1.5625f;
            d3 += r < 1.953125f;
            nv += 1.0;
        }
        acc[0] += sqrt(s_sq / (double)HW);
        acc[1] += sqrt(s_lg / (double)HW);
        acc[2] += s_ar / (double)HW;
        acc[3] += s_sr / (double)HW;
        acc[4] += d1 / (double)HW;
        acc[5] += d2 / (double)HW;
        acc[6] += d3 / (double)HW;
        acc[7] += nv;
    }
    for (int k = 0; k < 7; ++k) out[k] = acc[k] / (double)B;
    out[7] = acc[7];
}

DSO_API void dso_depth_metrics(const float* pred, const float* gt, int B, size_t HW, double* out) {
    dso_depth_metrics_eps(pred, gt, B, HW, 1e-6f, out);
}

/* ------------------------------------------------------------------------------------------ */
/* a4-a6  plane sweep -> cost volume -> soft-argmin depth                                      */
/* (synthesis.py:57-80, metrics.py:5-18, synthesis.py:119-138; callers train.py:49-59)         */
/* ------------------------------------------------------------------------------------------ */
/* src (C,Hs,Ws); depths[D] are the plane depths (double, as constants.py:19; each becomes a constant
 * fp32 depth map, synthesis.py:69); out (D,C,H,W). */
DSO_API void dso_synthesize_from_depths(const float* K, const float* Kinv, const float* pose, const double* depths,
                                        int D, const float* src, int C, int Hs, int Ws, int H, int W, float* out) {
    for (int k = 0; k < D; ++k) {
        float d = (float)depths[k];
        float* O = out + (size_t)k * C * H * W;
#pragma omp parallel for schedule(static)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                size_t p = (size_t)y * W + x;
                dso_proj_t pr;
                dso_samp_t s;
                project_pixel(K, Kinv, pose, (float)x, (float)y, d, &pr);
                sample_setup(pr.u, pr.v, Hs, Ws, &s);
                for (int c = 0; c < C; ++c)
                    O[(size_t)c * H * W + p] = sample_plane(src + (size_t)c * Hs * Ws, Hs, Ws, &s);
            }
    }
}

/* target (C,H,W), synth (D,C,H,W) -> cost (D,H,W) = mean_c |t - s|  (metrics.py:18) */
DSO_API void dso_cost_volume(const float* target, const float* synth, int D, int C, int H, int W, float* cost) {
    size_t HW = (size_t)H * W;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < D; ++k)
        for (size_t p = 0; p < HW; ++p) {
            float acc = 0.f;
            for (int c = 0; c < C; ++c) {
                float a = fabsf(target[(size_t)c * HW + p] - synth[((size_t)k * C + c) * HW + p]);
                acc = (c == 0) ? a : acc + a;
            }
            cost[(size_t)k * HW + p] = acc / (float)C;
        }
}

/* cost (N,D,H,W) -> depth (N,1,H,W): softmin over D of exp(c)^2-1, expected plane index m,
 * SID: exp((m+1)*ln2/D)-1, else m/D + 1/D   (synthesis.py:127-138) */
DSO_API void dso_depth_from_costs(const float* cost, int N, int D, size_t HW, int sid, float* depth) {
    const float ln2 = logf(2.0f);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n)
        for (size_t p = 0; p < HW; ++p) {
            const float* c = cost + (size_t)n * D * HW + p;
            /* softmin(x) = softmax(-x): subtract the max of -x, i.e. use the min of x */
            float mn = INFINITY;
            for (int k = 0; k < D; ++k) {
                float e = expf(c[(size_t)k * HW]);
                float v = e * e - 1.0f;
                mn = fminf(mn, v);
            }
            double den = 0.0, num = 0.0;
            for (int k = 0; k < D; ++k) {
                float e = expf(c[(size_t)k * HW]);
                float v = e * e - 1.0f;
                float w = expf(-(v - mn));
                den += (double)w;
                num += (double)w * (double)k;
            }
            float m = (float)(num / den);
            depth[(size_t)n * HW + p] = sid ? (expf((m + 1.0f) * ln2 / (float)D) - 1.0f) : (m / (float)D + 1.0f / (float)D);
        }
}

/* ---- the two resizes around the plane sweep ---------------------------------------------------------------
 * functional.interpolate(x, size=(Hout,Wout), mode="bilinear") of train.py:55 and mode="bicubic" of train.py:60:
 * align_corners=False, size given (so the scale is in/out in fp32), no antialias.  Source index
 * fma(scale, dst+0.5, -0.5) (torch's compiled kernels contract it; the golden vectors show it: separate rounding
 * moves the weights by an ulp of the index, 2e-6 in the result); bilinear clamps it at 0 and the upper neighbour at the last row/column; bicubic floors it,
 * clamps the four tap indices, and weights them with the Keys kernel, A = -0.75. */
DSO_API void dso_resize_bilinear(const float* in, int N, int Hin, int Win, int Hout, int Wout, float* out) {
    const float sy = (float)Hin / (float)Hout, sx = (float)Win / (float)Wout;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < Hout; ++oy) {
            const float ry = fmaxf(fmaf(sy, (float)oy + 0.5f, -0.5f), 0.f);
            const int y0 = (int)ry < Hin - 1 ? (int)ry : Hin - 1, y1 = y0 + 1 < Hin - 1 ? y0 + 1 : Hin - 1;
            const float ly = fminf(fmaxf(ry - (float)y0, 0.f), 1.f);
            for (int ox = 0; ox < Wout; ++ox) {
                const float rx = fmaxf(fmaf(sx, (float)ox + 0.5f, -0.5f), 0.f);
                const int x0 = (int)rx < Win - 1 ? (int)rx : Win - 1, x1 = x0 + 1 < Win - 1 ? x0 + 1 : Win - 1;
                const float lx = fminf(fmaxf(rx - (float)x0, 0.f), 1.f);
                const float* p = in + (size_t)n * Hin * Win;
                const float top = fmaf(1.f - lx, p[y0 * Win + x0], lx * p[y0 * Win + x1]);
                const float bot = fmaf(1.f - lx, p[y1 * Win + x0], lx * p[y1 * Win + x1]);
                out[((size_t)n * Hout + oy) * Wout + ox] = fmaf(1.f - ly, top, ly * bot);
            }
        }
}

static inline void cubic_coefs(float t, float c[4]) {
    const float A = -0.75f, x0 = t + 1.f, x2 = 1.f - t, x3 = 2.f - t;
    c[0] = ((A * x0 - 5.f * A) * x0 + 8.f * A) * x0 - 4.f * A;
    c[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
    c[2] = ((A + 2.f) * x2 - (A + 3.f)) * x2 * x2 + 1.f;
    c[3] = ((A * x3 - 5.f * A) * x3 + 8.f * A) * x3 - 4.f * A;
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

DSO_API void dso_resize_bicubic(const float* in, int N, int Hin, int Win, int Hout, int Wout, float* out) {
    const float sy = (float)Hin / (float)Hout, sx = (float)Win / (float)Wout;
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < Hout; ++oy) {
            const float ry = fmaf(sy, (float)oy + 0.5f, -0.5f), fy = floorf(ry);
            float cy[4];
            cubic_coefs(fminf(fmaxf(ry - fy, 0.f), 1.f), cy);
            for (int ox = 0; ox < Wout; ++ox) {
                const float rx = fmaf(sx, (float)ox + 0.5f, -0.5f), fx = floorf(rx);
                float cx[4];
                cubic_coefs(fminf(fmaxf(rx - fx, 0.f), 1.f), cx);
                const float* p = in + (size_t)n * Hin * Win;
                float acc = 0.f;
                for (int j = 0; j < 4; ++j) {
                    const float* row = p + (size_t)clampi((int)fy - 1 + j, 0, Hin - 1) * Win;
                    float r = 0.f;
                    for (int i = 0; i < 4; ++i) r += cx[i] * row[clampi((int)fx - 1 + i, 0, Win - 1)];
                    acc += cy[j] * r;
                }
                out[((size_t)n * Hout + oy) * Wout + ox] = acc;
            }
        }
}


/* ------------------------------------------------------------------------------------------ */
/* f.4  the resize of the loaders (dataset.py:51,110): tf.resize(img, (192, 640)) on a PIL     */
/* image = PIL's Image.resize(BILINEAR).  The arithmetic lives in Pillow (libImaging/Resample.c, */
/* absent from /root/reference; Pillow 12.2.0 in this image): a separable triangle filter whose   */
/* support grows with the down-scale factor (antialiasing), coefficients normalised in double,    */
/* rounded to 22-bit fixed point, each pass accumulated in int32 from 1 << 21 and clipped to      */
/* 0..255; the horizontal pass comes first and its result is rounded to uint8 before the vertical */
/* one.  Integer work: pinned bit-exact against PIL itself (tests/test_oracle_golden.py) and      */
/* against the frames the golden fixtures hold.                                                    */
/* ------------------------------------------------------------------------------------------ */
#define DSO_PIL_PRECISION_BITS (32 - 8 - 2)

static double dso_pil_triangle(double x) {
    if (x < 0.0) x = -x;
    return x < 1.0 ? 1.0 - x : 0.0;
}

/* coefficients of output sample xx along an axis of in_size -> out_size: first tap *xmin, count returned (<= ksize) */
static int dso_pil_coeffs(int in_size, int out_size, int xx, int* xmin_out, int* kk, int ksize) {
    double scale = (double)in_size / out_size, filterscale = scale < 1.0 ? 1.0 : scale;
    double support = 1.0 * filterscale, center = (xx + 0.5) * scale, ss = 1.0 / filterscale, ww = 0.0;
    double w[64];
    int xmin = (int)(center - support + 0.5), xmax = (int)(center + support + 0.5), x;
    if (xmin < 0) xmin = 0;
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    if (xmax > ksize) xmax = ksize;
    for (x = 0; x < xmax; ++x) {
        w[x] = dso_pil_triangle((x + xmin - center + 0.5) * ss);
        ww += w[x];
    }
    for (x = 0; x < xmax; ++x) {
        double k = ww != 0.0 ? w[x] / ww : w[x];
        kk[x] = k < 0 ? (int)(-0.5 + k * (1 << DSO_PIL_PRECISION_BITS)) : (int)(0.5 + k * (1 << DSO_PIL_PRECISION_BITS));
    }
    *xmin_out = xmin;
    return xmax;
}

static unsigned char dso_pil_clip8(int v) {
    v >>= DSO_PIL_PRECISION_BITS;   /* arithmetic shift, as Pillow's lookup table index */
    return (unsigned char)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

/* in (Hin,Win,C) uint8 interleaved -> out (Hout,Wout,C) uint8; tmp (Hin,Wout,C) uint8 scratch; returns 0, or -1 if the
 * filter would need more than 64 taps (down-scaling by more than ~31x) */
DSO_API int dso_pil_resize_bilinear_u8(const unsigned char* in, int Hin, int Win, int C, int Hout, int Wout,
                                       unsigned char* tmp, unsigned char* out) {
    double sx = (double)Win / Wout, sy = (double)Hin / Hout;
    int kx = (int)ceil(sx < 1.0 ? 1.0 : sx) * 2 + 1, ky = (int)ceil(sy < 1.0 ? 1.0 : sy) * 2 + 1;
    if (kx > 64 || ky > 64) return -1;
    for (int xx = 0; xx < Wout; ++xx) {
        int kk[64], xmin, n = dso_pil_coeffs(Win, Wout, xx, &xmin, kk, kx);
        for (int y = 0; y < Hin; ++y)
            for (int c = 0; c < C; ++c) {
                int acc = 1 << (DSO_PIL_PRECISION_BITS - 1);
                for (int x = 0; x < n; ++x) acc += (int)in[((size_t)y * Win + xmin + x) * C + c] * kk[x];
                tmp[((size_t)y * Wout + xx) * C + c] = dso_pil_clip8(acc);
            }
    }
    for (int yy = 0; yy < Hout; ++yy) {
        int kk[64], ymin, n = dso_pil_coeffs(Hin, Hout, yy, &ymin, kk, ky);
        for (int x = 0; x < Wout * C; ++x) {
            int acc = 1 << (DSO_PIL_PRECISION_BITS - 1);
            for (int y = 0; y < n; ++y) acc += (int)tmp[(size_t)(ymin + y) * Wout * C + x] * kk[y];
            out[(size_t)yy * Wout * C + x] = dso_pil_clip8(acc);
        }
    }
    return 0;
}
