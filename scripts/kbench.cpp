// kbench: times the four C-ABI calls of one fused reprojection-loss step for a given libdepthsynth build.
//   scripts/kbench <lib.so> B S H W iters
// Build: hipcc -O2 -o scripts/kbench scripts/kbench.cpp -ldl
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <random>
#include <string>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef int (*warp_fwd_t)(const float*, const float*, const float*, const float*, const float*, int, int, int, int, int, int, float*, void*);
typedef int (*warp_bwd_t)(const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, int, int, int, float*, float*, float*, void*, size_t, void*);
typedef size_t (*ws3_t)(int, int, int);
typedef int (*loss_fwd_t)(const float*, const float*, const float*, int, int, int, int, float*, float*, uint8_t*, float*, void*, size_t, void*);
typedef int (*loss_bwd_t)(const float*, const float*, const float*, const uint8_t*, const float*, int, int, int, int, float*, void*);

static float* dev_rand(size_t n, float lo, float hi, unsigned seed) {
    std::vector<float> h(n);
    std::mt19937 g(seed);
    std::uniform_real_distribution<float> d(lo, hi);
    for (auto& v : h) v = d(g);
    float* p; CK(hipMalloc(&p, n * 4)); CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
    return p;
}

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: kbench lib B S H W iters\n"); return 2; }
    void* h = dlopen(argv[1], RTLD_NOW);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    int B = atoi(argv[2]), S = atoi(argv[3]), H = atoi(argv[4]), W = atoi(argv[5]), iters = atoi(argv[6]);
    auto warp_fwd = (warp_fwd_t)dlsym(h, "ds_warp_fwd");
    auto warp_bwd = (warp_bwd_t)dlsym(h, "ds_warp_bwd");
    auto loss_fwd = (loss_fwd_t)dlsym(h, "ds_fused_loss_fwd");
    auto loss_bwd = (loss_bwd_t)dlsym(h, "ds_fused_loss_bwd");
    auto ws_loss = (ws3_t)dlsym(h, "ds_fused_loss_workspace_bytes");
    auto ws_warp = (ws3_t)dlsym(h, "ds_warp_bwd_workspace_bytes");
    const int N = B * S; const size_t HW = (size_t)H * W;
    float* src = dev_rand(N * 3 * HW, 0, 1, 1);
    float* tgt = dev_rand(B * 3 * HW, 0, 1, 2);
    float* depth = dev_rand(N * HW, 1, 80, 3);
    // K = normalised KITTI intrinsics scaled by W,H; Kinv its inverse; pose = identity + small translation
    float fx = 0.58f * W, fy = 1.92f * H, cx = 0.5f * W, cy = 0.5f * H;
    float Kh[16] = {fx, 0, cx, 0, 0, fy, cy, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    float Kih[16] = {1 / fx, 0, -cx / fx, 0, 0, 1 / fy, -cy / fy, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    std::vector<float> ph(N * 12, 0.f);
    for (int n = 0; n < N; ++n) { ph[n * 12 + 0] = ph[n * 12 + 5] = ph[n * 12 + 10] = 1.f; ph[n * 12 + 3] = 0.05f * (n % 3); ph[n * 12 + 11] = 0.1f; }
    float *K, *Kinv, *pose; CK(hipMalloc(&K, 64)); CK(hipMalloc(&Kinv, 64)); CK(hipMalloc(&pose, N * 48));
    CK(hipMemcpy(K, Kh, 64, hipMemcpyHostToDevice)); CK(hipMemcpy(Kinv, Kih, 64, hipMemcpyHostToDevice));
    CK(hipMemcpy(pose, ph.data(), N * 48, hipMemcpyHostToDevice));
    float *warped, *out, *idt, *sums, *g_sums, *g_warped, *g_depth, *g_pose; uint8_t* aux; void *w1, *w2;
    CK(hipMalloc(&warped, N * 3 * HW * 4)); CK(hipMalloc(&out, B * HW * 4)); CK(hipMalloc(&idt, B * HW * 4));
    CK(hipMalloc(&aux, B * HW)); CK(hipMalloc(&sums, B * 8)); CK(hipMalloc(&g_warped, N * 3 * HW * 4));
    CK(hipMalloc(&g_depth, N * HW * 4)); CK(hipMalloc(&g_pose, N * 48));
    std::vector<float> ones(B, 1.f); CK(hipMalloc(&g_sums, B * 4)); CK(hipMemcpy(g_sums, ones.data(), B * 4, hipMemcpyHostToDevice));
    size_t n1 = ws_loss(B, H, W), n2 = ws_warp(N, H, W);
    float* g_src = nullptr;  // DS_KBENCH_IMG=1: also form the source-image gradient
    typedef size_t (*ws6_t)(int, int, int, int, int, int);
    if (getenv("DS_KBENCH_IMG")) {
        CK(hipMalloc(&g_src, N * 3 * HW * 4)); CK(hipMemset(g_src, 0, N * 3 * HW * 4));
        if (auto wsi = (ws6_t)dlsym(h, "ds_warp_bwd_image_workspace_bytes")) n2 = wsi(N, 3, H, W, H, W);   // round 3: staging
    }
    CK(hipMalloc(&w1, n1 ? n1 : 4)); CK(hipMalloc(&w2, n2 ? n2 : 4));
    hipStream_t st = 0;
    hipEvent_t e[5]; for (auto& x : e) CK(hipEventCreate(&x));
    double t[4] = {0, 0, 0, 0};
    for (int it = -3; it < iters; ++it) {
        CK(hipEventRecord(e[0], st));
        int rc = warp_fwd(K, Kinv, pose, depth, src, N, 3, H, W, H, W, warped, st); if (rc) { fprintf(stderr, "warp_fwd rc=%d\n", rc); return 1; }
        CK(hipEventRecord(e[1], st));
        rc = loss_fwd(warped, src, tgt, B, S, H, W, out, idt, aux, sums, w1, n1, st); if (rc) { fprintf(stderr, "loss_fwd rc=%d\n", rc); return 1; }
        CK(hipEventRecord(e[2], st));
        rc = loss_bwd(warped, tgt, idt, aux, g_sums, B, S, H, W, g_warped, st); if (rc) { fprintf(stderr, "loss_bwd rc=%d\n", rc); return 1; }
        CK(hipEventRecord(e[3], st));
        rc = warp_bwd(K, Kinv, pose, depth, src, g_warped, N, 3, H, W, H, W, g_depth, g_pose, g_src, w2, n2, st); if (rc) { fprintf(stderr, "warp_bwd rc=%d\n", rc); return 1; }
        CK(hipEventRecord(e[4], st));
        CK(hipStreamSynchronize(st));
        if (it >= 0) for (int k = 0; k < 4; ++k) { float ms; CK(hipEventElapsedTime(&ms, e[k], e[k + 1])); t[k] += ms; }
    }
    // ---- experiment: the batch split into `parts` sub-batches, each on its own stream (VALU-bound loss kernels of one
    // part can overlap the gather-bound warp kernels of another); wall time of the whole step via host clock
    if (argc > 7) {
        const int parts = atoi(argv[7]);
        std::vector<hipStream_t> ss(parts);
        for (auto& x : ss) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        const int Bp = B / parts, Np = Bp * S;
        size_t n1p = ws_loss(Bp, H, W), n2p = ws_warp(Np, H, W);
        std::vector<void*> w1p(parts), w2p(parts);
        for (int p = 0; p < parts; ++p) { CK(hipMalloc(&w1p[p], n1p ? n1p : 4)); CK(hipMalloc(&w2p[p], n2p ? n2p : 4)); }
        auto step = [&]() {
            for (int p = 0; p < parts; ++p) {
                const size_t o = (size_t)p * Np, ob = (size_t)p * Bp;
                warp_fwd(K, Kinv, pose + o * 12, depth + o * HW, src + o * 3 * HW, Np, 3, H, W, H, W, warped + o * 3 * HW, ss[p]);
            }
            for (int p = 0; p < parts; ++p) {
                const size_t o = (size_t)p * Np, ob = (size_t)p * Bp;
                loss_fwd(warped + o * 3 * HW, src + o * 3 * HW, tgt + ob * 3 * HW, Bp, S, H, W, out + ob * HW, idt + ob * HW, aux + ob * HW, sums + ob * 2, w1p[p], n1p, ss[p]);
            }
            for (int p = 0; p < parts; ++p) {
                const size_t o = (size_t)p * Np, ob = (size_t)p * Bp;
                loss_bwd(warped + o * 3 * HW, tgt + ob * 3 * HW, idt + ob * HW, aux + ob * HW, g_sums + ob, Bp, S, H, W, g_warped + o * 3 * HW, ss[p]);
            }
            for (int p = 0; p < parts; ++p) {
                const size_t o = (size_t)p * Np;
                warp_bwd(K, Kinv, pose + o * 12, depth + o * HW, src + o * 3 * HW, g_warped + o * 3 * HW, Np, 3, H, W, H, W, g_depth + o * HW, g_pose + o * 12, nullptr, w2p[p], n2p, ss[p]);
            }
        };
        for (int i = 0; i < 5; ++i) step();
        CK(hipDeviceSynchronize());
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < iters; ++i) step();
        CK(hipDeviceSynchronize());
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / iters;
        printf("   split into %d parts on %d streams: %.4f ms per step\n", parts, parts, ms);
    }
    // ---- the forward loss as two launches of the G = S kernel (warped groups, identity groups) instead of one G = 2S launch
    typedef int (*rl_fwd_t)(const float*, const float*, long long, long long, int, int, int, int, float*, uint8_t*, void*);
    if (auto rl_fwd = (rl_fwd_t)dlsym(h, "ds_reprojection_loss_fwd")) {
        double tw = 0, ti = 0;
        for (int it = -3; it < iters; ++it) {
            CK(hipEventRecord(e[0], st));
            rl_fwd(warped, tgt, 3 * (long long)HW, 0, B, S, H, W, out, aux, st);
            CK(hipEventRecord(e[1], st));
            rl_fwd(src, tgt, 3 * (long long)HW, 0, B, S, H, W, idt, nullptr, st);
            CK(hipEventRecord(e[2], st));
            CK(hipStreamSynchronize(st));
            if (it >= 0) { float ms; CK(hipEventElapsedTime(&ms, e[0], e[1])); tw += ms; CK(hipEventElapsedTime(&ms, e[1], e[2])); ti += ms; }
        }
        printf("   reprojection_loss_fwd alone: warped groups %.4f ms, identity groups %.4f ms\n", tw / iters, ti / iters);
    }
    // ---- overlap probe: warp_fwd (memory-bound) and the identity groups' loss (VALU-bound) are independent.  Sequential
    // on one stream against concurrent on two streams (fork/join by events, every iteration), wall clock.
    if (auto rl_fwd = (rl_fwd_t)dlsym(h, "ds_reprojection_loss_fwd")) {
        hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        hipEvent_t ef, ej; CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
        auto wall = [&](auto&& body) {
            for (int i = 0; i < 5; ++i) body();
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < iters; ++i) body();
            CK(hipDeviceSynchronize());
            return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / iters;
        };
        double seq = wall([&] {
            warp_fwd(K, Kinv, pose, depth, src, N, 3, H, W, H, W, warped, s1);
            rl_fwd(src, tgt, 3 * (long long)HW, 0, B, S, H, W, idt, nullptr, s1);
        });
        double conc = wall([&] {
            CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0));
            warp_fwd(K, Kinv, pose, depth, src, N, 3, H, W, H, W, warped, s1);
            rl_fwd(src, tgt, 3 * (long long)HW, 0, B, S, H, W, idt, nullptr, s2);
            CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0));
        });
        double w_only = wall([&] { warp_fwd(K, Kinv, pose, depth, src, N, 3, H, W, H, W, warped, s1); });
        double i_only = wall([&] { rl_fwd(src, tgt, 3 * (long long)HW, 0, B, S, H, W, idt, nullptr, s1); });
        double full = wall([&] {
            warp_fwd(K, Kinv, pose, depth, src, N, 3, H, W, H, W, warped, s1);
            loss_fwd(warped, src, tgt, B, S, H, W, out, idt, aux, sums, w1, n1, s1);
            loss_bwd(warped, tgt, idt, aux, g_sums, B, S, H, W, g_warped, s1);
            warp_bwd(K, Kinv, pose, depth, src, g_warped, N, 3, H, W, H, W, g_depth, g_pose, g_src, w2, n2, s1);
        });
        typedef int (*step_t)(const float*, const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, int,
                              float*, float*, float*, uint8_t*, float*, float*, float*, float*, float*, void*, size_t, void*);
        typedef size_t (*ws4_t)(int, int, int, int);
        auto stepf = (step_t)dlsym(h, "ds_loss_step_fwdbwd");
        auto stepws = (ws4_t)dlsym(h, "ds_loss_step_fwdbwd_workspace_bytes");
        if (stepf && stepws) {
            size_t n3 = stepws(B, S, H, W);
            if (g_src) if (auto wsi = (ws4_t)dlsym(h, "ds_loss_step_fwdbwd_image_workspace_bytes")) n3 = wsi(B, S, H, W);
            void* w3; CK(hipMalloc(&w3, n3));
            double one = wall([&] {
                int rc = stepf(K, Kinv, pose, depth, src, tgt, nullptr, B, S, H, W, warped, out, idt, aux, sums, g_warped, g_depth, g_pose, g_src, w3, n3, s1);
                if (rc) { fprintf(stderr, "ds_loss_step_fwdbwd rc=%d\n", rc); exit(1); }
            });
            printf("   ds_loss_step_fwdbwd (one call, merged finalize): %.4f ms per step = %.2f Gpix/s\n", one, (double)N * HW / one / 1e6);
            // the same call captured into a HIP graph (one graph launch per step; 10 steps per graph as a second variant)
            for (int per : {1, 10}) {
                hipGraph_t gr; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s1, hipStreamCaptureModeGlobal));
                for (int k = 0; k < per; ++k)
                    stepf(K, Kinv, pose, depth, src, tgt, nullptr, B, S, H, W, warped, out, idt, aux, sums, g_warped, g_depth, g_pose, g_src, w3, n3, s1);
                CK(hipStreamEndCapture(s1, &gr));
                CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
                double gt = wall([&] { CK(hipGraphLaunch(ge, s1)); }) / per;
                printf("   ds_loss_step_fwdbwd replayed from a HIP graph (%d steps per graph): %.4f ms per step\n", per, gt);
                CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(gr));
            }
        }
        typedef int (*fb_t)(const float*, const float*, const float*, const float*, int, int, int, int, float*, uint8_t*, float*, float*, void*, size_t, void*);
        if (auto fb = (fb_t)dlsym(h, "ds_fused_loss_fb")) {
            auto fbws = (ws3_t)dlsym(h, "ds_fused_loss_fb_workspace_bytes");
            size_t n4 = fbws(B, H, W); void* w4; CK(hipMalloc(&w4, n4 ? n4 : 4));
            double t_fb = wall([&] { fb(warped, tgt, idt, nullptr, B, S, H, W, out, aux, sums, g_warped, w4, n4, s1); });
            double t_fwd = wall([&] { loss_fwd(warped, src, tgt, B, S, H, W, out, idt, aux, sums, w1, n1, s1); });
            double t_bwd = wall([&] { loss_bwd(warped, tgt, idt, aux, g_sums, B, S, H, W, g_warped, s1); });
            double t_wb = wall([&] { warp_bwd(K, Kinv, pose, depth, src, g_warped, N, 3, H, W, H, W, g_depth, g_pose, g_src, w2, n2, s1); });
            printf("   wall clock, each alone back to back: fused_loss_fb %.4f (+finalize), fused_loss_fwd %.4f (+finalize), fused_loss_bwd %.4f, warp_bwd %.4f (+finalize) ms\n", t_fb, t_fwd, t_bwd, t_wb);
        }
        printf("   wall clock, back to back: warp_fwd %.4f, identity loss %.4f, both sequential %.4f, on two streams %.4f; whole step %.4f ms\n",
               w_only, i_only, seq, conc, full);
    }
    std::vector<float> hs(B * 2); CK(hipMemcpy(hs.data(), sums, B * 8, hipMemcpyDeviceToHost));
    printf("%-40s B=%d S=%d %dx%d  warp_fwd %.4f  loss_fwd %.4f  loss_bwd %.4f  warp_bwd %.4f ms  (sum %.4f)  loss[0]=%.6f\n", argv[1], B, S, H, W,
           t[0] / iters, t[1] / iters, t[2] / iters, t[3] / iters, (t[0] + t[1] + t[2] + t[3]) / iters, hs[0]);
    return 0;
}
