// ds_step.hip -- a loss step (train.py:72-88 for B targets and its autograd) as two calls, as one call, and with the
// smoothness term of train.py:89 as one call.
//
// Host code and the small finalize kernels of the one-call forms.  A caller that goes through a slow foreign-function
// layer (ctypes, cgo, JNI) pays for every call; these entry points issue the launches of the calls they replace, on the
// caller's stream.
//
// Measured and dropped: cutting the batch into parts on side streams inside each call (fork and join by events).  The
// memory-bound warp kernels of one part do overlap the VALU-bound loss kernels of another, but a join after every half
// step leaves only two kernels per part to overlap and costs more than it hides (B=12, 192x640: 0.138 ms per step on one
// stream, 0.176 ms with 3 parts joined every step; bench.py --streams 3 --join).  bench.py's --streams pipelines
// independent sub-batches ACROSS steps instead, which needs no support here.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "ds_common.h"
#include "ds_internal.h"
#include "depthsynth.h"

namespace {

// the two fixed-order finalisations of a step in ONE launch: blocks [0,B) the per-image loss means, blocks [B, B+N) the
// pose gradients (a dependent launch costs ~4 us of drain + dispatch at this size, whatever it computes)
__global__ __launch_bounds__(64) void step_finalize_kernel(const float* __restrict__ loss_partials, int waves_per_image,
                                                           double inv_hw, float* __restrict__ sums, int B,
                                                           const float* __restrict__ pose_partials, int nblk,
                                                           float* __restrict__ g_pose) {
    const int i = blockIdx.x;
    if (i < B) ds::loss_finalize_wave(loss_partials, i, waves_per_image, inv_hw, sums, threadIdx.x);
    else ds::pose_finalize_wave(pose_partials, i - B, nblk, g_pose, threadIdx.x);
}

// The end of ds_train_step_fwdbwd: reg = sum of the smoothness kernel's per-wave partials / count (fixed order, double),
// total = sum_b sums[b][0] + scale * reg (train.py:88-89).  One workgroup.
constexpr int kTrainFinThreads = 1024;
__global__ __launch_bounds__(kTrainFinThreads) void train_finalize_kernel(const float* __restrict__ partials, int n,
                                                                          double inv_count, float scale,
                                                                          const float* __restrict__ sums, int B,
                                                                          float* __restrict__ reg, float* __restrict__ total) {
    __shared__ double lds[kTrainFinThreads / 64];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += kTrainFinThreads) a += (double)partials[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kTrainFinThreads / 64; ++w) t += lds[w];
        const float r = (float)(t * inv_count);
        float loss = 0.f;
        for (int b = 0; b < B; ++b) loss += sums[2 * b];
        reg[0] = r;
        total[0] = loss + scale * r;
    }
}

inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

inline size_t loss_partials_bytes(int B, int H, int W) {  // either form of the loss kernels
    const size_t a = ds_fused_loss_workspace_bytes(B, H, W), b = ds::fused_loss_fb_workspace_bytes(B, H, W);
    return align256(a > b ? a : b);
}

}  // namespace

extern "C" {

size_t ds_loss_step_workspace_bytes(int B, int S, int H, int W) {
    if (B <= 0 || S <= 0 || H < 2 || W < 2) return 0;
    // the loss kernels' partial sums and the warp backward's pose partials are never live at the same time
    const size_t a = ds_fused_loss_workspace_bytes(B, H, W), b = ds_warp_bwd_workspace_bytes(B * S, H, W);
    return a > b ? a : b;
}

int ds_loss_step_fwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                     const float* target, int B, int S, int H, int W, float* warped, float* out_loss, float* idt_loss,
                     uint8_t* aux, float* sums, void* workspace, size_t workspace_bytes, void* stream) {
    if (B <= 0 || S <= 0 || (long long)B * S > 65535) return DS_ERR_INVALID_ARG;
    if (S <= 3) {
        // the warp and the identity map from one launch (the loss workgroups' waves warp), then the warped groups' loss
        // against that map: the same maps as below, 60 instead of 68 us at B=12 S=2 192x640
        if (!sums) return DS_ERR_INVALID_ARG;
        hipStream_t st = (hipStream_t)stream;
        int rc = ds::warp_and_identity(K, Kinv, pose, depth, src, target, B, S, H, W, warped, idt_loss, st);
        if (rc != DS_OK) return rc;
        int wpi = 0;
        rc = ds::fused_loss_fwd_idm_partials(warped, target, idt_loss, B, S, H, W, out_loss, aux, workspace, workspace_bytes, &wpi,
                                             st);
        if (rc != DS_OK) return rc;
        hipLaunchKernelGGL(step_finalize_kernel, dim3(B), dim3(64), 0, st, (const float*)workspace, wpi, 1.0 / ((double)H * W),
                           sums, B, (const float*)nullptr, 0, (float*)nullptr);
        DS_CHECK_LAUNCH();
        return DS_OK;
    }
    int rc = ds_warp_fwd(K, Kinv, pose, depth, src, B * S, 3, H, W, H, W, warped, stream);
    if (rc != DS_OK) return rc;
    return ds_fused_loss_fwd(warped, src, target, B, S, H, W, out_loss, idt_loss, aux, sums, workspace, workspace_bytes,
                             stream);
}

int ds_loss_step_bwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                     const float* target, const float* warped, const float* idt_loss, const uint8_t* aux,
                     const float* g_sums, int B, int S, int H, int W, float* g_warped, float* g_depth, float* g_pose,
                     void* workspace, size_t workspace_bytes, void* stream) {
    if (B <= 0 || S <= 0 || (long long)B * S > 65535) return DS_ERR_INVALID_ARG;
    // validate the second call's arguments before the first one launches anything
    if (!K || !Kinv || !pose || !depth || !src || !g_depth || !g_pose) return DS_ERR_INVALID_ARG;
    if (!workspace || workspace_bytes < ds_warp_bwd_workspace_bytes(B * S, H, W)) return DS_ERR_WORKSPACE;
    int rc = ds_fused_loss_bwd(warped, target, idt_loss, aux, g_sums, B, S, H, W, g_warped, stream);
    if (rc != DS_OK) return rc;
    return ds_warp_bwd(K, Kinv, pose, depth, src, g_warped, B * S, 3, H, W, H, W, g_depth, g_pose, nullptr, workspace,
                       workspace_bytes, stream);
}

int ds_warp_identity_fwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                         const float* target, int B, int S, int H, int W, float* warped, float* idt_loss, void* stream) {
    if (B <= 0 || S <= 0 || (long long)B * S > 65535) return DS_ERR_INVALID_ARG;
    return ds::warp_and_identity(K, Kinv, pose, depth, src, target, B, S, H, W, warped, idt_loss, (hipStream_t)stream);
}

size_t ds_fused_loss_fb_warp_bwd_workspace_bytes(int B, int S, int H, int W) {
    return ds::fused_loss_fbw_supported(S) ? ds::fused_loss_fbw_workspace_bytes(B, S, H, W) : 0;
}

int ds_fused_loss_fb_warp_bwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                              const float* warped, const float* target, const float* idt_loss, const float* g_sums, int B,
                              int S, int H, int W, float* out_loss, uint8_t* aux, float* sums, float* g_warped,
                              float* g_depth, float* g_pose, void* workspace, size_t workspace_bytes, void* stream) {
    if (!sums || !g_pose) return DS_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    int wpi = 0;
    float* pose_partials = nullptr;
    int rc = ds::fused_loss_fbw_partials(K, Kinv, pose, depth, src, warped, target, idt_loss, g_sums, B, S, H, W, out_loss, aux,
                                         g_warped, g_depth, workspace, workspace_bytes, &wpi, &pose_partials, st);
    if (rc != DS_OK) return rc;
    hipLaunchKernelGGL(step_finalize_kernel, dim3(B + B * S), dim3(64), 0, st, (const float*)workspace, wpi,
                       1.0 / ((double)H * W), sums, B, (const float*)pose_partials, wpi, g_pose);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

size_t ds_loss_step_fwdbwd_workspace_bytes(int B, int S, int H, int W) {
    if (B <= 0 || S <= 0 || H < 2 || W < 2) return 0;
    // both sets of partial sums stay live until the single finalize launch at the end
    const size_t a = loss_partials_bytes(B, H, W) + ds_warp_bwd_workspace_bytes(B * S, H, W);
    const size_t b = ds::fused_loss_fbw_supported(S) ? ds::fused_loss_fbw_workspace_bytes(B, S, H, W) : 0;
    return a > b ? a : b;
}

size_t ds_loss_step_fwdbwd_image_workspace_bytes(int B, int S, int H, int W) {
    if (B <= 0 || S <= 0 || H < 2 || W < 2) return 0;
    // with g_src the loss and the warp's autograd are separate kernels: loss partials, then the warp backward's workspace
    // with the staging of the image gradient
    return loss_partials_bytes(B, H, W) + ds_warp_bwd_image_workspace_bytes(B * S, 3, H, W, H, W);
}

int ds_loss_step_fwdbwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                        const float* target, const float* g_sums, int B, int S, int H, int W, float* warped,
                        float* out_loss, float* idt_loss, uint8_t* aux, float* sums, float* g_warped, float* g_depth,
                        float* g_pose, float* g_src, void* workspace, size_t workspace_bytes, void* stream) {
    if (B <= 0 || S <= 0 || (long long)B * S > 65535 || H < 2 || W < 2) return DS_ERR_INVALID_ARG;
    if (!K || !Kinv || !pose || !depth || !src || !target || !warped || !out_loss || !idt_loss || !aux || !sums ||
        !g_depth || !g_pose)
        return DS_ERR_INVALID_ARG;
    if (S > DS_MAX_SOURCES) return DS_ERR_UNSUPPORTED;
    // d loss / d warped only crosses memory where the loss and the warp's autograd are separate kernels
    const bool one_kernel_backward = ds::fused_loss_fbw_supported(S) && !g_src;
    if (!g_warped && !one_kernel_backward) return DS_ERR_INVALID_ARG;
    if (!workspace || ((uintptr_t)workspace & 15) ||
        workspace_bytes < (g_src ? ds_loss_step_fwdbwd_image_workspace_bytes(B, S, H, W)
                                 : ds_loss_step_fwdbwd_workspace_bytes(B, S, H, W)))
        return DS_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const size_t n_loss = loss_partials_bytes(B, H, W);
    void* ws_loss = workspace;
    void* ws_warp = (char*)workspace + n_loss;
    int rc, wpi = 0, nblk = 0;
    if (ds::fused_loss_fb_supported(S) && g_src) {
        // with the image gradient: warp + identity map, fused loss forward+backward (d loss / d warped to memory), then
        // the warp backward kernel that scatters
        rc = ds::warp_and_identity(K, Kinv, pose, depth, src, target, B, S, H, W, warped, idt_loss, st);
        if (rc != DS_OK) return rc;
        rc = ds::fused_loss_fb_partials(warped, target, idt_loss, g_sums, B, S, H, W, out_loss, aux, g_warped, ws_loss, n_loss,
                                        &wpi, st);
        if (rc != DS_OK) return rc;
    } else if (one_kernel_backward) {
        // S <= 3 (S = 3, the reference's SEQ_LEN - 1, at 168 VGPRs = three waves per SIMD: still 9-13 % faster than the unfused
        // backward, profiles/r02_kb4).  The warp and the identity map (un-warped sources against the target: nothing of it
        // depends on depth or pose) share one launch; then the forward and the backward of the warped groups in ONE kernel:
        // the backward's window sums are the forward's
        rc = ds::warp_and_identity(K, Kinv, pose, depth, src, target, B, S, H, W, warped, idt_loss, st);
        if (rc != DS_OK) return rc;
        // ... and that kernel's fourth wave runs the warp's autograd on the rows of d loss / d warped as they are finished:
        // they never reach memory (g_warped stays untouched), and the step is three dependent launches and a finalize
        float* pose_partials = nullptr;
        rc = ds::fused_loss_fbw_partials(K, Kinv, pose, depth, src, warped, target, idt_loss, g_sums, B, S, H, W, out_loss, aux,
                                         nullptr, g_depth, workspace, workspace_bytes, &wpi, &pose_partials, st);
        if (rc != DS_OK) return rc;
        hipLaunchKernelGGL(step_finalize_kernel, dim3(B + B * S), dim3(64), 0, st, (const float*)workspace, wpi,
                           1.0 / ((double)H * W), sums, B, (const float*)pose_partials, wpi, g_pose);
        DS_CHECK_LAUNCH();
        return DS_OK;
    } else if (S == 3) {
        // three sources with the image gradient: the forward as for S <= 2 (warp + identity map in one launch, then the
        // warped groups against it), the backward unfused
        rc = ds::warp_and_identity(K, Kinv, pose, depth, src, target, B, S, H, W, warped, idt_loss, st);
        if (rc != DS_OK) return rc;
        rc = ds::fused_loss_fwd_idm_partials(warped, target, idt_loss, B, S, H, W, out_loss, aux, ws_loss, n_loss, &wpi, st);
        if (rc != DS_OK) return rc;
        rc = ds_fused_loss_bwd(warped, target, idt_loss, aux, g_sums, B, S, H, W, g_warped, stream);
        if (rc != DS_OK) return rc;
    } else {
        rc = ds_warp_fwd(K, Kinv, pose, depth, src, B * S, 3, H, W, H, W, warped, stream);
        if (rc != DS_OK) return rc;
        rc = ds::fused_loss_fwd_partials(warped, src, target, B, S, H, W, out_loss, idt_loss, aux, ws_loss, n_loss, &wpi, st);
        if (rc != DS_OK) return rc;
        rc = ds_fused_loss_bwd(warped, target, idt_loss, aux, g_sums, B, S, H, W, g_warped, stream);
        if (rc != DS_OK) return rc;
    }
    rc = ds::warp_bwd_partials(K, Kinv, pose, depth, src, g_warped, B * S, 3, H, W, H, W, g_depth, g_src, ws_warp,
                               workspace_bytes - n_loss, &nblk, st);
    if (rc != DS_OK) return rc;
    hipLaunchKernelGGL(step_finalize_kernel, dim3(B + B * S), dim3(64), 0, st, (const float*)ws_loss, wpi,
                       1.0 / ((double)H * W), sums, B, (const float*)ws_warp, nblk, g_pose);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

size_t ds_train_step_fwdbwd_workspace_bytes(int B, int S, int H, int W) {
    const size_t a = ds_loss_step_fwdbwd_workspace_bytes(B, S, H, W);
    if (a == 0) return 0;
    return align256(a) + ds::smooth_fb_partials_count(B, S, H, W) * sizeof(float);
}

int ds_train_step_fwdbwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                         const float* target, float smooth_scale, int B, int S, int H, int W, float* warped,
                         float* out_loss, float* idt_loss, uint8_t* aux, float* sums, float* reg, float* total,
                         float* g_warped, float* g_depth, float* g_pose, void* workspace, size_t workspace_bytes,
                         void* stream) {
    if (!reg || !total) return DS_ERR_INVALID_ARG;
    if (B <= 0 || S <= 0 || H < 2 || W < 2) return DS_ERR_INVALID_ARG;
    if (!workspace || ((uintptr_t)workspace & 15) || workspace_bytes < ds_train_step_fwdbwd_workspace_bytes(B, S, H, W))
        return DS_ERR_WORKSPACE;
    const size_t n_step = align256(ds_loss_step_fwdbwd_workspace_bytes(B, S, H, W));
    int rc = ds_loss_step_fwdbwd(K, Kinv, pose, depth, src, target, nullptr, B, S, H, W, warped, out_loss, idt_loss, aux, sums,
                                 g_warped, g_depth, g_pose, nullptr, workspace, n_step, stream);
    if (rc != DS_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    float* partials = (float*)((char*)workspace + n_step);
    int n = 0;
    rc = ds::smooth_fb_partials(depth, target, smooth_scale, B, S, H, W, g_depth, partials, &n, st);
    if (rc != DS_OK) return rc;
    hipLaunchKernelGGL(train_finalize_kernel, dim3(1), dim3(kTrainFinThreads), 0, st, (const float*)partials, n,
                       1.0 / ((double)B * S * H * W * 3.0), smooth_scale, (const float*)sums, B, reg, total);
    DS_CHECK_LAUNCH();
    return DS_OK;
}

}  // extern "C"
