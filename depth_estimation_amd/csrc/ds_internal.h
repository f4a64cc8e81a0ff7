// ds_internal.h -- what the translation units of libdepthsynth.so share with each other and with nobody else
// (hidden visibility: none of this is part of the C ABI in include/depthsynth.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <atomic>

#define DS_HIDDEN __attribute__((visibility("hidden")))

namespace ds {

// ---- fixed-order finalisation of partial sums (deterministic: one wave, strided double accumulation, xor tree) ----------
// per-image sums of the forward loss kernel's per-workgroup partials (ratio, out) -> means (2 floats)
__device__ __forceinline__ void loss_finalize_wave(const float* __restrict__ wave_partials, int b, int waves_per_image,
                                                   double inv_hw, float* __restrict__ sums, int lane) {
    double a0 = 0.0, a1 = 0.0;
    for (int i = lane; i < waves_per_image; i += 64) {
        a0 += (double)wave_partials[2 * ((size_t)b * waves_per_image + i) + 0];
        a1 += (double)wave_partials[2 * ((size_t)b * waves_per_image + i) + 1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a0 += __shfl_xor(a0, o, 64);
        a1 += __shfl_xor(a1, o, 64);
    }
    if (lane == 0) {
        sums[2 * b + 0] = (float)(a0 * inv_hw);
        sums[2 * b + 1] = (float)(a1 * inv_hw);
    }
}

// partials (N, nblk, 12) of the warp backward -> g_pose (N,12)
__device__ __forceinline__ void pose_finalize_wave(const float* __restrict__ partials, int n, int nblk,
                                                   float* __restrict__ g_pose, int lane) {
    double a[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) a[i] = 0.0;
    for (int b = lane; b < nblk; b += 64) {
        const float* q = partials + ((size_t)n * nblk + b) * 12;
#pragma unroll
        for (int i = 0; i < 12; ++i) a[i] += (double)q[i];
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) a[i] += __shfl_xor(a[i], o, 64);
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 12; ++i) g_pose[n * 12 + i] = (float)a[i];
    }
}

// ---- launchers without their finalize kernel (ds_step.hip merges the finalisations of a step into one launch) --------
// forward loss kernel of ds_fused_loss_fwd; *waves_per_image = partial-sum slots per image in `workspace`
DS_HIDDEN int fused_loss_fwd_partials(const float* warped, const float* src, const float* target, int B, int S, int H, int W,
                                      float* out_loss, float* idt_loss, uint8_t* aux, void* workspace,
                                      size_t workspace_bytes, int* waves_per_image, hipStream_t stream);
// ds_warp_fwd (C = 3, source size = grid size) and the identity loss map ds_reprojection_loss_fwd(src, target) in ONE launch
// the forward loss of the warped groups with the identity map given (photo_fwd_idm_kernel); partials as fused_loss_fwd_partials
DS_HIDDEN int fused_loss_fwd_idm_partials(const float* warped, const float* target, const float* idt_loss, int B, int S, int H,
                                          int W, float* out_loss, uint8_t* aux, void* workspace, size_t workspace_bytes,
                                          int* waves_per_image, hipStream_t stream);
DS_HIDDEN int warp_and_identity(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                                const float* target, int B, int S, int H, int W, float* warped, float* idt_loss,
                                hipStream_t stream);
// forward AND backward of the warped groups in one kernel, given the identity map (ds_photo.hip: photo_fb_kernel); S <= 2
DS_HIDDEN bool fused_loss_fb_supported(int S);
DS_HIDDEN bool fused_loss_fbw_supported(int S);   // with the warp's autograd in the fourth wave: also S = 3 (three waves per SIMD)
DS_HIDDEN size_t fused_loss_fb_workspace_bytes(int B, int H, int W);
DS_HIDDEN int fused_loss_fb_partials(const float* warped, const float* target, const float* idt_loss, const float* g_sums,
                                     int B, int S, int H, int W, float* out_loss, uint8_t* aux, float* g_warped,
                                     void* workspace, size_t workspace_bytes, int* waves_per_image, hipStream_t stream);
// the same kernel with the warp's autograd in its fourth wave: d loss / d warped goes from the channel waves to it through
// LDS (g_warped may be NULL: then it never reaches memory), g_depth and per-workgroup pose partials come out;
// *pose_partials (B*S, *waves_per_image, 12) lies inside `workspace`
DS_HIDDEN size_t fused_loss_fbw_workspace_bytes(int B, int S, int H, int W);
DS_HIDDEN int fused_loss_fbw_partials(const float* K, const float* Kinv, const float* pose, const float* depth,
                                      const float* src, const float* warped, const float* target, const float* idt_loss,
                                      const float* g_sums, int B, int S, int H, int W, float* out_loss, uint8_t* aux,
                                      float* g_warped, float* g_depth, void* workspace, size_t workspace_bytes,
                                      int* waves_per_image, float** pose_partials, hipStream_t stream);
// main kernel of ds_warp_bwd; *nblk = pose-gradient partials per image in `workspace`
DS_HIDDEN int warp_bwd_partials(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                                const float* g_warped, int N, int C, int Hs, int Ws, int H, int W, float* g_depth,
                                float* g_src, void* workspace, size_t workspace_bytes, int* nblk, hipStream_t stream);

// metrics.regularization_term and its gradient in one kernel (ds_misc.hip): g_depths (B,S,H,W) += scale * d reg / d depth;
// partials (smooth_fb_partials_count floats): per-wave sums whose total / (B*S*H*W*3) is reg.
DS_HIDDEN size_t smooth_fb_partials_count(int B, int S, int H, int W);
DS_HIDDEN int smooth_fb_partials(const float* depths, const float* targets, float scale, int B, int S, int H, int W,
                                 float* g_depths, float* partials, int* n_partials, hipStream_t stream);

// Compute units of the device that was current at the first call; cached for the process (std::atomic: concurrent first
// calls from two host threads may both query, both store the same value), 256 if the query fails.  Every launch geometry
// and workspace size derives from this one number, so all devices a process drives are assumed to be the same model
// (one node's MI355Xs are; INTEGRATION.md says so).  Results never depend on it, only how the work is cut.
inline int cu_count() {
    static std::atomic<int> cached{0};
    int n = cached.load(std::memory_order_relaxed);
    if (n == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        cached.store(cus, std::memory_order_relaxed);
        n = cus;
    }
    return n;
}

}  // namespace ds
