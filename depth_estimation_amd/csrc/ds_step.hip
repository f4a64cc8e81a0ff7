// ds_step.hip -- the two halves of a loss step (train.py:72-88 for B targets and its autograd) as one call each.
//
// Host code only: no kernels here.  A caller that goes through a slow foreign-function layer (ctypes, cgo, JNI) pays for
// every call; these two entry points issue the same launches as the four calls they replace, on the caller's stream.
//
// Measured and dropped: cutting the batch into parts on side streams inside each call (fork and join by events).  The
// memory-bound warp kernels of one part do overlap the VALU-bound loss kernels of another, but a join after every half
// step leaves only two kernels per part to overlap and costs more than it hides (B=12, 192x640: 0.138 ms per step on one
// stream, 0.176 ms with 3 parts joined every step; bench.py --streams 3 --join).  bench.py's --streams pipelines
// independent sub-batches ACROSS steps instead, which needs no support here.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "depthsynth.h"

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

}  // extern "C"
