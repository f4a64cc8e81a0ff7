/*
 * depthsynth.h -- C ABI of libdepthsynth.so: gfx950 (MI355X) kernels for the view-synthesis /
 * reprojection-loss hot path of banaga-benjamin/Depth_Estimation.
 *
 * The reference has no FFI: its interface for this path is the set of Python functions in
 * synthesis.py and metrics.py, called from train.py:46-92, test.py:47-81 and predict.py:47-60.
 * Each entry point below names the reference function (file:line) it replaces.  The Python mirror of
 * those functions (depth_estimation_amd/synthesis.py, metrics.py) binds these symbols with ctypes;
 * INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous fp32 (or uint8 where stated) unless marked HOST;
 *   - images are planar NCHW, coordinate grids are (N,H,W,2) x-first -- the reference's layouts;
 *   - K and Kinv are the reference's 4x4 row-major intrinsics (only [:3,:3] is read, synthesis.py:100,112);
 *   - `stream` is a hipStream_t (NULL = default stream); calls are asynchronous, never synchronise and
 *     never allocate: the caller owns outputs and workspaces;
 *   - return value: 0 on success, <0 for an argument error (DS_ERR_*), >0 = hipError_t from a launch.
 */
#ifndef DEPTHSYNTH_H_
#define DEPTHSYNTH_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DS_OK 0
#define DS_ERR_INVALID_ARG (-1)
#define DS_ERR_UNSUPPORTED (-2)
#define DS_ERR_WORKSPACE   (-3)
#define DS_ERR_SCRATCH_LIMIT (-4) /* ds_*_noalloc: the input needs more than the built-in scratch array */

#define DS_MAX_SOURCES 4        /* S per launch of the loss kernels (the reference uses S = SEQ_LEN-1 = 3) */
#define DS_AUX_ARGMIN_MASK 0x1f /* aux byte: bits 0-4 argmin source, bits 5-7 reserved (0) */

int ds_version(void);
const char* ds_error_string(int code);

/* ---- synthesis.py ------------------------------------------------------------------------------ */

/* synthesis.construct_pose_batch (synthesis.py:6-33): vec (N,6) -> pose (N,3,4), and its autograd. */
int ds_pose_from_vec_fwd(const float* vec, int N, float* pose, void* stream);
int ds_pose_from_vec_bwd(const float* vec, const float* g_pose, int N, float* g_vec, void* stream);

/* synthesis.reproject_from_depth (synthesis.py:83-116), batched over N (depth map, pose) pairs.
 * pose (N,3,4), depth (N,H,W) -> coords (N,H,W,2) in pixel units. */
int ds_reproject(const float* K, const float* Kinv, const float* pose, const float* depth, int N, int H, int W,
                 float* coords, void* stream);

/* autograd of ds_reproject: g_coords (N,H,W,2) -> g_depth (N,H,W), g_pose (N,3,4); workspace sized by
 * ds_warp_bwd_workspace_bytes(N,H,W). */
int ds_reproject_bwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* g_coords,
                     int N, int H, int W, float* g_depth, float* g_pose, void* workspace, size_t workspace_bytes,
                     void* stream);

/* synthesis.synthesize (synthesis.py:36-54).  grid (N,H,W,2) holds pixel coords on entry and the
 * NORMALISED coords on return (the reference mutates its argument, :50-51; normalisation uses the
 * SOURCE dims Hs,Ws, :49).  src (N,C,Hs,Ws) -> out (N,C,H,W). */
int ds_grid_sample_fwd(const float* src, float* grid, int N, int C, int Hs, int Ws, int H, int W, float* out,
                       void* stream);

/* autograd of synthesis.synthesize: grid_norm is the grid as ds_grid_sample_fwd left it.
 * g_grid (N,H,W,2) is d/d(pixel coords).  g_src (N,C,Hs,Ws) may be NULL; when given it must be
 * zero-filled by the caller (atomic accumulation). */
int ds_grid_sample_bwd(const float* src, const float* grid_norm, const float* g_out, int N, int C, int Hs, int Ws,
                       int H, int W, float* g_grid, float* g_src, void* stream);

/* fused reproject_from_depth + synthesize (train.py:72-77): warped (N,C,H,W). */
int ds_warp_fwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src, int N,
                int C, int Hs, int Ws, int H, int W, float* warped, void* stream);

/* workspace for the pose-gradient partials of ds_warp_bwd and ds_reproject_bwd */
size_t ds_warp_bwd_workspace_bytes(int N, int H, int W);
/* workspace of ds_warp_bwd WITH the source-image gradient (g_src != NULL): for C == 3 the pose partials plus the staging of
 * the scatter -- per wave of the launch (64 columns x 8 rows of pixels) a record, 16 row slots of 66 x 3 floats and a point
 * list: about 113 bytes per warped pixel (333 MB for 24 warps of 192 x 640), sized for the worst case (every finished
 * window row without a slot); only what a warp actually stages is touched (~16 bytes per pixel for a smooth warp, 24 when
 * every pixel leaves its wave on its own).  16-byte aligned, need not be initialised.  Other C: = ds_warp_bwd_workspace_bytes.
 * Limits of the C == 3 path: Hs <= 32767, Ws <= 65534 (DS_ERR_UNSUPPORTED beyond). */
size_t ds_warp_bwd_image_workspace_bytes(int N, int C, int Hs, int Ws, int H, int W);

/* autograd of ds_warp_fwd: g_depth (N,H,W), g_pose (N,3,4) (deterministic two-stage reduction, no atomics), g_src
 * (N,C,Hs,Ws) optional (NULL to skip).  C == 3: g_src is formed without atomics -- every texel is written once, the sum of
 * the staged contributions in a fixed order, bit-identical from run to run; it need not be cleared, and the workspace must
 * have ds_warp_bwd_image_workspace_bytes.  Other channel counts: float atomics onto a g_src the caller has zero-filled
 * (order-dependent in the last bits). */
int ds_warp_bwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                const float* g_warped, int N, int C, int Hs, int Ws, int H, int W, float* g_depth, float* g_pose,
                float* g_src, void* workspace, size_t workspace_bytes, void* stream);

/* synthesis.synthesize_from_depths (synthesis.py:57-80): D constant-depth planes (HOST double[D], as
 * constants.DEPTHS), one pose (3,4), src (C,Hs,Ws) -> out (D,C,H,W). */
int ds_synthesize_from_depths(const float* K, const float* Kinv, const float* pose, const double* depths_host, int D,
                              const float* src, int C, int Hs, int Ws, int H, int W, float* out, void* stream);

/* synthesis.depth_from_costs (synthesis.py:119-138): cost (N,D,H,W) -> depth (N,1,H,W). */
int ds_depth_from_costs(const float* cost, int N, int D, int H, int W, int sid, float* depth, void* stream);

/* fused synthesize_from_depths + metrics.cost_volumes + depth_from_costs (train.py:49-59) for NS
 * sources of one target: pose (NS,3,4), src (NS,C,Hs,Ws), target_small (C,H,W) -> depth (NS,1,H,W).
 * The (D,C,H,W) images and the (NS,D,H,W) cost volume are never materialised.  D <= 128. */
int ds_plane_sweep_depth(const float* K, const float* Kinv, const float* pose, const double* depths_host, int D,
                         const float* src, const float* target_small, int NS, int C, int Hs, int Ws, int H, int W,
                         int sid, float* depth, void* stream);

/* The same for a batch, one launch: the loop over the batch in train.py:46-63.  pose (B*S,3,4), src (B*S,C,Hs,Ws)
 * with the S sources of target b at rows b*S..b*S+S-1, targets_small (B,C,H,W) -> depth (B*S,1,H,W).  B*S <= 65535. */
int ds_plane_sweep_depth_batch(const float* K, const float* Kinv, const float* pose, const double* depths_host, int D,
                               const float* src, const float* targets_small, int B, int S, int C, int Hs, int Ws, int H,
                               int W, int sid, float* depth, void* stream);

/* functional.interpolate(x, size=(Hout,Wout), mode="bilinear") of train.py:55 (target -> cost-volume resolution) and
 * mode="bicubic" of train.py:60 (candidate depth -> image resolution); align_corners=False, no antialias.
 * in (N,Hin,Win) planes -> out (N,Hout,Wout). */
int ds_resize_bilinear(const float* in, int N, int Hin, int Win, int Hout, int Wout, float* out, void* stream);
int ds_resize_bicubic(const float* in, int N, int Hin, int Win, int Hout, int Wout, float* out, void* stream);

/* ---- metrics.py -------------------------------------------------------------------------------- */

/* metrics.cost_volumes (metrics.py:5-18): target (C,H,W), synth (D,C,H,W) -> cost (D,H,W). */
int ds_cost_volume(const float* target, const float* synth, int D, int C, int H, int W, float* cost, void* stream);

/* metrics.reprojection_error (metrics.py:21-51): nplanes = N*C planes of (H,W). */
int ds_ssim_fwd(const float* preds, const float* targets, int nplanes, int H, int W, float* out, void* stream);
/* its autograd w.r.t. preds (targets carry no gradient anywhere in the reference's callers). */
int ds_ssim_bwd(const float* preds, const float* targets, const float* g_out, int nplanes, int H, int W,
                float* g_preds, void* stream);

/* metrics.reprojection_loss (metrics.py:54-72), batched: preds (B,S,3,H,W); targets addressed as
 * targets + b*tgt_stride_b + s*tgt_stride_s (elements; stride_s = 0 for train.py:80's stacked target).
 * out (B,H,W) = min over S (first index on ties); aux (B,H,W) uint8 (see DS_AUX_ARGMIN_MASK). */
int ds_reprojection_loss_fwd(const float* preds, const float* targets, long long tgt_stride_b, long long tgt_stride_s,
                             int B, int S, int H, int W, float* out, uint8_t* aux, void* stream);
/* autograd w.r.t. preds: g_out (B,H,W) -> g_preds (B,S,3,H,W) (every element written). */
int ds_reprojection_loss_bwd(const float* preds, const float* targets, long long tgt_stride_b, long long tgt_stride_s,
                             const float* g_out, const uint8_t* aux, int B, int S, int H, int W, float* g_preds,
                             void* stream);

/* loss assembly of train.py:80-92 for B targets with S sources each: warped, src (B,S,3,H,W),
 * target (B,3,H,W) -> out_loss, idt_loss (B,H,W), aux (B,H,W) u8, sums (B,2) fp32:
 * sums[b][0] = mean(out/exp(idt+1e-6)) (train.py:88), sums[b][1] = mean(out) (train.py:92). */
size_t ds_fused_loss_workspace_bytes(int B, int H, int W);
int ds_fused_loss_fwd(const float* warped, const float* src, const float* target, int B, int S, int H, int W,
                      float* out_loss, float* idt_loss, uint8_t* aux, float* sums, void* workspace,
                      size_t workspace_bytes, void* stream);
/* autograd of sums[b][0] w.r.t. warped: g_sums (B) = d(loss)/d(sums[b][0]) (NULL = all ones) -> g_warped (B,S,3,H,W). */
int ds_fused_loss_bwd(const float* warped, const float* target, const float* idt_loss, const uint8_t* aux,
                      const float* g_sums, int B, int S, int H, int W, float* g_warped, void* stream);

/* ds_warp_fwd (C = 3, source size = grid size) and the identity loss map of train.py:85 -- idt_loss (B,H,W) =
 * metrics.reprojection_loss(src_imgs, target_imgs) per target, i.e. ds_reprojection_loss_fwd(src, target) -- in ONE
 * launch: the identity term depends on neither depth nor pose, so its workgroups share the launch with the warp's. */
int ds_warp_identity_fwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                         const float* target, int B, int S, int H, int W, float* warped, float* idt_loss, void* stream);

/* Forward and backward of the warped groups in ONE kernel, given the identity map idt_loss (B,H,W) =
 * ds_reprojection_loss_fwd(src, target): out_loss, aux, sums as ds_fused_loss_fwd, g_warped as ds_fused_loss_bwd (g_sums
 * (B) or NULL = ones), bit for bit, without the backward's recomputation of the forward's window sums.  S <= 2
 * (DS_ERR_UNSUPPORTED beyond: use the pair above). */
size_t ds_fused_loss_fb_workspace_bytes(int B, int H, int W);
int ds_fused_loss_fb(const float* warped, const float* target, const float* idt_loss, const float* g_sums, int B, int S, int H,
                     int W, float* out_loss, uint8_t* aux, float* sums, float* g_warped, void* workspace,
                     size_t workspace_bytes, void* stream);

/* ds_fused_loss_fb with the warp's autograd in the same kernel (its fourth wave takes the rows of d loss / d warped from
 * the loss waves through LDS): everything of a step that follows ds_warp_identity_fwd.  Inputs of ds_fused_loss_fb and
 * ds_warp_bwd (source size = grid size); outputs out_loss, aux, sums, g_depth (B,S,H,W), g_pose (B,S,3,4) and, if
 * g_warped is not NULL, d loss / d warped too (otherwise it never reaches memory).  S <= 3 (DS_ERR_UNSUPPORTED beyond). */
size_t ds_fused_loss_fb_warp_bwd_workspace_bytes(int B, int S, int H, int W);
int ds_fused_loss_fb_warp_bwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                              const float* warped, const float* target, const float* idt_loss, const float* g_sums, int B,
                              int S, int H, int W, float* out_loss, uint8_t* aux, float* sums, float* g_warped,
                              float* g_depth, float* g_pose, void* workspace, size_t workspace_bytes, void* stream);

/* The two halves of a loss step in one call each, on the caller's stream: the results of ds_warp_fwd + ds_fused_loss_fwd,
 * and of ds_fused_loss_bwd + ds_warp_bwd (train.py:72-88 and its autograd for B targets; source and grid sizes equal).
 * Arguments as in the four calls; g_warped (B,S,3,H,W) is scratch for the backward.  Half the calls for a host layer that
 * pays per call; for S <= 3 the forward half runs as ds_warp_identity_fwd followed by the loss of the warped groups against
 * that identity map (the same maps bit for bit; the means from another order of the partial sums). */
size_t ds_loss_step_workspace_bytes(int B, int S, int H, int W);
int ds_loss_step_fwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                     const float* target, int B, int S, int H, int W, float* warped, float* out_loss, float* idt_loss,
                     uint8_t* aux, float* sums, void* workspace, size_t workspace_bytes, void* stream);
int ds_loss_step_bwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                     const float* target, const float* warped, const float* idt_loss, const uint8_t* aux,
                     const float* g_sums, int B, int S, int H, int W, float* g_warped, float* g_depth, float* g_pose,
                     void* workspace, size_t workspace_bytes, void* stream);

/* The whole loss step -- train.py:72-88 for B targets AND its autograd -- in one call: forward and backward launches
 * back to back on the caller's stream, the two fixed-order finalisations (per-target means, pose gradients) merged into
 * one launch at the end (S <= 3 without g_src: three launches -- warp + identity loss, loss forward + adjoint + the
 * warp's autograd, finalize; otherwise five).  The loss of train.py:88 is linear in every
 * sums[b][0], so the gradients can be formed before the caller's backward runs: g_sums (B) = d(loss)/d(sums[b][0]), or
 * NULL for a unit upstream gradient -- what loss.backward() passes for the reference's summed loss (train.py:88,103); a
 * caller with another upstream factor scales g_depth and g_pose by it.  Outputs as in the four calls it replaces;
 * g_warped (B,S,3,H,W) is scratch; for S <= 3 without g_src the gradient w.r.t. the warped images passes from the loss
 * kernel's waves to the warp's autograd inside one kernel, the buffer is left untouched and may be NULL (with g_src, or
 * S = 4, NULL is DS_ERR_INVALID_ARG).  g_src (B,S,3,H,W), optional (NULL
 * to skip): the gradient w.r.t. the source images through the warp (every texel written, no atomics, bit-identical
 * from run to run: see ds_warp_bwd).  workspace 16-byte aligned, sized by ds_loss_step_fwdbwd_workspace_bytes, or by
 * ds_loss_step_fwdbwd_image_workspace_bytes when g_src is given. */
size_t ds_loss_step_fwdbwd_workspace_bytes(int B, int S, int H, int W);
size_t ds_loss_step_fwdbwd_image_workspace_bytes(int B, int S, int H, int W);
int ds_loss_step_fwdbwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                        const float* target, const float* g_sums, int B, int S, int H, int W, float* warped,
                        float* out_loss, float* idt_loss, uint8_t* aux, float* sums, float* g_warped, float* g_depth,
                        float* g_pose, float* g_src, void* workspace, size_t workspace_bytes, void* stream);

/* train.py:72-89 for B targets and its autograd in one call: ds_loss_step_fwdbwd (unit upstream gradient, no image
 * gradient) plus the smoothness term of train.py:89 on the S depth maps of every target against that target
 * (metrics.regularization_term, metrics.py:75-97) with its gradient accumulated onto g_depth -- forward value and
 * gradient from one kernel -- and a last launch that forms reg (1) = regularization_term and total (1) = sum_b sums[b][0]
 * + smooth_scale * reg.  g_depth, g_pose = d total / d depth, d total / d pose; a caller with another upstream factor
 * scales them.  Everything else as ds_loss_step_fwdbwd (g_warped may be NULL for S <= 3); workspace from
 * ds_train_step_fwdbwd_workspace_bytes. */
size_t ds_train_step_fwdbwd_workspace_bytes(int B, int S, int H, int W);
int ds_train_step_fwdbwd(const float* K, const float* Kinv, const float* pose, const float* depth, const float* src,
                         const float* target, float smooth_scale, int B, int S, int H, int W, float* warped,
                         float* out_loss, float* idt_loss, uint8_t* aux, float* sums, float* reg, float* total,
                         float* g_warped, float* g_depth, float* g_pose, void* workspace, size_t workspace_bytes,
                         void* stream);

/* metrics.regularization_term (metrics.py:75-97): depths (N,1,H,W), targets (N,C,H,W) -> value (1)
 * fp32 on device; workspace from ds_smooth_workspace_bytes. */
size_t ds_smooth_workspace_bytes(int N, int H, int W);
int ds_smooth_fwd(const float* depths, const float* targets, int N, int C, int H, int W, float* value,
                  void* workspace, size_t workspace_bytes, void* stream);
/* g_value (1, device) -> g_depths (N,1,H,W) */
int ds_smooth_bwd(const float* depths, const float* targets, const float* g_value, int N, int C, int H, int W,
                  float* g_depths, void* stream);
/* The same with the target stack of train.py:80-83 left implicit: depths (B*S,1,H,W), the S maps of target b at rows
 * b*S..b*S+S-1, targets (B,C,H,W) -> mean over all B*S maps (= the reference's value when called per sample with
 * equal-sized samples, see fused.train_step_loss).  workspace from ds_smooth_workspace_bytes(B*S, H, W). */
int ds_smooth_batch_fwd(const float* depths, const float* targets, int B, int S, int C, int H, int W, float* value,
                        void* workspace, size_t workspace_bytes, void* stream);
/* accumulate != 0: g_depths += the gradient (the depth maps usually carry the photometric gradient already). */
int ds_smooth_batch_bwd(const float* depths, const float* targets, const float* g_value, int B, int S, int C, int H, int W,
                        int accumulate, float* g_depths, void* stream);

/* metrics.rmse / rmsle / abs_rel / sq_rel (metrics.py:100-153; test.py:78-81) in ONE pass over
 * pred, gt (B,1,H,W): out (8) fp32 = rmse, rmsle, abs_rel, sq_rel, delta<1.25, <1.25^2, <1.25^3 (not in
 * the reference: same all-pixel denominator, parity unpinned), number of valid pixels.  counts (B,4) int64, optional
 * (NULL to skip): per image the exact numbers of valid pixels under the three thresholds and of valid (gt > 0) pixels,
 * so that callers can renormalise by the valid count.  workspace 16-byte aligned. */
size_t ds_metrics_workspace_bytes(int B, int H, int W);
int ds_depth_metrics(const float* pred, const float* gt, int B, int H, int W, float* out, long long* counts,
                     void* workspace, size_t workspace_bytes, void* stream);
/* The same with the eps argument of metrics.rmsle (metrics.py:114,122: log(p + eps) - log(g + eps); a Python float added to
 * an fp32 tensor, hence an fp32 value).  ds_depth_metrics is this call with the reference's default 1e-6. */
int ds_depth_metrics_eps(const float* pred, const float* gt, int B, int H, int W, float eps, float* out, long long* counts,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ds_depth_metrics without a caller workspace and without any allocation.  No counterpart in the reference: it exists
 * for callers that cannot allocate device memory around the call (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  The partial sums live in a DS_NOALLOC_SCRATCH_BYTES array inside the library's
 * device image (one copy per GPU, held until the process ends), so ds_metrics_workspace_bytes(B,H,W) must not exceed
 * DS_NOALLOC_SCRATCH_BYTES (B * ceil(H*W/2048) <= ~32700; 64 x 320 x 1024 needs 0.33 MB): larger inputs return
 * DS_ERR_SCRATCH_LIMIT.  Calls on one GPU share that array: calls that overlap in time on one GPU (different streams or
 * threads) race; order them.  The first call on a device looks the array's address up (hipGetSymbolAddress; no memory
 * is allocated), so make one call outside a capture first.  Same kernels and results as ds_depth_metrics. */
#define DS_NOALLOC_SCRATCH_BYTES (1u << 20)
int ds_depth_metrics_noalloc(const float* pred, const float* gt, int B, int H, int W, float* out, long long* counts,
                             void* stream);

/* Input pipeline to the device: torchvision.transforms.functional.to_tensor on the resized frame (dataset.py:51,110):
 * frames (N,H,W,3) uint8 interleaved -> out (N,3,H,W) fp32 = value / 255 (for frames resized on the host; see
 * ds_ingest_resize_u8 for the resize itself).
 * frames 4-byte aligned, out 16-byte aligned. */
int ds_ingest_u8(const uint8_t* frames, int N, int H, int W, float* out, void* stream);

/* The resize of the loaders on the device (dataset.py:51,110: tf.resize(img, (192, 640)) on a PIL image = PIL's
 * antialiased BILINEAR resize; Pillow's fixed-point arithmetic, bit-exact): raw frames (N,Hin,Win,3) uint8 interleaved ->
 * out_u8 (N,Hout,Wout,3) uint8 (what PIL returns) and / or out_f32 (N,3,Hout,Wout) fp32 = value / 255 (to_tensor of it);
 * either may be NULL.  workspace from ds_ingest_resize_workspace_bytes.  Down-scaling by more than ~31x per axis:
 * DS_ERR_UNSUPPORTED. */
size_t ds_ingest_resize_workspace_bytes(int N, int Hin, int Win, int Hout, int Wout);
int ds_ingest_resize_u8(const uint8_t* frames, int N, int Hin, int Win, int Hout, int Wout, uint8_t* out_u8, float* out_f32,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Ground-truth depth to the device (dataset.py:116-120): the resized 16-bit PNG samples (N,H,W) uint16 ->
 * out (N,1,H,W) fp32 = value / 2**16.  depth 4-byte aligned, out 8-byte aligned.  The reference holds no depth PNG
 * and torchvision is absent here: parity unpinned, checked against the direct statement value / 65536. */
int ds_ingest_depth_u16(const uint16_t* depth, int N, int H, int W, float* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEPTHSYNTH_H_ */
