/* weclip_hip.h -- C ABI of libweclip_hip.so (MI355X / gfx950 HIP kernels for the WeCLIP
 * forward/CAM hot path).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'ed, or a torch CUDA tensor's data_ptr())
 *     unless the parameter name starts with `h_` (host array, read during the call);
 *   - tensors are dense row-major in the stated shape; inputs are never written;
 *   - `stream` is a hipStream_t (NULL = default stream); calls only enqueue work, never sync;
 *   - return value: 0 = ok, non-zero = error (WC_ERR_ARG 1: rejected argument, nothing was
 *     launched; WC_ERR_HIP 2: HIP runtime error); wc_last_error() gives the message
 *     (thread-local).  The Python host raises RuntimeError, like the stock torch ops the
 *     reference calls do.
 *   - no global state except thread-local error text; re-entrant per device.
 *
 * Each entry cites the reference interface it replaces (paths relative to the reference
 * repository dayae1204/WeCLIP-ViT-CoMer).
 */
#ifndef WECLIP_HIP_H
#define WECLIP_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ------------------------------------------------------------------------- */
int wc_version(void);
const char* wc_last_error(void);
int wc_device_count(void);
int wc_device_arch(int dev, char* buf, int buflen);

/* Per-kernel HIP-event timing of the dominant kernels (GEMMs, attention, PAR), for bench.py's roofline leg.
 * wc_prof_enable(n) clears the records and records one of every n instrumented launches on average (a fixed pseudo-random
 * pick, so that the sample does not alias with the step; n = 1: all; an event pair fences its launch, ~6 us), (0) stops.
 * The sweeps of a PAR.forward group share ONE pair (always recorded).  wc_prof_report synchronises the device and writes one
 * "kernel name \t launches \t total ms \t algorithmic work (flop or bytes) \t ms scaled by each record's sampling weight"
 * line per kernel. */
void wc_prof_enable(int stride);
/* Group tag appended to the kernel names of the launches that follow ("@vit_attn": in-projection, attention, head-mean
 * maps and out-projection of an encoder block, the north star's "ViT attention" group); NULL / "" clears it. */
void wc_prof_tag(const char* tag);
int wc_prof_report(char* buf, int cap);

/* ---- PAR: pixel-adaptive refinement -------------------------------------------------- */
/* WeCLIP_model/PAR.py:64-88 (`PAR.forward` up to `aff`, incl. get_dilated_neighbors :39-49 and
 * get_pos :51-62).  img (B,3,H,W) f32 -> aff (B, 8*n_dil, H, W) f32. */
int wc_par_affinity(const float* img, float* aff, int B, int H, int W,
                    const int* h_dilations, int n_dil, void* stream);
/* WeCLIP_model/PAR.py:88-91, one iteration: masks_out = sum_t aff_t * masks_in(nbr_t).
 * masks (B,C,H,W) f32; in and out must differ. */
int wc_par_iterate(const float* aff, const float* masks_in, float* masks_out, int B, int C,
                   int H, int W, const int* h_dilations, int n_dil, void* stream);
/* WeCLIP_model/PAR.py:64-92, whole `PAR.forward` for imgs already at mask resolution.
 * out/tmp: (B,C,H,W) f32 workspaces (result in out); aff_ws: min(group,B)*8*n_dil*H*ceil64(W) f32.
 * Images are swept in groups of `group` so a group's aff planes stay cache resident. */
int wc_par_forward(const float* img, const float* masks, float* out, float* tmp, float* aff_ws,
                   int B, int C, int H, int W, const int* h_dilations, int n_dil, int num_iter,
                   int group, void* stream);
/* Same sweep with the affinities stored as 16-bit fixed-point pairs + one fp32 scale per pixel between the iterations
 * (half the HBM bytes per sweep; |rounding error| <= max_t a_t * 7.7e-6 per weight, error-diffused so a pixel's weights
 * keep their sum): the `fast` precision mode.  6 dilations (48 taps) only; aff_ws as for wc_par_forward. */
int wc_par_forward_h(const float* img, const float* masks, float* out, float* tmp, float* aff_ws,
                   int B, int C, int H, int W, const int* h_dilations, int n_dil, int num_iter,
                   int group, void* stream);
/* WeCLIP_model/model_attn_aff_voc.py:49-57 (`_refine_cams` tail): labels = valid_key[argmax_c].
 * valid_key (B,C) i64, nch (B) i32 channels in use per image (NULL = C), labels (B,H,W) i64. */
int wc_par_labels(const float* masks, const int64_t* valid_key, const int* nch, int64_t* labels,
                  int B, int C, int H, int W, void* stream);

/* ---- bilinear plane resize ------------------------------------------------------------ */
/* cv2.resize(INTER_LINEAR) in clip/clip_tool.py:202-216 (align_corners=0);
 * F.interpolate(..., bilinear, align_corners=True) in WeCLIP_model/PAR.py:67;
 * F.interpolate(segs, bilinear, align_corners=False) in scripts/dist_clip_voc.py:250.
 * src (planes,Hs,Ws) f32 -> dst (planes,Hd,Wd) f32. */
int wc_bilinear_resize(const float* src, float* dst, int planes, int Hs, int Ws, int Hd, int Wd,
                       int align_corners, void* stream);
/* backward of the same (up-sampling): gsrc (planes,Hs,Ws) = sum over destination pixels of their
 * gradient gdst (planes,Hd,Wd) times the interpolation weight; separable two-pass gather, no atomics
 * (tmp: planes*Hs*Wd f32 workspace)
 * (autograd of F.interpolate at scripts/dist_clip_voc.py:250). */
int wc_bilinear_resize_bwd(const float* gdst, float* gsrc, float* tmp, int planes, int Hs, int Ws, int Hd,
                           int Wd, int align_corners, void* stream);

/* ---- segmentation loss fused with the logit up-sampling -------------------------------- */
/* scripts/dist_clip_voc.py:250 (bilinear up-sampling of seg to H x W) + get_seg_loss :105-113.
 * seg (B,nc,h,w) f32 low-res logits, label (B,H,W) i64 (ignore = 255).
 * wc_seg_loss_fwd: sums (8) = [sum nll over label==0, #label==0, sum nll over fg labels, #fg,
 *                  loss = 0.5*(sums[0]/sums[1] + sums[2]/sums[3]), 0.5/sums[1], 0.5/sums[3], 0]
 *                  (sums[5..6]: the backward weights wts for an upstream gradient of 1);
 *                  part: ceil(W/64)*ceil(H/4)*B*4 f32.
 * wc_seg_loss_bwd: ghr (B,nc,H,W) = wts[bg|fg] * (softmax - onehot) per pixel (0 for ignored); the
 *                  low-res gradient is wc_bilinear_resize_bwd(ghr). */
int wc_seg_loss_fwd(const float* seg, const int64_t* label, float* part, float* sums, int B, int nc, int h, int w,
                    int H, int W, int ignore, void* stream);
int wc_seg_loss_bwd(const float* seg, const int64_t* label, const float* wts, float* ghr, int B, int nc, int h,
                    int w, int H, int W, int ignore, void* stream);
/* The same backward without the (B,nc,H,W) high-resolution gradient: the soft-max gradient is formed per pixel inside the
 * Y pass of the separable bilinear backward (fixed summation order), then the X pass.  tmp: workspace B*nc*h*W f32;
 * out: (B,nc,h,w).  Same summation order as wc_seg_loss_bwd + wc_bilinear_resize_bwd (bit-identical for nc > 24; for
 * nc <= 24 the per-pixel log-sum-exp is formed max-first: equal to rounding). */
int wc_seg_loss_bwd_fused(const float* seg, const int64_t* label, const float* wts, float* tmp, float* out, int B, int nc,
                          int h, int w, int H, int W, int ignore, void* stream);

/* Training form of the two above: loss sums (as wc_seg_loss_fwd) AND d loss / d seg for an upstream gradient of 1 in ONE
 * pass over the pixels (a label count first gives the class weights).  cnt: 2048 floats, part: 4 floats per 64 x 4 block
 * of (W, h) per image, tmp: 2*B*nc*h*W floats, grad: (B,nc,h,w).  nc <= 24.  (scripts/dist_clip_voc.py:105-113,250) */
int wc_seg_loss_fwd_bwd(const float* seg, const int64_t* label, float* cnt, float* part, float* sums, float* tmp, float* grad,
                        int B, int nc, int h, int w, int H, int W, int ignore, void* stream);

/* Affinity loss fused with the affinity-label construction (reference utils/camutils.py:226-247 +
 * scripts/dist_clip_voc.py:116-133 radius mask + utils/losses.py:11-22): attn_pred (B,hw,hw) f32, cam_label (B,H,W)
 * int64 pseudo labels (nearest down-sampled to h x w inside), Chebyshev `radius`.
 * fwd: sums (8) = [sum_pos(1-p), n_pos, sum_neg(p), n_neg, loss = 0.5*s0/(s1+1) + 0.5*s2/(s3+1),
 *      -0.5/(s1+1), 0.5/(s3+1), 0]  (sums[5..6]: the backward coef for an upstream gradient of 1);
 *      part: workspace 4*B*ceil(hw/8) f32.
 * bwd: dap (B,hw,hw) = coef[0] on positive pairs, coef[1] on negative pairs, 0 elsewhere (coef: 2 device floats). */
int wc_aff_loss_fwd(const float* attn_pred, const int64_t* cam_label, float* part, float* sums, int B, int h,
                    int w, int H, int W, int radius, int ignore, void* stream);
int wc_aff_loss_bwd(const int64_t* cam_label, const float* coef, float* dap, int B, int h, int w, int H, int W,
                    int radius, int ignore, void* stream);

/* ---- MFMA GEMM ------------------------------------------------------------------------ */
/* C[M,N] = epilogue(sum_{s<nseg} A_s[M,K] * W_s[N,K]^T), fp16 operands (K contiguous), fp32
 * accumulate on v_mfma_f32_32x32x16_f16.  Replaces F.linear / nn.Linear / 1x1 Conv2d / bmm at
 * clip/myAtt.py:201 (in-proj), :321 (fp16 out-proj), clip/model.py:198-202 (MLP), :264-268
 * (patch-embed conv k16 s16 == GEMM), :420 (visual.proj), WeCLIP_model/segformer_head.py:22-28,76,
 * WeCLIP_model/Decoder/TransDecoder.py:122, WeCLIP_model/model_attn_aff_voc.py:136 (bmm).
 * nseg 2/3 = split precision (x = hi + lo): extra (A,W) pairs are accumulated into the same tile.
 * epilogue: v = acc + bias[n]; if round16: v = fp16(v); if n < scale_cols: v *= scale;
 *           v = act(v) (0 none, 1 QuickGELU, 2 ReLU, 3 sigmoid); v += resid[m*ldr + n];
 *           C32[m,n] = v; C16[m,n] = fp16(v); C16lo[m,n] = fp16(v - C16).
 *           P32 (optional) receives the pre-activation value.  act 4 (backward of QuickGELU,
 *           clip/model.py:186-188): v *= d/du[u*sigmoid(1.702u)] at u = aux[arow*ldaux + n],
 *           arow = rowmap[m / rpg]*rpg + m % rpg (rowmap NULL: arow = m).
 *           act 5 (backward of ReLU, segformer_head.py:26): v *= (auxh[m*ldaux + n] > 0), auxh fp16.
 *           act 6: exact GELU 0.5 v (1 + erf(v / sqrt 2)) (nn.GELU of the ViT-CoMer inserts); act 7: its backward,
 *           v *= Phi(u) + u phi(u) at u = aux[...] as for act 4.
 *           cscale (optional): v *= cscale[z*sCS + n] after the bias (Dropout2d mask/(1-p) of image z,
 *           segformer_head.py:78).
 * batch > 1: operand/output/residual batch strides sA/sW/sC/sR in elements.  K % 64 == 0, lda/ldw % 8 == 0. */
int wc_gemm_f16(const void* A0, const void* A1, const void* A2, const void* W0, const void* W1,
                const void* W2, int nseg, int M, int N, int K, long lda, long ldw, int batch,
                long sA, long sW, long sC, const float* bias, const float* resid, long ldr, long sR,
                float* C32,
                void* C16, void* C16lo, long ldc, int act, int round16, float scale,
                int scale_cols, float* P32, const float* aux, const int* rowmap, int rpg,
                long ldaux, const void* auxh, const float* cscale, long sCS, void* stream);

/* Grouped form of wc_gemm_f16: several small GEMMs of one shape in ONE launch -- the 11 adapter MLPs of
 * WeCLIP_model/segformer_head.py:58-76 (one nn.Linear pair per encoder block, the reference loops over them) times
 * the B images.  Batch index z splits into z2 = z / zdiv (group: adapter) and z1 = z % zdiv (image): A, W and the
 * outputs move by z1*s? + z2*s?2 elements, the bias by z2*sB2 and the act-5 aux by z2*sX2; everything else as
 * wc_gemm_f16 (which is this entry with zdiv = batch and zero second-level strides). */
int wc_gemm_f16_grouped(const void* A0, const void* A1, const void* A2, const void* W0, const void* W1,
                        const void* W2, int nseg, int M, int N, int K, long lda, long ldw, int batch,
                        long sA, long sW, long sC, const float* bias, const float* resid, long ldr, long sR,
                        float* C32, void* C16, void* C16lo, long ldc, int act, int round16, float scale,
                        int scale_cols, float* P32, const float* aux, const int* rowmap, int rpg,
                        long ldaux, const void* auxh, const float* cscale, long sCS, int zdiv, long sA2, long sW2,
                        long sC2, long sB2, long sX2, void* stream);

/* Row-streaming GEMM for tall, narrow products: C[M, N <= 256] = epilogue(A[M, K] W[N, K]^T), K = 128 | 256, fp16 operands,
 * fp32 accumulate (csrc/gemm_row.hip).  Replaces the same F.linear / 1x1-conv call sites as wc_gemm_f16 where the output row is
 * one complete channel vector: the Linear layers of the ViT-CoMer inserts (no reference code: ViT_CoMer.pdf section 3.2-3.3,
 * SURVEY.md section 8 row a-9) and WeCLIP_model/segformer_head.py:22-28, Decoder/TransDecoder.py:98-125 at N = 256.  The launch
 * is bound by HBM bytes, not flops: weights stay in registers, A and the side input stream through LDS once, every output row
 * leaves as whole coalesced rows.  Epilogue, in this order:  v = (acc + bias[n]) * cscale[n];  P32 = v;  act (0 none, 2 ReLU,
 * 6 GELU(erf); 5: v *= (auxh[m, n] > 0); 7: v *= GELU'(aux[m, n]));  v += resid[m, n];  C32 = v;  C16 = fp16(v);  and, for
 * N == 256, up to two LayerNorms of the finished row (fp32 statistics, clip/model.py:177-183 arithmetic) written as fp16:
 * ln_o{0,1}[m, :] = LN(v; ln_g, ln_b, eps) -- the operand of the next GEMM, without a LayerNorm launch.  At most ONE side input
 * per launch (resid, or aux with act 7, or auxh with act 5).  Any subset of outputs; null pointers = absent.
 * ldc = row pitch of C32 / P32, ldc16 = row pitch of C16 (it may be a column slice of a wider buffer).
 * lda, ldw % 8 == 0; ldc, ldc16, ldr, ldaux % 4 == 0 (ldaux % 8 for act 5); fp32 buffers 16-byte, fp16 outputs 8-byte aligned.
 * wc_gemm_row_supported: 1 when (M, N, K) is a shape this entry accepts. */
int wc_gemm_row_supported(int M, int N, int K);
int wc_gemm_row_f16(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const float* bias,
                    const float* cscale, int act, const float* aux, const void* auxh, long ldaux, const float* resid, long ldr,
                    float* C32, void* C16, float* P32, long ldc, long ldc16, const float* ln_g0, const float* ln_b0, void* ln_o0,
                    const float* ln_g1, const float* ln_b1, void* ln_o1, float eps, void* stream);

/* `groups` products of one shape in one launch of the row-streaming kernel (the eleven adapter MLPs' second Linear and its input
 * gradient, WeCLIP_model/segformer_head.py:22-28,69-80): group i reads A + i*gA, W + i*gW, bias + i*gB, auxh + i*gX and writes
 * C32 / C16 + i*gC (element offsets; e.g. gC = 256 with ldc16 = 11*256 drops every adapter's output into its column slice of the
 * fuse input).  act 0 | 2 | 5. */
int wc_gemm_row_f16_grouped(const void* A, long lda, const void* W, long ldw, int M, int N, int K, const float* bias, int act,
                            const void* auxh, long ldaux, float* C32, void* C16, long ldc, long ldc16, int groups, long gA,
                            long gW, long gB, long gC, long gX, void* stream);

/* Which kernel wc_gemm_f16 runs for a shape (for profiling / roofline bookkeeping only):
 * 0 = 128x128x64 kernel, 1 = 256x256x64 ping-pong kernel, 2 = ping-pong kernel + 128x128 kernel on the ragged
 * last M % 256 rows (two launches). */
int wc_gemm_plan(int M, int N, int K, int nseg, int batch);
/* Per-shape timing of the GEMM entry points when WECLIP_GEMM_LOG=1 (an event pair around every call): writes
 * "entry M= N= K= seg= batch= plan= act=\tcalls\tms\tflop" lines into buf, clears the log, returns the number of lines. */
int wc_gemm_log_report(char* buf, int cap);
/* out[i] = alpha * sum_s part[s*n + i]: reduction of split-K partial products (the slices are a
 * batched wc_gemm_f16 over K ranges: sA = sW = K/slices, sC = M*N). */
int wc_sum_slices(const float* part, float* out, int nslices, long n, float alpha, void* stream);

/* Split-K reduction of a weight-gradient GEMM whose X^T operand carried a ones row (the bias-gradient column):
 * part (nslices, rows, cols+1) -> out_w (rows, cols) dense and out_b (rows); replaces the autograd-produced
 * .weight/.bias gradients of nn.Linear / 1x1 nn.Conv2d (reference WeCLIP_model/segformer_head.py:22-28). */
int wc_sum_slices_wb(const float* part, float* out_w, float* out_b, int nslices, int rows, int cols,
                     float alpha, void* stream);
/* `groups` such reductions of one shape in one launch: part (groups, nslices, rows, cols+1); group i writes
 * out_w + i*gW and out_b + i*gB (elements) -- the per-adapter gradient views of a flat gradient bucket. */
int wc_sum_slices_wb_grouped(const float* part, float* out_w, float* out_b, int nslices, int rows, int cols,
                             float alpha, int groups, long gW, long gB, void* stream);
/* Many split-K reductions in one launch.  jobs: HOST array of count x 8 int64 {part, out_w, out_b (device pointers), nslices,
 * rows, cols, alpha as IEEE float bits, slice stride in elements (0 = rows * (cols + 1); larger when the job reduces a row
 * range of a wider partial matrix)}; each job = one wc_sum_slices_wb (same summation order).  The jobs travel by value
 * in the kernel arguments (graph-capturable, no device table). */
int wc_sum_slices_wb_multi(const int64_t* jobs, int count, void* stream);

/* Weight-gradient GEMM on row-major operands: part[z, n, k] = sum over the tokens m of slice z of dY[m, n] * X[m, k]
 * (and, with bias != 0, one more column k = K holding sum_m dY[m, n]); z = 0 .. ceil(M / mslice) - 1, mslice a multiple
 * of 64.  fp16 operands (M, lda) / (M, ldx), fp32 partials (nslices, N, K + bias) to be summed by wc_sum_slices(_wb).
 * Replaces the autograd weight/bias gradients of nn.Linear / 1x1 nn.Conv2d (reference
 * WeCLIP_model/segformer_head.py:22-28,69-80; Decoder/TransDecoder.py:98-125) without transposed operand copies.
 * The X row of token m is (m / x_rpg) * x_gs + m % x_rpg + x_off (use x_rpg = M, x_gs = 0, x_off = 0 for a
 * dense matrix) -- lets X be the patch rows of a (B, 1 + hw, C) token tensor.  zeros: >= 16 zero bytes. */
int wc_gemm_km_f16(const void* dY, long lda, const void* X, long ldx, const void* zeros, int M, int N, int K,
                   int x_rpg, int x_gs, int x_off, int mslice, int bias, float* part, void* stream);
/* `groups` weight gradients of one shape in one launch (the 11 adapter Linears, segformer_head.py:58-76): group i
 * reads dY + i*gA and X + i*gX (elements) and writes part + i*nslices*N*(K+bias). */
int wc_gemm_km_f16_grouped(const void* dY, long lda, const void* X, long ldx, const void* zeros, int M, int N, int K,
                           int x_rpg, int x_gs, int x_off, int mslice, int bias, float* part, int groups, long gA,
                           long gX, void* stream);
/* fp32 -> fp16 hi (+ lo = fp16(x - hi), may be NULL): `.half()` casts of weights/activations
 * (clip/model.py:457-478 convert_weights; clip/myAtt.py:321). */
int wc_split_f16(const float* x, void* hi, void* lo, long n, void* stream);

/* Many fp32 weight matrices -> fp16 operands in one launch: row-major hi[,lo] (R,C) and transposed hiT[,loT]
 * (C, ldT >= R; padding columns are left untouched -- keep them zero).  table (device): count rows of 8 int64
 * {src, hi, lo|0, hiT|0, loT|0, R, C, ldT}.  Replaces the per-layer `.half()` / `.t()` of the trainable weights. */
int wc_convert_weights(const int64_t* table, int count, int blocks_per_tensor, void* stream);

/* One AdamW step over many fp32 parameter tensors in one launch (reference utils/optimizer.py:3-33: torch.optim.AdamW
 * with the poly-warm-up lr written per group).  table (device): count rows of 8 int64 {param, grad, exp_avg, exp_avg_sq,
 * numel, 0, 0, 0}.  p *= 1 - lr*wd; m = lerp(m, g, 1-b1); v = v*b2 + (1-b2) g^2; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
 * with bias_correction{1,2} = 1 - beta^step (the update order of torch._multi_tensor_adam). */
int wc_adamw_multi(const int64_t* table, int count, double lr, double beta1, double beta2, double eps,
                   double weight_decay, double bias_correction1, double bias_correction2, int blocks_per_tensor,
                   void* stream);

/* ---- LayerNorm ------------------------------------------------------------------------ */
/* clip/model.py:177-183 (`LayerNorm.forward`, fp32 math).  x (rows, D) f32 with row stride ldx;
 * outputs (each optional): y32 f32, y16 fp16 hi, y16lo fp16 residual; all dense (rows, D). */
int wc_layernorm(const float* x, long ldx, const float* w, const float* b, float eps, float* y32,
                 void* y16, void* y16lo, long rows, int D, void* stream);

/* ---- multi-head attention -------------------------------------------------------------- */
/* Packed in-projection output qkv (B*L, 3E) fp16, E = H*DH, q pre-scaled by log2(e)/sqrt(DH)
 * (wc_gemm_f16 scale/scale_cols).  DH in {32, 64}.
 * wc_attn_fwd:  clip/myAtt.py:21-64 without materialising the scores: out (B*L, E) fp16 =
 *               softmax(QK^T)V with heads merged (the `.half()` input of the out-projection,
 *               myAtt.py:319-321); out32 (optional) the same before fp16 rounding;
 *               lse (B,H,L) f32 = log2-sum-exp2 of each score row.  V is read row-major from qkv.
 * wc_attn_mean: clip/myAtt.py:325-326: mean (B,L,L) f32 = (1/H) sum_h softmax_h. */
int wc_attn_fwd(const void* qkv, void* out, float* out32, float* lse, int B, int L, int H, int DH,
                void* stream);
int wc_attn_mean(const void* qkv, const float* lse, float* mean, int B, int L, int H, int DH,
                 void* stream);

/* ---- ViT patch embedding ---------------------------------------------------------------- */
/* clip/model.py:264-272.  wc_patchify: img (B,3,H,W) f32 -> im2col rows (B*h*w, 3*P*P) fp16 hi
 * (+lo, may be NULL) in the conv weight's (c,ky,kx) order, so conv1 == wc_gemm_f16.
 * wc_cls_rows: x[b,0,:] = class_embedding + pos[0] for the token tensor x (B,L,E) f32. */
int wc_patchify(const float* img, void* hi, void* lo, int B, int H, int W, int P, void* stream);
int wc_cls_rows(float* x, const float* cls, const float* pos0, int B, int L, int E, void* stream);

/* ---- GradCAM on the last CLIP block (analytic, batched over (image, class) pairs) ------- */
/* Replaces pytorch_grad_cam/base_cam.py:62-154, grad_cam.py:16-23,
 * activations_and_gradients.py:19-47 (autograd through CLIP.forward_last_layer,
 * clip/model.py:407-429).  P pairs; pair_img[p] = image of pair p, pair_cls[p] = index of the
 * target row among that pair's text rows.  L tokens, E width, Ed joint embedding width.
 *
 * wc_cam_head: x2 (B,L,E) f32 block output -> ln_post, mean over patch tokens, @proj (E,Ed),
 *   cosine logits against pre-normalised text rows text[text_idx[p,t]] (t < n_text[p], row
 *   stride Tmax), softmax -> probs (P,Tmax); df (P,E) = d probs[p,cls] / d pooled feature.
 *   partial: workspace B*ceil(L/16)*E f32.  logit_scale = exp(CLIP.logit_scale).
 * wc_lnpost_bwd: dx2 (P,L,E) = gs * backward of ln_post+mean-pool (token 0 gets 0);
 *   written as f32 and as fp16 hi/lo GEMM operands.
 * wc_ln2_bwd_add: g16 = fp16((dx2 + LN2_bwd(da2; x1)) / gs): the gradient arriving at the fp16
 *   out-projection output (autograd rounds it to fp16, clip/myAtt.py:321).
 * wc_attn_bwd_colsum: column sums over tokens of dq,dk,dv -> c (P,3E) f32 from the packed
 *   qkv (B*L,3E fp16, q pre-scaled), dO (P*L,E fp16), o32 (B*L,E), lse (B,H,L).
 *   Workspaces: delta,u,dS0,P0 each (P,H,L) f32.
 * wc_rowvec_matmul: out (P,E) = scale * c (P,N) @ W (N,E), fp32 FMA; ws: workspace 16*P*E f32.
 * wc_cam_map: cam (P,L-1) = scale_cam(scale_cam(relu(A w))) with A = a32[img, 1:, :]
 *   (base_cam.py:56-60,144-154; utils/image.py:51-61). */
int wc_cam_head(const float* x2, const float* lnw, const float* lnb, const float* proj,
                const float* text, const int* text_idx, const int* n_text, const int* pair_img,
                const int* pair_cls, float logit_scale, float* partial, float* probs, float* df,
                int B, int P, int L, int E, int Ed, int Tmax, void* stream);
int wc_lnpost_bwd(const float* df, const float* x2, const float* lnw, float gs, const int* pair_img,
                  float* d32, void* dhi, void* dlo, int P, int L, int E, void* stream);
int wc_ln2_bwd_add(const float* da2, const float* dx2, const float* x1, const float* lnw, float gs,
                   const int* pair_img, void* g16, int P, int L, int E, void* stream);
int wc_attn_bwd_colsum(const void* qkv, const void* dO, const float* o32, const float* lse,
                       const int* pair_img, float* delta, float* u, float* dS0, float* P0, float* c,
                       int P, int L, int H, int DH, void* stream);
int wc_rowvec_matmul(const float* c, const float* W, float* out, float* ws, int P, int N, int E, float scale,
                     void* stream);
int wc_cam_map(const float* a32, const float* w, const int* pair_img, float* cam, int P, int L, int E,
               void* stream);

/* ---- attention-affinity CAM refinement -------------------------------------------------- */
/* clip/clip_tool.py:152-191, compute_trans_mat :64-80, clip/utils.py:115-142 (scoremap2bbox),
 * generate_cam_label :202-216, WeCLIP_model/model_attn_aff_voc.py:158-163.  hw = L-1 patch tokens.
 * wc_aff_weight:      W (B,hw,hw) = sum_l wgt[b,l] * maps[l][b,1:,1:] (* seg[b] if seg != NULL);
 *                     h_maps: HOST array of nmaps (<=12) device pointers to (B,L,L) head-mean maps.
 * wc_aff_seg_weights: seg-trans layer selection (:158-167): diff (B,nmaps) workspace,
 *                     wgt[b,l] = [diff <= mean diff] / (count + 1e-5).
 * wc_matvec:          out (B,hw,K) = f(W x) (transpose=0) or f(W^T x) (transpose=1),
 *                     x = X * sin[:,None]; f = 1/v if recip else alpha*v*sout[:,None] + add.
 *                     Sinkhorn (:64-75) as scale vectors: c = 1/(W^T r), r = 1/(W c), 3 rounds;
 *                     T_sym x = (r*(W(c*x)) + c*(W^T(r*x)))/2; refined = T_sym(T_sym(mask*cam)).
 * wc_tsym:            materialise T_sym (B,hw,hw) (public compute_trans_mat only).
 * wc_box_mask:        per pair p (cam (P,h*w) in [0,1]): u8 quantise, > int(thr*max), 8-connected
 *                     components, boxes [x0,y0,min(x1+1,w-1),min(y1+1,h-1)], half-open fill;
 *                     V[pair_img[p], :, pair_slot[p]] = mask*cam (V is (B,hw,K));
 *                     optional mask_out (P,hw) f32, boxes (P,maxbox,4) i32 + nbox (P).
 * wc_cam_upsample:    R (B,hw,K) refined maps -> cams (B,C,H,W): channel 1+k = bilinear(half-pixel)
 *                     of min-max(R_k) for k < nk[b] (0 beyond), channel 0 = 1 - max_k.
 *                     stats: workspace B*K*2 f32. */
int wc_aff_weight(const float* const* h_maps, int nmaps, const float* wgt, const float* seg, float* W,
                  int B, int L, void* stream);
int wc_aff_seg_weights(const float* const* h_maps, int nmaps, const float* seg, float* diff, float* wgt,
                       int B, int L, void* stream);
int wc_matvec(const float* W, const float* X, const float* sin, const float* sout, const float* add,
              float* out, int B, int hw, int K, int transpose, int recip, float alpha, void* stream);
/* Fused forms for hw %% 4 == 0 (wc_aff_fused_supported): a workgroup owns 32 complete rows of W, so one read of W serves a
 * Sinkhorn row pass AND the next column pass, or both halves of T_sym X; W is read 5 times per batch instead of 10.
 * wc_aff_weight_c1:     W as wc_aff_weight + c1 = 1 / colsum(W).                ws: B*ceil(hw/8)*hw f32
 * wc_aff_sinkhorn_step: r = 1/(W c); unless last: c_next = 1/(W^T r).            ws: B*ceil(hw/32)*hw f32
 * wc_aff_tsym_apply:    out (B,hw,K) = T_sym X, K <= 4.  y1: B*hw*K f32; ws: B*ceil(hw/32)*hw*K f32. */
int wc_aff_fused_supported(int hw);
int wc_aff_weight_c1(const float* const* h_maps, int nmaps, const float* wgt, const float* seg, float* W, float* c1,
                     float* ws, int B, int L, void* stream);
int wc_aff_sinkhorn_step(const float* W, const float* c, float* r, float* c_next, float* ws, int B, int hw, int last,
                         void* stream);
int wc_aff_tsym_apply(const float* W, const float* r, const float* c, const float* X, float* out, float* y1, float* ws,
                      int B, int hw, int K, void* stream);
int wc_tsym(const float* W, const float* r, const float* c, float* T, int B, int hw, void* stream);
int wc_box_mask(const float* cam, const int* pair_img, const int* pair_slot, float* V, float* mask_out,
                int* boxes, int* nbox, int maxbox, int P, int h, int w, int K, double thr, void* stream);
int wc_cam_upsample(const float* R, const int* nk, float* stats, float* cams, int B, int h, int w, int K,
                    int C, int H, int W, void* stream);

/* ---- backward helpers of the trainable adapters / decoder ------------------------------ */
/* autograd through WeCLIP_model/segformer_head.py:69-80, Decoder/TransDecoder.py:63-125,
 * model_attn_aff_voc.py:134-137 (all run by torch.autograd in the reference).
 * wc_transpose_f16:  out[c, b*oR + r] = fp16(scale*src[b,r,c]) (hi[,lo]); src f32 or f16 (src_f32),
 *                    row stride ld, batch stride sSrc; out row stride ldo: operands of dW = dY^T X.
 * wc_colsum:         out[c] = alpha * sum_r src[r,c] (bias gradients); part: ceil(R/256)*C f32.
 * wc_layernorm_bwd:  dx = add + LN_bwd(dy; x, w) as f32 and/or fp16(dx*out_scale);
 *                    dgb (2,D) = alpha*[sum dy*xhat ; sum dy]; part: ceil(rows/16)*2*D f32.
 * wc_sigmoid_gram_bwd: S[b] = scale*(Z + Z^T), Z = dAP*AP*(1-AP) (fp16 hi[,lo], row stride ldo) so dF = S F.
 * wc_colscale_split: y = alpha * x * cs[row / rows_per_batch, col] (cs may be NULL) -> f32 (optional), fp16 hi[,lo]. */
/* wc_attn_bwd: backward of clip/myAtt.py:21-64 without storing L x L tensors: from the packed qkv
 * (q pre-scaled), dO (B*L,E) fp16, o32, lse -> dqkv (B*L,3E) fp16 hi (+lo, may be NULL), gradients
 * w.r.t. the UNSCALED in-projection output.  Workspaces qt, kt, dot: B*H*DH*Lp halves each;
 * delta: B*H*L floats. */
int wc_attn_bwd(const void* qkv, const void* dO, const float* o32, const float* lse, void* qt, void* kt,
                void* dot, float* delta, void* dqkv_hi, void* dqkv_lo, int B, int L, int Lp, int H, int DH,
                void* stream);
int wc_transpose_f16(const void* src, int src_f32, long ld, long sSrc, void* hi, void* lo, long ldo,
                     long oR, int batch, int R, int C, float scale, void* stream);
int wc_colsum(const void* src, int src_f32, long ld, float* part, float* out, long R, int C, float alpha,
              int round16, void* stream);
int wc_layernorm_bwd(const float* dy, const float* x, const float* w, const float* add, float eps,
                     float* dx32, void* dx16, float out_scale, float* part, float* dgb, float alpha,
                     long rows, int D, void* stream);
/* the same with the incoming gradient as fp16 rows dy16 (rows, D) */
int wc_layernorm_bwd_h(const void* dy16, const float* x, const float* w, const float* add, float eps,
                       float* dx32, void* dx16, float out_scale, float* part, float* dgb, float alpha,
                       long rows, int D, void* stream);
/* two LayerNorms of ONE input x (the ViT-CoMer CTI normalises c1 once for the values of one deformable attention and once
 * for the queries of the other: this package's own WeCLIP_model/comer.py CTI.forward; no reference code) back-propagated in one
 * pass: dx = LN_bwd_a(dya16) + LN_bwd_b(dyb16) [+ add]; dgba / dgbb (2, D) = alpha * [dgamma; dbeta] of each; D <= 256;
 * part: >= min(ceil(rows / 16), 2048) * 4 * D floats */
int wc_layernorm_bwd2_h(const void* dya16, const float* wa, const void* dyb16, const float* wb, const float* x,
                        const float* add, float eps, float* dx32, void* dx16, float out_scale, float* part, float* dgba,
                        float* dgbb, float alpha, long rows, int D, void* stream);
int wc_sigmoid_gram_bwd(const float* dAP, const float* AP, void* hi, void* lo, int B, int n, int ldo,
                        float scale, void* stream);
int wc_colscale_split(const float* x, const float* cs, float* out32, void* hi, void* lo, long rows, int C,
                      int rows_per_batch, float alpha, void* stream);

/* ---- ViT-CoMer: multi-scale deformable attention core ---------------------------------- */
/* No reference code exists for the CoMer inserts (paper ViT_CoMer.pdf 3.3 / Deformable-DETR eq. 2-3).
 * value (N,S,M,D) f32, S = sum_l H_l*W_l (h_shapes: HOST array of n_levels (H,W) pairs);
 * loc (N,Lq,M,nL,P,2) f32 in [0,1] as (x,y); attn (N,Lq,M,nL,P) f32; out (N,Lq,M*D) f32:
 * out = sum_l sum_p attn * bilinear_zero_pad(value_l, loc*size - 0.5).
 * wc_msda_bwd: gvalue (every element written once: no initialisation needed), gloc, gattn.  grad_value is a bucketed
 *              gather (count / scan / fill per pixel, then a per-pixel sum in 64-bit fixed point, scale 2^40 / max|gout|):
 *              no float atomics anywhere, the result is independent of the execution order.  gmax: workspace of one
 *              uint32; ws: workspace of N*M*S*2 + N*M*n_levels*Lq*P*8 int32; a level may have at most 16384 pixels. */
int wc_msda_fwd(const float* value, const int* h_shapes, int n_levels, const float* loc, const float* attn,
                float* out, int N, int Lq, int M, int D, int P, void* stream);
/* The same with the value tensor as f32 or f16 (value_is_f16: half the gather traffic; needs D % 4 == 0 and M*D/4 dividing
 * 256) and an optional fp16 copy of the output (the MFMA operand of MSDeformAttn's output projection); out or out16 may be
 * NULL. */
int wc_msda_fwd_h(const void* value, int value_is_f16, const int* h_shapes, int n_levels, const float* loc,
                  const float* attn, float* out, void* out16, int N, int Lq, int M, int D, int P, void* stream);
int wc_msda_bwd(const float* value, const int* h_shapes, int n_levels, const float* loc, const float* attn,
                const float* gout, float* gvalue, float* gloc, float* gattn, void* gmax, void* ws, int N, int Lq, int M,
                int D, int P, void* stream);
/* The same with value and / or the output gradient gout as f16 (*_is_f16) and the value gradient as f32 (gvalue) and / or
 * f16 (gvalue16: the operand of the value projection's gradient GEMMs); either of the two may be NULL. */
int wc_msda_bwd_h(const void* value, int value_is_f16, const int* h_shapes, int n_levels, const float* loc,
                  const float* attn, const void* gout, int gout_is_f16, float* gvalue, void* gvalue16, float* gloc,
                  float* gattn, void* gmax, void* ws, int N, int Lq, int M, int D, int P, void* stream);

/* Depth-wise conv2d (stride 1, zero "same" padding k/2, odd k <= 7) of the MRFP block of the ViT-CoMer inserts
 * (nn.Conv2d(C, C, k, padding=k//2, groups=C); no reference code).  x, y, dy, dx: (N, C, H, W) f32;
 * w, dw: (C, 1, k, k); bias, db: (C) (bias / db may be NULL); part: workspace N*C*(k*k+1) f32. */
int wc_dwconv_fwd(const float* x, const float* w, const float* bias, float* y, int N, int C, int H, int W, int k,
                  void* stream);
int wc_dwconv_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, float* part,
                  int N, int C, int H, int W, int k, void* stream);

/* ---- ViT-CoMer spatial-prior conv stem on token rows (no reference code) ---- */
/* Activations are rows x[(n*H + y)*W + x][c] (NHWC).  A 3x3 / stride-s / pad-1 convolution = wc_im2col3x3 + wc_gemm_f16
 * against Wmat[o][(ky*3+kx)*C + c] = w[o][c][ky][kx] (zero padded to Kp); its input gradient = wc_gemm_f16 (dY W) +
 * wc_col2im3x3 (a deterministic gather).  wc_groupnorm_relu_*: nn.GroupNorm(G, C) followed by ReLU, all reductions in
 * a fixed order.  stats: N*G*2 f32 (mean, rstd); part: N*ceil(HW/64)*G*2 f32; gpart like part; cpart:
 * N*ceil(HW/64)*C*2 f32; gsum: N*G*2 f32. */
int wc_im2col3x3(const float* x, void* cols_hi, void* cols_lo, int N, int H, int W, int C, int stride, int Kp,
                 void* stream);
int wc_col2im3x3(const float* dcols, float* dx, int N, int H, int W, int C, int stride, int Kp, void* stream);
int wc_groupnorm_relu_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats, float* part,
                          int N, int HW, int C, int G, float eps, void* stream);
int wc_groupnorm_relu_bwd(const float* x, const float* y, const float* dy, const float* stats, const float* gamma,
                          float* dx, float* dgamma, float* dbeta, float* gpart, float* cpart, float* gsum, int N,
                          int HW, int C, int G, void* stream);

/* ---- multi-scale + flip inference and the confusion histogram ----------------------------- */
/* test_msc_flip_coco.py:61-96 (`validate`: [img, flip] at every scale, flip-average, mean over scales, bilinear to
 * the label size, argmax) and utils/evaluate.py:10-16 (_fast_hist), per image, on the device.
 * wc_scale_flip_pair: dst (2,C,Hd,Wd) = [bilinear(src (C,Hs,Ws)), its horizontal flip]; scale_y/x = the source step
 *                     per destination pixel (in/out for F.interpolate(size=), 1/s for F.interpolate(scale_factor=s));
 *                     plain copy + flip when nothing is resized.
 * wc_flip_avg:        out (C,Hd,Wd) = (accumulate ? out : 0) + weight * (R(segs[0]) + flip(R(segs[1]))) / 2 with
 *                     segs (2,C,Hs,Ws) and R = F.interpolate(size=(Hd,Wd), bilinear, align_corners=False).
 * wc_resize_argmax:   pred (Hd,Wd) int64 = argmax_c R(seg (C,Hs,Ws)); the (C,Hd,Wd) logits are never materialised.
 * wc_confusion_hist:  hist (nc,nc) int64 += counts of (true, pred) pairs over n pixels whose true label is in
 *                     [0,nc); *flag is set to 1 if a prediction is outside [0,nc) (int32, caller-zeroed). */
int wc_scale_flip_pair(const float* src, float* dst, int C, int Hs, int Ws, int Hd, int Wd, float scale_y,
                       float scale_x, void* stream);
int wc_flip_avg(const float* segs, float* out, int C, int Hs, int Ws, int Hd, int Wd, float weight, int accumulate,
                void* stream);
int wc_resize_argmax(const float* seg, long* pred, int C, int Hs, int Ws, int Hd, int Wd, void* stream);
int wc_confusion_hist(const long* label_true, const long* label_pred, long* hist, int* flag, long n, int nc,
                      void* stream);

/* Fused forms of the two calls above for the CTI configurations ((n_levels, P) = (3, 4) or (1, 4); wc_msda_fused_supported
 * returns 1): the sampling locations and soft-maxed weights are computed inside the forward kernel from the raw rows
 * ow (N*Lq, ld) = [M*nL*P*2 offsets | M*nL*P logits] of the fused sampling_offsets | attention_weights Linear, their biases
 * (optional) and the reference points ref (Lq, nl_ref, 2); loc / attn are written for the backward.  The backward writes the
 * gradient of the raw rows as the f16 GEMM operand dow16 (N*Lq, ld), padding columns zeroed. */
int wc_msda_fused_supported(int n_levels, int M, int D, int P);
int wc_msda_fwd_f(const void* value, int value_is_f16, const int* h_shapes, int n_levels, const float* ow, int ld,
                  const float* bias_off, const float* bias_aw, const float* ref, int nl_ref, float* loc, float* attn,
                  float* out, void* out16, int N, int Lq, int M, int D, int P, void* stream);
int wc_msda_bwd_f(const void* value, int value_is_f16, const int* h_shapes, int n_levels, const float* loc,
                  const float* attn, const void* gout, int gout_is_f16, float* gvalue, void* gvalue16, void* dow16,
                  int ld, void* gmax, void* ws, int N, int Lq, int M, int D, int P, void* stream);

/* ---- fused glue of the ViT-CoMer insert engine (csrc/comer.hip; no reference code exists: ViT_CoMer.pdf §3.2-3.3) ---- */
/* MRFP's depth-wise convolutions on token rows x (N, S, C) f32, S = sum of the n_levels maps h_shapes = {H0, W0, H1, W1, ...}:
 * 3x3 filters w3 (C/2, 9) + b3 on channels [0, C/2), 5x5 filters w5 (C/2, 25) + b5 on [C/2, C), zero padding, all levels in
 * one launch.  y (f32, may be NULL) = conv + bias; g16 (f16, may be NULL) = GELU(y) (erf form), the next FC's operand. */
int wc_mrfp_dwconv_fwd(const float* x, const float* w3, const float* b3, const float* w5, const float* b5, float* y,
                       void* g16, const int* h_shapes, int n_levels, int N, int C, void* stream);
/* Backward: dy (f16 if dy_is_f16 else f32); dx32 / dx16 (either may be NULL) = gradient w.r.t. x; dw3 / db3 / dw5 / db5 =
 * alpha * filter / bias gradients (two-stage fixed-order reduction); part: workspace of n_parts * C * 26 floats with n_parts
 * from wc_mrfp_dwconv_parts (HOST out-parameter). */
int wc_mrfp_dwconv_parts(const int* h_shapes, int n_levels, int N, int C, long* n_parts);
int wc_mrfp_dwconv_bwd(const void* dy, int dy_is_f16, const float* x, const float* w3, const float* w5, float* dx32, void* dx16,
                       float* dw3, float* db3, float* dw5, float* db5, float* part, float alpha, const int* h_shapes,
                       int n_levels, int N, int C, void* stream);
/* MSDeformAttn's sampling_offsets | attention_weights outputs ow (N*Lq, ld) [M*nL*P*2 offsets, then M*nL*P logits per row]
 * -> loc (N,Lq,M,nL,P,2) = ref + offset / (W_l, H_l) and attn (N,Lq,M,nL,P) = softmax over nL*P.  ref (Lq, nl_ref, 2),
 * nl_ref = 1 (one point for all levels) or n_levels.  bias_off (M*nL*P*2) / bias_aw (M*nL*P), optional: the two Linears'
 * biases, added here when the fused GEMM ran without one. */
int wc_msda_prep_fwd(const float* ow, const float* bias_off, const float* bias_aw, const float* ref, float* loc, float* attn,
                     const int* h_shapes, int n_levels, int N, int Lq, int M, int P, int ld, int nl_ref, void* stream);
/* Its backward: dow (N*Lq, ld) as f32 and / or f16 from the gradients of loc and attn; columns >= 3*M*nL*P are zeroed. */
int wc_msda_prep_bwd(const float* gloc, const float* gattn, const float* attn, float* dow32, void* dow16,
                     const int* h_shapes, int n_levels, int N, int Lq, int M, int P, int ld, void* stream);
/* dst[b][r][0:C] (f16; row stride ld_dst, batch stride s_dst) = src[b][r][0:C] (f32 if src_is_f32 else f16; ld_src, s_src). */
int wc_rows_copy_f16(const void* src, int src_is_f32, void* dst, int B, int R, int C, long ld_src, long s_src,
                     long ld_dst, long s_dst, void* stream);
/* dst[b][r][0:C] (f32, dense rows of C, batch stride s_dst) += alpha * src[b][r][0:C] (f32; ld_src, s_src). */
int wc_rows_add_f32(const float* src, float* dst, int B, int R, int C, long ld_src, long s_src, long s_dst, float alpha,
                    void* stream);

/* Gradients of the gated CTI output projection v1 = v + gamma * (o1 Wop^T + bop) from the un-gated weight-gradient products
 * G (C, K) = dv1^T o1 and gs (C) = dv1^T 1:  dWop = diag(gamma) G, dbop = gamma * gs, dgamma = rowsum(Wop * G) + bop * gs
 * (all f32; no reference code: ViT_CoMer.pdf section 3.3, the gate of the CNN -> ViT injection). */
int wc_cti_gate_grads(const float* G, const float* gs, const float* gamma, const float* Wop, const float* bop, float* dWop,
                      float* dbop, float* dgamma, int C, int K, void* stream);

/* ---- device-side input pipeline ------------------------------------------------------------ */
/* datasets/transforms.py:26-49 (random_scaling), :70-84 (random_fliplr), :119-176 (random_crop, zero padding),
 * :8-15 (normalize_img) + HWC->CHW (datasets/voc.py:137-143) for a batch in one gather kernel.
 * src_u8 (B,Hs,Ws,3) uint8; params: B records of 8 x 32 bit {float scale; int flip, rh, rw, pad_y, pad_x, crop_y, crop_x}
 * (device memory; the random draws are made on the host); dst (B,3,crop,crop) f32; mean3 / std3: HOST float[3];
 * coeff_ws: device scratch of *n_ints ints as reported by wc_augment_workspace_ints(B, crop, &n_ints) (the per-coordinate filter tables).
 * The rescale is Pillow's Image.BILINEAR for 8-bit images reproduced exactly (transforms.py:41: triangle filter of support
 * max(in/out, 1), double-precision coefficients rounded to 22-bit fixed point, uint8 rounding after the horizontal and after
 * the vertical pass).  PRECONDITION, checked on the device: in/out <= 4 on both axes (Hs <= 4 rh, Ws <= 4 rw; the reference's
 * rescale_range is [0.5, 2.0]).  The filter tables hold 9 taps; an output coordinate that would need more makes every pixel
 * it touches NaN instead of applying a truncated (non-Pillow) filter.  data.DeviceAugment also rejects it on the host when
 * the params are host-resident. */
int wc_augment_workspace_ints(int B, int crop, long* n_ints);       /* HOST out-parameter */
int wc_augment_normalize(const void* src_u8, const void* params, float* dst, int* coeff_ws, int B, int Hs, int Ws, int crop,
                         const float* mean3, const float* std3, void* stream);

#ifdef __cplusplus
}
#endif
#endif
