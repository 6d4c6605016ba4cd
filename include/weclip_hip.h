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
 * out/tmp: (B,C,H,W) f32 workspaces (result in out); aff_ws: min(group,B)*8*n_dil*H*W f32.
 * Images are swept in groups of `group` so a group's aff planes stay cache resident. */
int wc_par_forward(const float* img, const float* masks, float* out, float* tmp, float* aff_ws,
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

#ifdef __cplusplus
}
#endif
#endif
