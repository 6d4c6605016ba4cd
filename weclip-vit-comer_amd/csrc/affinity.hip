// Attention-affinity CAM refinement for gfx950 (HBM/L2-bound, fp32).
//
// Replaces reference clip/clip_tool.py:152-191 + compute_trans_mat :64-80 + clip/utils.py:115-142
// (scoremap2bbox, cv2 on the CPU) + generate_cam_label :202-216 and the bg/concat glue of
// WeCLIP_model/model_attn_aff_voc.py:158-163:
//   aff_weight_kernel : W = mean of the selected head-mean maps [1:,1:] (optionally the seg-trans
//                       masked mean times the predicted affinity)           -> (B, hw, hw)
//   Sinkhorn as scale vectors: 3 x (column-normalise, row-normalise) of W only rescales rows and
//                       columns, so T = diag(r) W diag(c) with c = 1/(W^T r), r = 1/(W c); T is never
//                       written.  matvec_cols / matvec_rows are the two passes over W.
//   refinement        : (T_sym^2 * mask) @ cam = T_sym (T_sym (mask * cam)),
//                       T_sym x = (r*(W(c*x)) + c*(W^T(r*x)))/2 : four mat-vec passes, all classes of
//                       an image at once (the reference materialises T_sym, T_sym^2 and T*mask).
//   box_mask_kernel   : u8 quantise, threshold, 8-connected components, bounding boxes, half-open
//                       box fill -- on the device, one workgroup per (image, class) pair.
//   cam_upsample_kernel: per-map min-max, bilinear (half-pixel) resize to H x W, bg = 1 - max_k.
#include "common.h"

#define MAXK 4   // classes per pass in the mat-vec kernels

struct MapPtrs {
    const float* p[12];
};

// W[b,i,j] = sum_l wgt[b,l] * maps[l][b, i+1, j+1]  (* seg[b,i,j] if seg)
__global__ __launch_bounds__(256) void aff_weight_kernel(MapPtrs maps, int nmaps, const float* __restrict__ wgt,
                                                          const float* __restrict__ seg, float* __restrict__ W,
                                                          int L) {
    const int hw = L - 1;
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
    if (j >= hw) return;
    const long src = (long)b * L * L + (long)(i + 1) * L + (j + 1);
    // all map loads are issued before the first FMA (the rolled loop kept one 4-byte load in flight per lane and
    // streamed the 8 x 67 MB of maps at 3.2 TB/s); same summation order
    float v[12];
#pragma unroll
    for (int l = 0; l < 12; ++l) v[l] = l < nmaps ? maps.p[l][src] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int l = 0; l < 12; ++l)
        if (l < nmaps) s = fmaf(wgt[b * nmaps + l], v[l], s);
    const long dst = ((long)b * hw + i) * hw + j;
    if (seg) s *= seg[dst];
    W[dst] = s;
}

// Layer selection of the seg-trans branch (clip_tool.py:158-167): keep layer l iff
//   diff_l <= mean_l diff,   diff_l = sum_ij (seg[b,i,j] - maps[l][b,i+1,j+1]) = S - A_l.
// The seg term S is the same for every layer and cancels: diff_l <= mean(diff)  <=>  A_l >= mean(A) with
// A_l = sum_ij maps[l][b,i+1,j+1].  Only A_l is reduced (~1e3, where fp32 spacing is 1e-4, instead of S - A_l ~5e5
// with spacing 0.03-0.06), in a FIXED order: one partial per (image, layer, row), then a serial sum over the rows --
// no atomics, so the decision (and with it the pseudo labels) is reproducible run to run.
// rowsum[(b*nmaps + l)*hw + i] = sum_j maps[l][b,i+1,j+1]
__global__ __launch_bounds__(256) void aff_rowsum_kernel(MapPtrs maps, int nmaps, float* __restrict__ rowsum, int L) {
    __shared__ float red[16];
    const int hw = L - 1, i = blockIdx.x, l = blockIdx.y, b = blockIdx.z;
    const float* row = maps.p[l] + (long)b * L * L + (long)(i + 1) * L + 1;
    float s = 0.f;
    for (int j = threadIdx.x; j < hw; j += 256) s += row[j];
    s = block_sum(s, red);
    if (threadIdx.x == 0) rowsum[((long)b * nmaps + l) * hw + i] = s;
}

// diff[b,l] = -A_l (same ordering as the reference's S - A_l);  wgt[b,l] = keep / (nkeep + 1e-5)
__global__ __launch_bounds__(64) void aff_keep_kernel(const float* __restrict__ rowsum, float* __restrict__ diff,
                                                       float* __restrict__ wgt, int nmaps, int hw) {
    __shared__ float A[12];
    const int b = blockIdx.x, lane = threadIdx.x;
    for (int l = 0; l < nmaps; ++l) {
        const float* rs = rowsum + ((long)b * nmaps + l) * hw;
        float s = 0.f;
        for (int i = lane; i < hw; i += 64) s += rs[i];      // fixed lane-strided order, then a fixed butterfly
        s = wave_sum(s);
        if (lane == 0) A[l] = s;
    }
    __syncthreads();
    if (lane != 0) return;
    float m = 0.f;
    for (int l = 0; l < nmaps; ++l) m += A[l];
    m /= nmaps;
    float n = 0.f;
    for (int l = 0; l < nmaps; ++l) n += (A[l] >= m) ? 1.f : 0.f;
    for (int l = 0; l < nmaps; ++l) {
        diff[b * nmaps + l] = -A[l];
        wgt[b * nmaps + l] = ((A[l] >= m) ? 1.f : 0.f) / (n + 1e-5f);
    }
}

// out[b,j,k] = f( sum_i W[b,i,j] * X[b,i,k] * (sin ? sin[b,i] : 1) )      (W^T x)
// f(v) = recip ? 1/v : alpha * v * (sout ? sout[b,j] : 1) + (add ? add[b,j,k] : 0)
// block = 64 columns x MVC_SLICES row-slices (one wave each: 16 waves per CU keep enough 256-B row reads in
// flight to stream W); X/out are (B, hw, K) with K <= MAXK per launch slice.
#define MVC_SLICES 16
__global__ __launch_bounds__(64 * MVC_SLICES) void matvec_cols_kernel(const float* __restrict__ W, const float* __restrict__ X,
                                                           const float* __restrict__ sin, const float* __restrict__ sout,
                                                           const float* __restrict__ add, float* __restrict__ out,
                                                           int hw, int K, int k0, int recip, float alpha) {
    __shared__ float red[MVC_SLICES][64][MAXK];
    constexpr int NS = MVC_SLICES;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane, b = blockIdx.y;
    const int kn = (K - k0 < MAXK) ? K - k0 : MAXK;
    float acc[MAXK] = {0.f, 0.f, 0.f, 0.f};
    if (j < hw) {
        const float* Wb = W + (long)b * hw * hw + j;
        int i = wv;
        for (; i + 7 * NS < hw; i += 8 * NS) {      // 8 rows per trip: their W loads are issued before the FMAs
            float wr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                wr[u] = Wb[(long)(i + NS * u) * hw] * (sin ? sin[(long)b * hw + i + NS * u] : 1.f);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float* xr = X + ((long)b * hw + i + NS * u) * K + k0;
#pragma unroll
                for (int k = 0; k < MAXK; ++k)
                    if (k < kn) acc[k] = fmaf(wr[u], xr[k], acc[k]);
            }
        }
        for (; i < hw; i += NS) {
            const float wv_ = Wb[(long)i * hw] * (sin ? sin[(long)b * hw + i] : 1.f);
            const float* xr = X + ((long)b * hw + i) * K + k0;
#pragma unroll
            for (int k = 0; k < MAXK; ++k)
                if (k < kn) acc[k] = fmaf(wv_, xr[k], acc[k]);
        }
    }
#pragma unroll
    for (int k = 0; k < MAXK; ++k) red[wv][lane][k] = acc[k];
    __syncthreads();
    if (wv == 0 && j < hw) {
        for (int k = 0; k < kn; ++k) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < MVC_SLICES; ++q) v += red[q][lane][k];
            const long o = ((long)b * hw + j) * K + k0 + k;
            if (recip) v = 1.0f / v;
            else {
                v *= alpha * (sout ? sout[(long)b * hw + j] : 1.f);
                if (add) v += add[o];
            }
            out[o] = v;
        }
    }
}

// out[b,i,k] = f( sum_j W[b,i,j] * X[b,j,k] * (sin ? sin[b,j] : 1) )      (W x), one wave per row
__global__ __launch_bounds__(256) void matvec_rows_kernel(const float* __restrict__ W, const float* __restrict__ X,
                                                           const float* __restrict__ sin, const float* __restrict__ sout,
                                                           const float* __restrict__ add, float* __restrict__ out,
                                                           int hw, int K, int k0, int recip, float alpha) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (i >= hw) return;
    const int kn = (K - k0 < MAXK) ? K - k0 : MAXK;
    float acc[MAXK] = {0.f, 0.f, 0.f, 0.f};
    const float* Wr = W + ((long)b * hw + i) * hw;
    int j = lane;
    for (; j + 192 < hw; j += 256) {       // 4 row segments per trip, loads first
        float wr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) wr[u] = Wr[j + 64 * u] * (sin ? sin[(long)b * hw + j + 64 * u] : 1.f);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* xr = X + ((long)b * hw + j + 64 * u) * K + k0;
#pragma unroll
            for (int k = 0; k < MAXK; ++k)
                if (k < kn) acc[k] = fmaf(wr[u], xr[k], acc[k]);
        }
    }
    for (; j < hw; j += 64) {
        const float wv_ = Wr[j] * (sin ? sin[(long)b * hw + j] : 1.f);
        const float* xr = X + ((long)b * hw + j) * K + k0;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < kn) acc[k] = fmaf(wv_, xr[k], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < MAXK; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0)
        for (int k = 0; k < kn; ++k) {
            float v = acc[k];
            const long o = ((long)b * hw + i) * K + k0 + k;
            if (recip) v = 1.0f / v;
            else {
                v *= alpha * (sout ? sout[(long)b * hw + i] : 1.f);
                if (add) v += add[o];
            }
            out[o] = v;
        }
}

// ---------------------------------------------------------------------------------------------
// Fused sweeps (hw % 4 == 0): a workgroup owns AF_RPB complete rows of one image's W, so ONE read of those rows serves
// a row pass and the following column pass of the Sinkhorn iteration -- r_i = 1 / sum_j W_ij c_j, then the column partial
// sum_{i in block} W_ij r_i with the rows still in registers -- and both halves of T_sym X (the row product and the
// transposed product).  Column partials (one per workgroup) are combined by aff_colreduce_kernel in a fixed order.
// W is read 5 times per batch instead of 10 (1 x with aff_weight's own column partial, 3 x Sinkhorn, 2 x T_sym) and in
// 16-byte pieces; with 16 images it (67 MB) stays in the 256 MiB Infinity Cache across the sweeps.
#define AF_RPB 32          // rows per workgroup of the sweeps
#define AF_RPBW 8          // rows per workgroup of aff_weight_colpart_kernel (it streams 8-12 maps: more workgroups in flight)
#define AF_RG 4            // rows whose loads are in flight together
#define AF_MAXCH 2         // column chunks of 1024 (hw <= 2048)

// W rows + column partial of the FIRST Sinkhorn pass (r = 1):  colpart[(b*nblk + blk)*hw + j] = sum_{i in blk} W[b,i,j]
__global__ __launch_bounds__(256) void aff_weight_colpart_kernel(MapPtrs maps, int nmaps, const float* __restrict__ wgt,
                                                                  const float* __restrict__ seg, float* __restrict__ W,
                                                                  float* __restrict__ colpart, int L) {
    const int hw = L - 1, blk = blockIdx.x, b = blockIdx.y, nblk = gridDim.x;
    const int i0 = blk * AF_RPBW, i1 = min(i0 + AF_RPBW, hw);
    float wl[12];
#pragma unroll
    for (int l = 0; l < 12; ++l) wl[l] = l < nmaps ? wgt[b * nmaps + l] : 0.f;
    for (int j = threadIdx.x; j < hw; j += 256) {
        // all AF_RPBW rows of this column together, two maps per trip: 16 independent 4-byte loads in flight per lane
        float sacc[AF_RPBW];
#pragma unroll
        for (int r = 0; r < AF_RPBW; ++r) sacc[r] = 0.f;
        const long base = (long)b * L * L + (long)(i0 + 1) * L + (j + 1);
#pragma unroll
        for (int l = 0; l < 12; l += 2) {
            if (l >= nmaps) break;
            float v0[AF_RPBW], v1[AF_RPBW];
            const bool two = l + 1 < nmaps;
#pragma unroll
            for (int r = 0; r < AF_RPBW; ++r) {
                const long src = base + (long)(i0 + r < i1 ? r : 0) * L;
                v0[r] = maps.p[l][src];
                v1[r] = two ? maps.p[l + 1][src] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < AF_RPBW; ++r) {
                sacc[r] = fmaf(wl[l], v0[r], sacc[r]);          // same order as aff_weight_kernel: l ascending
                if (two) sacc[r] = fmaf(wl[l + 1], v1[r], sacc[r]);
            }
        }
        float cs = 0.f;
        float sg[AF_RPBW];              // attn_pred factors of the rows (seg-trans mode), requested together
        if (seg) {
#pragma unroll
            for (int r = 0; r < AF_RPBW; ++r) sg[r] = seg[((long)b * hw + min(i0 + r, i1 - 1)) * hw + j];
        }
#pragma unroll
        for (int r = 0; r < AF_RPBW; ++r) {
            if (i0 + r >= i1) break;
            const long dst = ((long)b * hw + i0 + r) * hw + j;
            float v = sacc[r];
            if (seg) v *= sg[r];
            W[dst] = v;
            cs += v;
        }
        colpart[((long)b * nblk + blk) * hw + j] = cs;
    }
}

// MODE 0: r = 1 / (W c) and colpart = sum_i W_ij r_i      (one Sinkhorn row pass + the next column pass)
// MODE 1: r = 1 / (W c) only                              (the last row pass)
// MODE 2: y1[i,k] = r_i sum_j W_ij c_j X[j,k];  colpart[blk][j,k] = sum_i W_ij r_i X[i,k]      (both halves of T_sym X)
template <int MODE, int K>
__global__ __launch_bounds__(256) void aff_sweep_kernel(const float* __restrict__ W, const float* __restrict__ c,
                                                         const float* __restrict__ rin, const float* __restrict__ X,
                                                         float* __restrict__ rout, float* __restrict__ y1,
                                                         float* __restrict__ colpart, int hw) {
    __shared__ float red[2][4][AF_RG][K];
    const int blk = blockIdx.x, b = blockIdx.y, nblk = gridDim.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blk * AF_RPB, i1 = min(i0 + AF_RPB, hw);
    const float* Wb = W + (long)b * hw * hw;
    const float* cb = c + (long)b * hw;
    const int nch = (hw + 1023) / 1024;
    // this thread's columns: 4 per chunk
    float cx[AF_MAXCH][4][K];          // c_j (MODE 0/1)  or  c_j * X[j,k] (MODE 2)
    float ca[AF_MAXCH][4][K];          // column accumulators
    bool okc[AF_MAXCH];
#pragma unroll
    for (int ch = 0; ch < AF_MAXCH; ++ch) {
        const int j = ch * 1024 + tid * 4;
        okc[ch] = ch < nch && j < hw;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                ca[ch][q][k] = 0.f;
                const int jc = okc[ch] ? j + q : 0;          // (unconditional loads, masked: see the row loads below)
                float v = cb[jc];
                if (MODE == 2) v *= X[((long)b * hw + jc) * K + k];
                cx[ch][q][k] = v * (okc[ch] ? 1.f : 0.f);
            }
    }
    int buf = 0;
    for (int ig = i0; ig < i1; ig += AF_RG, buf ^= 1) {
        float4 w[AF_RG][AF_MAXCH];
#pragma unroll
        for (int r = 0; r < AF_RG; ++r)
#pragma unroll
            for (int ch = 0; ch < AF_MAXCH; ++ch) {
                // (unconditional load from a clamped address, masked afterwards: behind a bounds branch hipcc waits for every
                //  load before it issues the next one)
                const bool ok = okc[ch] && ig + r < i1;
                const float4 t = *reinterpret_cast<const float4*>(Wb + (long)min(ig + r, i1 - 1) * hw + (okc[ch] ? ch * 1024 + tid * 4 : 0));
                const float mk = ok ? 1.f : 0.f;            // (a multiplication: a select is turned back into a branch around the load)
                w[r][ch] = make_float4(t.x * mk, t.y * mk, t.z * mk, t.w * mk);
            }
        float rix[AF_RG], xix[AF_RG][K];          // MODE 2: r_i and X[i, k] of the group's rows, requested with the rows of W
        if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < AF_RG; ++r) {
                const long i = (long)b * hw + min(ig + r, i1 - 1);
                rix[r] = rin[i];
#pragma unroll
                for (int k = 0; k < K; ++k) xix[r][k] = X[i * K + k];
            }
        }
        float p[AF_RG][K];
#pragma unroll
        for (int r = 0; r < AF_RG; ++r)
#pragma unroll
            for (int k = 0; k < K; ++k) {
                float a = 0.f;
#pragma unroll
                for (int ch = 0; ch < AF_MAXCH; ++ch)
                    a += w[r][ch].x * cx[ch][0][k] + w[r][ch].y * cx[ch][1][k] + w[r][ch].z * cx[ch][2][k] + w[r][ch].w * cx[ch][3][k];
                p[r][k] = wave_sum(a);
            }
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < AF_RG; ++r)
#pragma unroll
                for (int k = 0; k < K; ++k) red[buf][wave][r][k] = p[r][k];
        }
        __syncthreads();               // (red is double buffered: one barrier per row group)
#pragma unroll
        for (int r = 0; r < AF_RG; ++r) {
            const int i = ig + r;
            if (i >= i1) continue;
            float rv[K];               // MODE 0/1: r_i ; MODE 2: r_i * X[i,k]
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const float dot = red[buf][0][r][k] + red[buf][1][r][k] + red[buf][2][r][k] + red[buf][3][r][k];
                if (MODE == 2) {
                    const float ri = rix[r];
                    if (tid == 0) y1[((long)b * hw + i) * K + k] = ri * dot;
                    rv[k] = ri * xix[r][k];
                } else {
                    rv[k] = 1.0f / dot;
                    if (tid == 0) rout[(long)b * hw + i] = rv[k];
                }
            }
            if (MODE != 1) {
#pragma unroll
                for (int ch = 0; ch < AF_MAXCH; ++ch)
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        ca[ch][0][k] = fmaf(w[r][ch].x, rv[k], ca[ch][0][k]);
                        ca[ch][1][k] = fmaf(w[r][ch].y, rv[k], ca[ch][1][k]);
                        ca[ch][2][k] = fmaf(w[r][ch].z, rv[k], ca[ch][2][k]);
                        ca[ch][3][k] = fmaf(w[r][ch].w, rv[k], ca[ch][3][k]);
                    }
            }
        }
    }
    if (MODE != 1) {
#pragma unroll
        for (int ch = 0; ch < AF_MAXCH; ++ch) {
            if (!okc[ch]) continue;
            const int j = ch * 1024 + tid * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int k = 0; k < K; ++k) colpart[(((long)b * nblk + blk) * hw + j + q) * K + k] = ca[ch][q][k];
        }
    }
}

// mode 0: out[b,j] = 1 / sum_blk colpart[b,blk,j]                                       (column scale of the Sinkhorn pass)
// mode 1: out[b,j,k] = 0.5 * (y1[b,j,k] + c[b,j] * sum_blk colpart[b,blk,j,k])          (T_sym X)
__global__ __launch_bounds__(256) void aff_colreduce_kernel(const float* __restrict__ colpart, const float* __restrict__ y1,
                                                             const float* __restrict__ c, float* __restrict__ out, int hw,
                                                             int K, int nblk, int mode) {
    const int e = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;       // e = j*K + k
    if (e >= hw * K) return;
    // 8 partial rows in flight (the grid is small: hw * K / 256 x B workgroups), summed in a fixed order
    const float* cp = colpart + (long)b * nblk * hw * K + e;
    const long st = (long)hw * K;
    float s = 0.f;
    int blk = 0;
    for (; blk + 8 <= nblk; blk += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = cp[(blk + u) * st];
        s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; blk < nblk; ++blk) s += cp[blk * st];
    if (mode == 0) out[(long)b * hw + e] = 1.0f / s;
    else out[(long)b * hw * K + e] = 0.5f * (y1[(long)b * hw * K + e] + c[(long)b * hw + e / K] * s);
}

// T_sym[b,i,j] = (r_i W_ij c_j + c_i W_ji r_j) / 2   (materialised only for the public compute_trans_mat)
__global__ __launch_bounds__(256) void tsym_kernel(const float* __restrict__ W, const float* __restrict__ r,
                                                    const float* __restrict__ c, float* __restrict__ T, int hw) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
    if (j >= hw) return;
    const float* Wb = W + (long)b * hw * hw;
    const float* rb = r + (long)b * hw;
    const float* cb = c + (long)b * hw;
    T[((long)b * hw + i) * hw + j] = 0.5f * (rb[i] * Wb[(long)i * hw + j] * cb[j] + cb[i] * Wb[(long)j * hw + i] * rb[j]);
}

// One workgroup per pair: box mask of the CAM and V[img, l, slot] = mask ? cam : 0.
// clip/utils.py:115-142 (scoremap2bbox) + clip/clip_tool.py:179-190.
__global__ __launch_bounds__(256) void box_mask_kernel(const float* __restrict__ cam, const int* __restrict__ pair_img,
                                                        const int* __restrict__ pair_slot, float* __restrict__ V,
                                                        float* __restrict__ mask_out, int* __restrict__ boxes,
                                                        int* __restrict__ nbox, int maxbox, int h, int w, int K,
                                                        double thr) {
    extern __shared__ int smi[];   // lab[hw] | xmin[hw] | xmax[hw] | ymin[hw] | ymax[hw] | roots[hw] | misc[4]
    const int hw = h * w, p = blockIdx.x, tid = threadIdx.x;
    int* lab = smi;
    int* xmin = lab + hw;
    int* xmax = xmin + hw;
    int* ymin = xmax + hw;
    int* ymax = ymin + hw;
    int* roots = ymax + hw;
    int* misc = roots + hw;   // [0] max u8, [1] changed flag, [2] root count
    const float* cp = cam + (long)p * hw;
    if (tid == 0) { misc[0] = 0; misc[2] = 0; }
    __syncthreads();
    int mx = 0;
    for (int i = tid; i < hw; i += 256) {
        const int u = (int)(unsigned char)(cp[i] * 255.0f);
        mx = u > mx ? u : mx;
    }
    atomicMax(&misc[0], mx);
    __syncthreads();
    const int theta = (int)(thr * (double)misc[0]);
    for (int i = tid; i < hw; i += 256) {
        const int u = (int)(unsigned char)(cp[i] * 255.0f);
        lab[i] = (u > theta) ? i : -1;
        xmin[i] = w; xmax[i] = -1; ymin[i] = h; ymax[i] = -1;
    }
    __syncthreads();
    // min-label propagation over 8-neighbourhoods until stable
    for (int it = 0; it < hw + 2; ++it) {
        if (tid == 0) misc[1] = 0;
        __syncthreads();
        int changed = 0;
        for (int i = tid; i < hw; i += 256) {
            int l = lab[i];
            if (l < 0) continue;
            const int y = i / w, x = i - y * w;
            int best = l;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    const int yy = y + dy, xx = x + dx;
                    if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                    const int nl = lab[yy * w + xx];
                    if (nl >= 0 && nl < best) best = nl;
                }
            if (best < l) { lab[i] = best; changed = 1; }
        }
        if (changed) misc[1] = 1;
        __syncthreads();
        if (misc[1] == 0) break;
        // pointer jumping: a label is also a pixel index, so lab[lab[i]] is a label of the same component that is at
        // least as small -- the minimum then travels along whole chains per sweep (O(log diameter) sweeps instead of
        // O(diameter)); racing reads only see other valid labels of the component, the fixed point is unchanged
        for (int i = tid; i < hw; i += 256) {
            const int l = lab[i];
            if (l >= 0) {
                const int ll = lab[l];
                if (ll < l) lab[i] = ll;
            }
        }
        __syncthreads();
    }
    // labels may still be chains l -> lab[l] -> ... ; resolve to the root (fixed point lab[r] == r)
    for (int i = tid; i < hw; i += 256) {
        int l = lab[i];
        if (l < 0) continue;
        while (lab[l] != l) l = lab[l];
        const int y = i / w, x = i - y * w;
        atomicMin(&xmin[l], x); atomicMax(&xmax[l], x);
        atomicMin(&ymin[l], y); atomicMax(&ymax[l], y);
    }
    __syncthreads();
    for (int i = tid; i < hw; i += 256)
        if (lab[i] == i) roots[atomicAdd(&misc[2], 1)] = i;
    __syncthreads();
    const int nroot = misc[2];
    if (boxes) {   // estimated_boxes of scoremap2bbox: [x0, y0, min(x0+w, W-1), min(y0+h, H-1)]
        if (tid == 0) nbox[p] = nroot;
        for (int k = tid; k < nroot && k < maxbox; k += 256) {
            const int r = roots[k];
            int* bx = boxes + ((long)p * maxbox + k) * 4;
            bx[0] = xmin[r]; bx[1] = ymin[r];
            bx[2] = (xmax[r] + 1 < w - 1) ? xmax[r] + 1 : w - 1;
            bx[3] = (ymax[r] + 1 < h - 1) ? ymax[r] + 1 : h - 1;
        }
    }
    const int img = pair_img[p], slot = pair_slot[p];
    for (int i = tid; i < hw; i += 256) {
        const int y = i / w, x = i - y * w;
        int inside = 0;
        for (int k = 0; k < nroot && !inside; ++k) {
            const int r = roots[k];
            const int x0 = xmin[r], y0 = ymin[r];
            int x1 = xmax[r] + 1, y1 = ymax[r] + 1;
            x1 = x1 < w - 1 ? x1 : w - 1;
            y1 = y1 < h - 1 ? y1 : h - 1;
            inside = (x >= x0 && x < x1 && y >= y0 && y < y1);
        }
        if (mask_out) mask_out[(long)p * hw + i] = inside ? 1.f : 0.f;
        if (slot >= 0) V[((long)img * hw + i) * K + slot] = inside ? cp[i] : 0.f;      // slot -1: a padding pair (PairPlan pad)
    }
}

// stats[b,k] = (min, max(z - min)) of refined map k of image b;  R is (B, hw, K)
__global__ __launch_bounds__(256) void refined_minmax_kernel(const float* __restrict__ R, float* __restrict__ stats,
                                                              int hw, int K) {
    __shared__ float red[16];
    const int k = blockIdx.x, b = blockIdx.y;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < hw; i += 256) {
        const float v = R[((long)b * hw + i) * K + k];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    mn = block_min(mn, red);
    mx = block_max(mx, red);
    if (threadIdx.x == 0) {
        stats[((long)b * K + k) * 2] = mn;
        stats[((long)b * K + k) * 2 + 1] = mx - mn;
    }
}

// cams[b, 1+k, y, x] = bilinear((R_k - min)/(1e-7 + max)), cams[b, 0] = 1 - max_k   (k < nk[b])
__global__ __launch_bounds__(256) void cam_upsample_kernel(const float* __restrict__ R, const float* __restrict__ stats,
                                                            const int* __restrict__ nk, float* __restrict__ cams,
                                                            int h, int w, int K, int C, int H, int W, float sy, float sx) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (x >= W || y >= H) return;
    float fy = fmaxf(sy * (y + 0.5f) - 0.5f, 0.f), fx = fmaxf(sx * (x + 0.5f) - 0.5f, 0.f);
    int y0 = (int)fy, x0 = (int)fx;
    if (y0 > h - 1) y0 = h - 1;
    if (x0 > w - 1) x0 = w - 1;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0, hy = 1.f - ly, hx = 1.f - lx;
    const long HW = (long)H * W;
    const int n = nk[b];
    float best = -INFINITY;
    const float* Rb = R + (long)b * h * w * K;
    for (int k = 0; k < n; ++k) {
        const float mn = stats[((long)b * K + k) * 2], den = 1e-7f + stats[((long)b * K + k) * 2 + 1];
        const float v00 = (Rb[((long)y0 * w + x0) * K + k] - mn) / den, v01 = (Rb[((long)y0 * w + x1) * K + k] - mn) / den;
        const float v10 = (Rb[((long)y1 * w + x0) * K + k] - mn) / den, v11 = (Rb[((long)y1 * w + x1) * K + k] - mn) / den;
        const float v = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
        cams[((long)b * C + 1 + k) * HW + (long)y * W + x] = v;
        best = fmaxf(best, v);
    }
    for (int k = n; k < C - 1; ++k) cams[((long)b * C + 1 + k) * HW + (long)y * W + x] = 0.f;
    cams[(long)b * C * HW + (long)y * W + x] = 1.0f - best;
}

// ---------------------------------------------------------------------------------------------
extern "C" int wc_aff_weight(const float* const* h_maps, int nmaps, const float* wgt, const float* seg, float* W,
                             int B, int L, void* stream) {
    WC_CHECK_ARG(h_maps && nmaps >= 1 && nmaps <= 12 && wgt && W && B > 0 && L > 1, "wc_aff_weight: bad argument");
    MapPtrs mp;
    for (int i = 0; i < 12; ++i) mp.p[i] = i < nmaps ? h_maps[i] : nullptr;
    const int hw = L - 1;
    hipLaunchKernelGGL(aff_weight_kernel, dim3(wc_cdiv(hw, 256), hw, B), dim3(256), 0, (hipStream_t)stream, mp, nmaps,
                       wgt, seg, W, L);
    WC_LAUNCH_CHECK("aff_weight_kernel");
    return WC_OK;
}

extern "C" int wc_aff_seg_weights(const float* const* h_maps, int nmaps, float* rowsum, float* diff, float* wgt,
                                  int B, int L, void* stream) {
    WC_CHECK_ARG(h_maps && nmaps >= 1 && nmaps <= 12 && rowsum && diff && wgt && B > 0 && L > 1,
                 "wc_aff_seg_weights: bad argument");
    MapPtrs mp;
    for (int i = 0; i < 12; ++i) mp.p[i] = i < nmaps ? h_maps[i] : nullptr;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(aff_rowsum_kernel, dim3(L - 1, nmaps, B), dim3(256), 0, st, mp, nmaps, rowsum, L);
    WC_LAUNCH_CHECK("aff_rowsum_kernel");
    hipLaunchKernelGGL(aff_keep_kernel, dim3(B), dim3(64), 0, st, rowsum, diff, wgt, nmaps, L - 1);
    WC_LAUNCH_CHECK("aff_keep_kernel");
    return WC_OK;
}

extern "C" int wc_matvec(const float* W, const float* X, const float* sin, const float* sout, const float* add,
                         float* out, int B, int hw, int K, int transpose, int recip, float alpha, void* stream) {
    WC_CHECK_ARG(W && X && out && B > 0 && hw > 0 && K > 0 && B <= 65535, "wc_matvec: bad argument");
    WC_CHECK_ARG(out != X && out != sin, "wc_matvec: out must not alias an input vector");
    hipStream_t st = (hipStream_t)stream;
    for (int k0 = 0; k0 < K; k0 += MAXK) {
        if (transpose)
            hipLaunchKernelGGL(matvec_cols_kernel, dim3(wc_cdiv(hw, 64), B), dim3(64 * MVC_SLICES), 0, st, W, X, sin, sout, add,
                               out, hw, K, k0, recip, alpha);
        else
            hipLaunchKernelGGL(matvec_rows_kernel, dim3(wc_cdiv(hw, 4), B), dim3(256), 0, st, W, X, sin, sout, add,
                               out, hw, K, k0, recip, alpha);
        WC_LAUNCH_CHECK("matvec kernel");
    }
    return WC_OK;
}

static bool aff_fused_ok(int hw) { return hw % 4 == 0 && hw <= 1024 * AF_MAXCH && hw >= 4; }

// Whether the fused sweeps apply to this token count (hw % 4 == 0: 16-byte row pieces); else use wc_aff_weight / wc_matvec.
extern "C" int wc_aff_fused_supported(int hw) { return aff_fused_ok(hw) ? 1 : 0; }

// W (B,hw,hw) as wc_aff_weight, plus the first Sinkhorn column scale c1 = 1 / colsum(W) (B,hw).
// ws: B * ceil(hw/8) * hw floats.
extern "C" int wc_aff_weight_c1(const float* const* h_maps, int nmaps, const float* wgt, const float* seg, float* W, float* c1,
                                float* ws, int B, int L, void* stream) {
    WC_CHECK_ARG(h_maps && nmaps >= 1 && nmaps <= 12 && wgt && W && c1 && ws && B > 0 && B <= 65535 && L > 1 && aff_fused_ok(L - 1),
                 "wc_aff_weight_c1: bad argument");
    MapPtrs mp;
    for (int i = 0; i < 12; ++i) mp.p[i] = i < nmaps ? h_maps[i] : nullptr;
    const int hw = L - 1, nblk = wc_cdiv(hw, AF_RPBW);
    hipStream_t st = (hipStream_t)stream;
    const int pr = wc_prof_begin(stream);
    hipLaunchKernelGGL(aff_weight_colpart_kernel, dim3(nblk, B), dim3(256), 0, st, mp, nmaps, wgt, seg, W, ws, L);
    wc_prof_end(pr, "aff_weight_colpart_kernel", (double)B * hw * hw * 4.0 * (nmaps + 1 + (seg ? 1 : 0)), stream);
    WC_LAUNCH_CHECK("aff_weight_colpart_kernel");
    hipLaunchKernelGGL(aff_colreduce_kernel, dim3(wc_cdiv(hw, 256), B), dim3(256), 0, st, ws, nullptr, nullptr, c1, hw, 1, nblk, 0);
    WC_LAUNCH_CHECK("aff_colreduce_kernel");
    return WC_OK;
}

// One fused Sinkhorn step: r = 1 / (W c) and, unless last, c_next = 1 / (W^T r).  ws as above.
extern "C" int wc_aff_sinkhorn_step(const float* W, const float* c, float* r, float* c_next, float* ws, int B, int hw,
                                    int last, void* stream) {
    WC_CHECK_ARG(W && c && r && ws && (last || c_next) && B > 0 && B <= 65535 && aff_fused_ok(hw), "wc_aff_sinkhorn_step: bad argument");
    const int nblk = wc_cdiv(hw, AF_RPB);
    hipStream_t st = (hipStream_t)stream;
    const int pr = wc_prof_begin(stream);
    if (last) hipLaunchKernelGGL((aff_sweep_kernel<1, 1>), dim3(nblk, B), dim3(256), 0, st, W, c, nullptr, nullptr, r, nullptr, ws, hw);
    else hipLaunchKernelGGL((aff_sweep_kernel<0, 1>), dim3(nblk, B), dim3(256), 0, st, W, c, nullptr, nullptr, r, nullptr, ws, hw);
    // algorithmic bytes: a row pass (+ a column pass) over W
    wc_prof_end(pr, "aff_sweep_kernel", (double)B * hw * hw * 4.0 * (last ? 1 : 2), stream);
    WC_LAUNCH_CHECK("aff_sweep_kernel");
    if (!last) {
        hipLaunchKernelGGL(aff_colreduce_kernel, dim3(wc_cdiv(hw, 256), B), dim3(256), 0, st, ws, nullptr, nullptr, c_next, hw, 1, nblk, 0);
        WC_LAUNCH_CHECK("aff_colreduce_kernel");
    }
    return WC_OK;
}

// out (B,hw,K) = T_sym X,  T_sym = (diag(r) W diag(c) + diag(c) W^T diag(r)) / 2, in ONE read of W.  K <= 4.
// y1: workspace B*hw*K; ws: B * ceil(hw/32) * hw * K floats.
extern "C" int wc_aff_tsym_apply(const float* W, const float* r, const float* c, const float* X, float* out, float* y1,
                                 float* ws, int B, int hw, int K, void* stream) {
    WC_CHECK_ARG(W && r && c && X && out && y1 && ws && out != X && B > 0 && B <= 65535 && K >= 1 && K <= 4 && aff_fused_ok(hw),
                 "wc_aff_tsym_apply: bad argument (K <= 4)");
    const int nblk = wc_cdiv(hw, AF_RPB);
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(nblk, B);
    const int pr = wc_prof_begin(stream);
    switch (K) {
        case 1: hipLaunchKernelGGL((aff_sweep_kernel<2, 1>), grid, dim3(256), 0, st, W, c, r, X, nullptr, y1, ws, hw); break;
        case 2: hipLaunchKernelGGL((aff_sweep_kernel<2, 2>), grid, dim3(256), 0, st, W, c, r, X, nullptr, y1, ws, hw); break;
        case 3: hipLaunchKernelGGL((aff_sweep_kernel<2, 3>), grid, dim3(256), 0, st, W, c, r, X, nullptr, y1, ws, hw); break;
        default: hipLaunchKernelGGL((aff_sweep_kernel<2, 4>), grid, dim3(256), 0, st, W, c, r, X, nullptr, y1, ws, hw); break;
    }
    wc_prof_end(pr, "aff_sweep_kernel", (double)B * hw * hw * 4.0 * 2, stream);
    WC_LAUNCH_CHECK("aff_sweep_kernel");
    hipLaunchKernelGGL(aff_colreduce_kernel, dim3(wc_cdiv(hw * K, 256), B), dim3(256), 0, st, ws, y1, c, out, hw, K, nblk, 1);
    WC_LAUNCH_CHECK("aff_colreduce_kernel");
    return WC_OK;
}

extern "C" int wc_tsym(const float* W, const float* r, const float* c, float* T, int B, int hw, void* stream) {
    WC_CHECK_ARG(W && r && c && T && B > 0 && hw > 0, "wc_tsym: bad argument");
    hipLaunchKernelGGL(tsym_kernel, dim3(wc_cdiv(hw, 256), hw, B), dim3(256), 0, (hipStream_t)stream, W, r, c, T, hw);
    WC_LAUNCH_CHECK("tsym_kernel");
    return WC_OK;
}

extern "C" int wc_box_mask(const float* cam, const int* pair_img, const int* pair_slot, float* V, float* mask_out,
                           int* boxes, int* nbox, int maxbox, int P, int h, int w, int K, double thr, void* stream) {
    WC_CHECK_ARG(cam && pair_img && pair_slot && V && P > 0 && h > 0 && w > 0 && K > 0, "wc_box_mask: bad argument");
    WC_CHECK_ARG(!boxes || (nbox && maxbox > 0), "wc_box_mask: boxes needs nbox and maxbox");
    const size_t sm = ((size_t)6 * h * w + 4) * sizeof(int);
    WC_CHECK_ARG(sm <= 160 * 1024, "wc_box_mask: CAM grid too large (h*w <= 6800)");
    hipLaunchKernelGGL(box_mask_kernel, dim3(P), dim3(256), sm, (hipStream_t)stream, cam, pair_img, pair_slot, V,
                       mask_out, boxes, nbox, maxbox, h, w, K, thr);
    WC_LAUNCH_CHECK("box_mask_kernel");
    return WC_OK;
}

extern "C" int wc_cam_upsample(const float* R, const int* nk, float* stats, float* cams, int B, int h, int w, int K,
                               int C, int H, int W, void* stream) {
    WC_CHECK_ARG(R && nk && stats && cams && B > 0 && h > 0 && w > 0 && K > 0 && C >= 2 && C <= K + 1 && H > 0 && W > 0,
                 "wc_cam_upsample: bad argument");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(refined_minmax_kernel, dim3(K, B), dim3(256), 0, st, R, stats, h * w, K);
    WC_LAUNCH_CHECK("refined_minmax_kernel");
    hipLaunchKernelGGL(cam_upsample_kernel, dim3(wc_cdiv(W, 64), wc_cdiv(H, 4), B), dim3(256), 0, st, R, stats, nk,
                       cams, h, w, K, C, H, W, (float)h / H, (float)w / W);
    WC_LAUNCH_CHECK("cam_upsample_kernel");
    return WC_OK;
}
