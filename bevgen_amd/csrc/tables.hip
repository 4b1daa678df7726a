// Setup-time device kernels: derived tables built once in bevgen_finalize (attention-bias matrices, visibility masks,
// weight re-layouts) and the tiny per-step helpers of the Route A decode loop.
#include "common.h"
#include "model.h"

namespace bevgen {

// attn_bias = tril_scatter(camera_bias_emb) + prob_matrix    (mingpt_sparse.py:375-380 / muse_maskgit_pytorch.py:343-348)
// torch.tril_indices(L, L) enumerates (r, c<=r) row-major, so entry (r,c) is element r(r+1)/2 + c of the parameter.
__global__ void build_attn_bias_kernel(const float* __restrict__ emb, const float* __restrict__ prob, float* __restrict__ out, int L) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)L * L) return;
    const int r = (int)(i / L), c = (int)(i % L);
    float v = (emb && c <= r) ? emb[(long)r * (r + 1) / 2 + c] : 0.f;
    if (prob) v += prob[i];
    out[i] = v;
}
void launch_build_attn_bias(const float* emb, const float* prob, float* out, int L, hipStream_t s) {
    hipLaunchKernelGGL(build_attn_bias_kernel, dim3(cdiv((long)L * L, 256)), dim3(256), 0, s, emb, prob, out, L);
    LAUNCH_CHECK();
}

// Route M (muse_maskgit_pytorch.py:150-156): self  bias = pad_left0(attn_bias[K:, K:]),  cross bias = pad_left0(attn_bias[K:, :K]);
// columns beyond the real keys get -1e30 so that the flash kernel needs no key-bound test.
__global__ void build_muse_bias_kernel(const float* __restrict__ ab, int L, int K, int N, float* __restrict__ bs, int ldS, float* __restrict__ bc, int ldC) {
    const int q = blockIdx.x;
    for (int j = threadIdx.x; j < ldS; j += blockDim.x)
        bs[(long)q * ldS + j] = j == 0 ? 0.f : (j <= N ? ab[(long)(K + q) * L + K + j - 1] : kNegBig);
    for (int j = threadIdx.x; j < ldC; j += blockDim.x)
        bc[(long)q * ldC + j] = j == 0 ? 0.f : (j <= K ? ab[(long)(K + q) * L + j - 1] : kNegBig);
}
void launch_build_muse_bias(const float* ab, int L, int K, int N, float* bs, int ldS, float* bc, int ldC, hipStream_t s) {
    hipLaunchKernelGGL(build_muse_bias_kernel, dim3(N), dim3(256), 0, s, ab, L, K, N, bs, ldS, bc, ldC);
    LAUNCH_CHECK();
}

// Route A visibility = element mask x block layout (sparse_self_attention.py:63-85, 153-173), kept as its factors (kernels.h SparseVis)
__global__ void build_allowed_kernel(const float* __restrict__ mask, uint8_t* __restrict__ out, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = mask[i] != 0.f ? 1 : 0;
}
void launch_build_allowed(const float* mask, uint8_t* out, long n, hipStream_t s) {
    hipLaunchKernelGGL(build_allowed_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, mask, out, n);
    LAUNCH_CHECK();
}
// layout int64 [heads, nb, nb] -> lay uint8 [heads, nb, nb] and, per (head, block row), chunks[0] = count, chunks[1..] = ascending ids of the 16-key chunks that
// overlap a present block (chunks_ld >= ceil(L / 16) + 1).  One workgroup per (head, block row): threads test chunks, one thread compacts (<= 1024 chunks).
__global__ __launch_bounds__(256) void build_layout_kernel(const int64_t* __restrict__ layout, uint8_t* __restrict__ lay, uint16_t* __restrict__ chunks, int heads, int nb, int blk, int L, int chunks_ld) {
    __shared__ uint8_t present[1024];
    const int i = blockIdx.x;   // (head, block row)
    const int64_t* src = layout + (long)i * nb;
    uint8_t* dst = lay + (long)i * nb;
    for (int j = threadIdx.x; j < nb; j += blockDim.x) dst[j] = src[j] != 0 ? 1 : 0;
    const int nc = (L + 15) / 16;
    for (int c = threadIdx.x; c < nc; c += blockDim.x) {
        bool any = false;
        for (int k = c * 16; k < min(L, c * 16 + 16) && !any; k += (blk < 16 ? blk : 16)) any = src[k / blk] != 0;
        present[c] = any ? 1 : 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint16_t* out = chunks + (long)i * chunks_ld;
        int cnt = 0;
        for (int c = 0; c < nc; ++c)
            if (present[c]) out[1 + cnt++] = (uint16_t)c;
        out[0] = (uint16_t)cnt;
    }
}
void launch_build_layout(const int64_t* layout, uint8_t* lay, uint16_t* chunks, int heads, int nb, int blk, int L, int chunks_ld, hipStream_t s) {
    BG_REQUIRE((L + 15) / 16 <= 1024, "build_layout: at most 1024 key chunks per row (L = %d)", L);
    hipLaunchKernelGGL(build_layout_kernel, dim3(heads * nb), dim3(256), 0, s, layout, lay, chunks, heads, nb, blk, L, chunks_ld);
    LAUNCH_CHECK();
}

// out[h][r][c] = visible(h, r, c) ? scale * add[r][c] : -1e30   (c >= cols -> -1e30): bias operand of the flash kernel for Route A (prefill, operator seam)
__global__ void build_masked_bias_kernel(const float* __restrict__ add, SparseVis v, float* __restrict__ out, int rows, int cols, int ldout, int ldadd, float scale) {
    const int r = blockIdx.x, h = blockIdx.y;
    const uint8_t* arow = v.allowed + (long)h * v.allowed_head_stride + (long)r * v.ldallowed;
    const uint8_t* lrow = v.lay + (long)h * v.lay_head_stride + (long)(r / v.blk) * v.nb;
    for (int c = threadIdx.x; c < ldout; c += blockDim.x) {
        float val = kNegBig;
        if (c < cols && (!v.has_allowed || arow[c]) && (!v.has_lay || lrow[c / v.blk])) val = add ? scale * add[(long)r * ldadd + c] : 0.f;
        out[((long)h * rows + r) * ldout + c] = val;
    }
}
void launch_build_masked_bias(const float* add, const SparseVis& v0, float* out, int heads, int rows, int cols, int ldout, int ldadd, float scale, hipStream_t s) {
    const SparseVis v = vis_fix(v0, out);
    hipLaunchKernelGGL(build_masked_bias_kernel, dim3(rows, heads), dim3(256), 0, s, add, v, out, rows, cols, ldout, ldadd, scale);
    LAUNCH_CHECK();
}

// Route A decode: x[b,:] = (x_tok_emb[tok[b]] + img_embed[b, cam, cell]) + x_pos_emb[j],  j = forward_shuffle_idx[step]
// (mingpt_sparse.py:331-365 restricted to the one new row; the step lives on the device so the launch is graph-replayable)
__global__ __launch_bounds__(256) void ar_step_embed_kernel(const int64_t* __restrict__ tok, const float* __restrict__ tok_emb, const float* __restrict__ img_embed,
                                                            const float* __restrict__ pos_emb, const int64_t* __restrict__ fwd_idx, const int* __restrict__ d_step,
                                                            float* __restrict__ x, int C, int T, int D, int vocab_rows) {
    const int b = blockIdx.x;
    const long j = fwd_idx[*d_step];
    long id = tok[b];
    id = id < 0 ? 0 : (id >= vocab_rows ? vocab_rows - 1 : id);
    for (int o = threadIdx.x; o < D; o += 256) {
        float v = tok_emb[id * D + o];
        if (img_embed) v += img_embed[((long)b * C * T + j) * D + o];
        x[(long)b * D + o] = v + pos_emb[j * D + o];
    }
}
void launch_ar_step_embed(const int64_t* tok, const float* tok_emb, const float* img_embed, const float* pos_emb, const int64_t* fwd_idx, const int* d_step,
                          float* x, int B, int C, int T, int D, int vocab_rows, hipStream_t s) {
    hipLaunchKernelGGL(ar_step_embed_kernel, dim3(B), dim3(256), 0, s, tok, tok_emb, img_embed, pos_emb, fwd_idx, d_step, x, C, T, D, vocab_rows);
    LAUNCH_CHECK();
}

// out[b, fwd_idx[step]] = tok[b]   (x[:, i, k] = ix, cond_transformer_multi_view.py:219)
__global__ void store_tokens_kernel(const int64_t* __restrict__ tok, const int64_t* __restrict__ fwd_idx, const int* __restrict__ d_step, int64_t* __restrict__ out, int B, int N) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) out[(long)b * N + fwd_idx[*d_step]] = tok[b];
}
void launch_store_tokens(const int64_t* tok, const int64_t* fwd_idx, const int* d_step, int64_t* out, int B, int N, hipStream_t s) {
    hipLaunchKernelGGL(store_tokens_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, tok, fwd_idx, d_step, out, B, N);
    LAUNCH_CHECK();
}

__global__ void gather_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int row, int rows_per_batch, int D) {
    const int b = blockIdx.x;
    for (int o = threadIdx.x; o < D; o += blockDim.x) out[(long)b * D + o] = x[((long)b * rows_per_batch + row) * D + o];
}
void launch_gather_rows(const float* x, float* out, int B, int row, int rows_per_batch, int D, hipStream_t s) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3(B), dim3(256), 0, s, x, out, row, rows_per_batch, D);
    LAUNCH_CHECK();
}

// Shared condition prefix (BASELINE config 5: several samples per BEV layout): copy the first `rows` cache rows of sequence slot `src` to the
// slots [dst0, dst0 + count) of every layer and head; cache layout [layer][B][H][L][64] (elem_bytes per element), src itself is skipped.
__global__ void replicate_prefix_kernel(char* __restrict__ kc, char* __restrict__ vc, int B, int H, int L, int rows, int src, int dst0, int count, int elem_bytes) {
    const int layer = blockIdx.z, h = blockIdx.y;
    const long row_b = 64L * elem_bytes;
    const long blk = ((long)layer * B * H + h) * L * row_b;                 // + slot * H * L * row_b
    const long slot_b = (long)H * L * row_b;
    const long n16 = (long)rows * row_b / 16;
    for (int j = 0; j < count; ++j) {
        const int d = dst0 + j;
        if (d == src) continue;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long)gridDim.x * blockDim.x) {
            reinterpret_cast<uint4*>(kc + blk + d * slot_b)[i] = reinterpret_cast<const uint4*>(kc + blk + src * slot_b)[i];
            reinterpret_cast<uint4*>(vc + blk + d * slot_b)[i] = reinterpret_cast<const uint4*>(vc + blk + src * slot_b)[i];
        }
    }
}
void launch_replicate_prefix(void* kc, void* vc, int layers, int B, int H, int L, int rows, int src, int dst0, int count, int elem_bytes, hipStream_t s) {
    hipLaunchKernelGGL(replicate_prefix_kernel, dim3(8, H, layers), dim3(256), 0, s, reinterpret_cast<char*>(kc), reinterpret_cast<char*>(vc), B, H, L, rows, src, dst0, count,
                       elem_bytes);
    LAUNCH_CHECK();
}

// decode_weights = f16: the matrix is replaced by its fp16-representable rounding (so that every consumer - prefill GEMMs, split planes, decode kernels - sees
// the same values) and a packed fp16 copy is written for the kernels that stream it
__global__ void round_to_f16_kernel(float* __restrict__ w, _Float16* __restrict__ h, long n, unsigned* __restrict__ status) {
    unsigned bad = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const _Float16 v = (_Float16)w[i];
        guard_half(v, bad);
        w[i] = (float)v;
        if (h) h[i] = v;
    }
    if (bad) status_raise(status, BG_ST_F16_RANGE);   // a weight with |w| >= 65520 has no fp16 image (reported by bevgen_finalize)
}
void launch_round_to_f16(float* w, void* h, long n, hipStream_t s) {
    hipLaunchKernelGGL(round_to_f16_kernel, dim3((int)std::min<long>((n + 255) / 256, 8192)), dim3(256), 0, s, w, reinterpret_cast<_Float16*>(h), n, status_current());
    LAUNCH_CHECK();
}

// Consumer-side constants of a LayerNorm folded into the projection behind it (GemmArgs::ln_in_*): Wg = W o gamma (columns scaled; columns >= Kg - the zero padding of a
// padded matrix - stay 0) and cs[n] = sum_k Wg[n][k] in fp64.  One wave per row.
__global__ __launch_bounds__(256) void ln_fold_weight_kernel(const float* __restrict__ W, const float* __restrict__ gamma, float* __restrict__ Wg, float* __restrict__ cs, int N,
                                                             int K, int Kg) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    double c = 0.0;
    for (int k = lane; k < K; k += 64) {
        const float v = k < Kg ? W[(long)n * K + k] * gamma[k] : 0.f;
        Wg[(long)n * K + k] = v;
        c += (double)v;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) cs[n] = (float)c;
}
void launch_ln_fold_weight(const float* W, const float* gamma, float* Wg, float* cs, int N, int K, int Kg, hipStream_t s) {
    hipLaunchKernelGGL(ln_fold_weight_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, W, gamma, Wg, cs, N, K, Kg);
    LAUNCH_CHECK();
}

// Row order of the GEGLU up-projection weight for the fused epilogue (EPI_GEGLU): W [2F, D] (rows 0..F-1 = x, F..2F-1 = gate, muse_net:74) ->
// Wo [2 Fpad, D], 64-row groups: group (t, wn) = [x rows 64t + 32wn .. +31 | gate rows of the same 32 outputs]; rows of outputs >= F are zero
__global__ void geglu_weight_order_kernel(const float* __restrict__ W, float* __restrict__ Wo, int F, int Fpad, int D) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)2 * Fpad * D;
    if (i >= total) return;
    const int col = (int)(i % D);
    const int row = (int)(i / D);              // output row of Wo
    const int grp = row >> 6, within = row & 63;
    const int is_gate = within >> 5, f = grp * 32 + (within & 31);   // grp = 2 t + wn  ->  outputs 32 grp .. 32 grp + 31
    Wo[i] = f < F ? W[((long)is_gate * F + f) * D + col] : 0.f;
}
void launch_geglu_weight_order(const float* W, float* Wo, int F, int Fpad, int D, hipStream_t s) {
    const long total = (long)2 * Fpad * D;
    hipLaunchKernelGGL(geglu_weight_order_kernel, dim3(cdiv(total, 256)), dim3(256), 0, s, W, Wo, F, Fpad, D);
    LAUNCH_CHECK();
}

__global__ void increment_kernel(int* p) { *p += 1; }
void launch_increment(int* p, hipStream_t s) {
    hipLaunchKernelGGL(increment_kernel, dim3(1), dim3(1), 0, s, p);
    LAUNCH_CHECK();
}

// conv kernels: [Cout][Cin][kh][kw] (torch) -> [Cout][kh][kw][Cin] (K-contiguous over (tap, channel) for the implicit GEMM)
__global__ void relayout_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ o, int cout, int cin, int kh, int kw) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)cout * cin * kh * kw;
    if (i >= total) return;
    const int ci = (int)(i % cin);
    long t = i / cin;
    const int x = (int)(t % kw); t /= kw;
    const int y = (int)(t % kh);
    const int co = (int)(t / kh);
    o[i] = w[(((long)co * cin + ci) * kh + y) * kw + x];
}
void launch_relayout_conv_weight(const float* w, float* o, int cout, int cin, int kh, int kw, hipStream_t s) {
    hipLaunchKernelGGL(relayout_conv_weight_kernel, dim3(cdiv((long)cout * cin * kh * kw, 256)), dim3(256), 0, s, w, o, cout, cin, kh, kw);
    LAUNCH_CHECK();
}

__global__ void fuse_qkv_kernel(const float* wq, const float* wk, const float* wv, const float* bq, const float* bk, const float* bv, float* w, float* b, int D) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long dd = (long)D * D;
    if (i < dd) { w[i] = wq[i]; w[dd + i] = wk[i]; w[2 * dd + i] = wv[i]; }
    if (i < D) { b[i] = bq[i]; b[D + i] = bk[i]; b[2 * D + i] = bv[i]; }
}
void launch_fuse_qkv(const float* wq, const float* wk, const float* wv, const float* bq, const float* bk, const float* bv, float* w, float* b, int D, hipStream_t s) {
    hipLaunchKernelGGL(fuse_qkv_kernel, dim3(cdiv((long)D * D, 256)), dim3(256), 0, s, wq, wk, wv, bq, bk, bv, w, b, D);
    LAUNCH_CHECK();
}

__global__ void pad_rows_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int ld_dst, int cols) {
    const int r = blockIdx.x;
    for (int c = threadIdx.x; c < ld_dst; c += blockDim.x) dst[(long)r * ld_dst + c] = c < cols ? src[(long)r * ld_src + c] : 0.f;
}
void launch_pad_rows(const float* src, int ld_src, float* dst, int ld_dst, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(pad_rows_kernel, dim3(rows), dim3(256), 0, s, src, ld_src, dst, ld_dst, cols);
    LAUNCH_CHECK();
}

__global__ void fill_i64_kernel(int64_t* p, long n, int64_t v) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
void launch_fill_i64(int64_t* p, long n, int64_t v, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(fill_i64_kernel, dim3((int)std::min<long>((n + 255) / 256, 4096)), dim3(256), 0, s, p, n, v);
    LAUNCH_CHECK();
}

}  // namespace bevgen
