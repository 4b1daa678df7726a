// Embedding assembly and q/k/v layout kernels (all bandwidth-bound, one wave or one block per token row).
//
// Reference arithmetic:
//   geometric image embedding  mingpt_sparse.py:336-349 == muse_maskgit_pytorch.py:314-327
//   BEV / condition embedding  mingpt_sparse.py:352-358 == muse_maskgit_pytorch.py:333-340
//   Route M q/k/v preparation  muse_maskgit_pytorch.py:132-146 (x8, null-kv concat, l2norm eps 1e-12, q_scale/k_scale)
#include "common.h"
#include "kernels.h"

namespace bevgen {

// ---------------------------------------------------------------------------------------------- camera embeddings
// one block (256 threads) per (b, cam, t); t == T computes c_embed[b,cam,:] only.
__global__ __launch_bounds__(256) void camera_embed_kernel(const float* __restrict__ I_inv, const float* __restrict__ E_inv, const float* __restrict__ plane,
                                                           const float* __restrict__ Wimg, const float* __restrict__ Wcam, float* __restrict__ img,
                                                           float* __restrict__ c_embed, int C, int T, int D) {
    __shared__ float red[4];
    const int t = blockIdx.x, cam = blockIdx.y, b = blockIdx.z;
    const float* Ii = I_inv + ((long)b * C + cam) * 9;
    const float* Ei = E_inv + ((long)b * C + cam) * 16;
    const float c4[4] = {Ei[3], Ei[7], Ei[11], Ei[15]};  // E_inv[..., :, 3]
    if (t == T) {
        for (int o = threadIdx.x; o < D; o += 256) {
            const float* w = Wcam + o * 4;
            c_embed[((long)b * C + cam) * D + o] = ((w[0] * c4[0] + w[1] * c4[1]) + w[2] * c4[2]) + w[3] * c4[3];
        }
        return;
    }
    const float px = plane[t], py = plane[T + t], pz = plane[2 * T + t];
    float camv[4];
#pragma unroll
    for (int r = 0; r < 3; ++r) camv[r] = (Ii[3 * r] * px + Ii[3 * r + 1] * py) + Ii[3 * r + 2] * pz;
    camv[3] = 1.f;
    float d[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) d[r] = ((Ei[4 * r] * camv[0] + Ei[4 * r + 1] * camv[1]) + Ei[4 * r + 2] * camv[2]) + Ei[4 * r + 3] * camv[3];
    // e[o] = Wimg[o,:].d - Wcam[o,:].c ; out = e / (||e|| + 1e-7)
    float ss = 0.f;
    float* out = img + (((long)b * C + cam) * T + t) * D;
    for (int o = threadIdx.x; o < D; o += 256) {
        const float* wi = Wimg + o * 4;
        const float* wc = Wcam + o * 4;
        const float de = ((wi[0] * d[0] + wi[1] * d[1]) + wi[2] * d[2]) + wi[3] * d[3];
        const float ce = ((wc[0] * c4[0] + wc[1] * c4[1]) + wc[2] * c4[2]) + wc[3] * c4[3];
        const float e = de - ce;
        out[o] = e;
        ss = fmaf(e, e, ss);
    }
    ss = wave_sum(ss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    const float nrm = sqrtf((red[0] + red[1]) + (red[2] + red[3])) + 1e-7f;
    for (int o = threadIdx.x; o < D; o += 256) out[o] = out[o] / nrm;
}

void launch_camera_embed(const float* I_inv, const float* E_inv, const float* plane, const float* Wimg, const float* Wcam, float* img, float* c_embed,
                         int B, int C, int T, int D, hipStream_t s) {
    hipLaunchKernelGGL(camera_embed_kernel, dim3(T + 1, C, B), dim3(256), 0, s, I_inv, E_inv, plane, Wimg, Wcam, img, c_embed, C, T, D);
    LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------- condition (BEV) embedding
__global__ __launch_bounds__(256) void cond_embed_kernel(const int64_t* __restrict__ cond_ids, const float* __restrict__ cond_tok, const float* __restrict__ cond_pos,
                                                         const float* __restrict__ bev_grid, const float* __restrict__ Wbev, const float* __restrict__ bbev,
                                                         const float* __restrict__ bev_cam_pos, const float* __restrict__ c_embed, float* __restrict__ out,
                                                         int C, int K, int D, int cond_vocab) {
    const int k = blockIdx.x, b = blockIdx.y;
    long id = cond_ids[(long)b * K + k];
    id = id < 0 ? 0 : (id >= cond_vocab ? cond_vocab - 1 : id);
    const float gx = bev_grid ? bev_grid[k] : 0.f, gy = bev_grid ? bev_grid[K + k] : 0.f;
    for (int o = threadIdx.x; o < D; o += 256) {
        float v = cond_tok[id * D + o];
        if (bev_grid) {
            const float grid_embed = (gx * Wbev[2 * o] + gy * Wbev[2 * o + 1]) + bbev[o];
            float cams = 0.f;
            for (int c = 0; c < C; ++c) cams += bev_cam_pos[((long)c * K + k) * D + o] + c_embed[((long)b * C + c) * D + o];
            v += grid_embed - cams;
        }
        out[((long)b * K + k) * D + o] = v + cond_pos[(long)k * D + o];
    }
}

void launch_cond_embed(const int64_t* cond_ids, const float* cond_tok, const float* cond_pos, const float* bev_grid, const float* Wbev, const float* bbev,
                       const float* bev_cam_pos, const float* c_embed, float* out, int B, int C, int K, int D, int cond_vocab, hipStream_t s) {
    hipLaunchKernelGGL(cond_embed_kernel, dim3(K, B), dim3(256), 0, s, cond_ids, cond_tok, cond_pos, bev_grid, Wbev, bbev, bev_cam_pos, c_embed, out, C, K, D, cond_vocab);
    LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------- token embedding
__global__ __launch_bounds__(256) void token_embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ tok_emb, const float* __restrict__ img,
                                                          const float* __restrict__ pos, float* __restrict__ x, int N, int D, int vocab_rows) {
    const int n = blockIdx.x, b = blockIdx.y;
    long id = ids[(long)b * N + n];
    id = id < 0 ? 0 : (id >= vocab_rows ? vocab_rows - 1 : id);
    const long row = (long)b * N + n;
    for (int o4 = threadIdx.x; o4 < D / 4; o4 += 256) {
        float4 v = reinterpret_cast<const float4*>(tok_emb + id * D)[o4];
        if (img) {
            const float4 g = reinterpret_cast<const float4*>(img + row * D)[o4];
            v.x += g.x; v.y += g.y; v.z += g.z; v.w += g.w;
        }
        const float4 p = reinterpret_cast<const float4*>(pos + (long)n * D)[o4];
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        reinterpret_cast<float4*>(x + row * D)[o4] = v;
    }
}

void launch_token_embed(const int64_t* ids, const float* tok_emb, const float* img, const float* pos, float* x, int B, int N, int D, int vocab_rows, hipStream_t s) {
    BG_REQUIRE(D % 4 == 0, "token_embed: D must be a multiple of 4");
    hipLaunchKernelGGL(token_embed_kernel, dim3(N, B), dim3(256), 0, s, ids, tok_emb, img, pos, x, N, D, vocab_rows);
    LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------- Route M q / kv preparation
// one wave per (row, head): lane = head-dim element.  F.normalize: x / max(||x||, 1e-12)
__global__ __launch_bounds__(256) void muse_q_prep_kernel(const float* __restrict__ qraw, const float* __restrict__ q_scale, float* __restrict__ Q, int H, int Nq, long total) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b*Nq + n)*H + h
    if (w >= total) return;
    const int h = (int)(w % H);
    const long bn = w / H;
    const int n = (int)(bn % Nq);
    const long b = bn / Nq;
    const float v = qraw[bn * (H * 64) + h * 64 + lane] * 8.0f;  // q = q * self.scale (muse_net:134)
    const float nrm = fmaxf(sqrtf(wave_sum(v * v)), 1e-12f);
    Q[((b * H + h) * Nq + n) * 64 + lane] = (v / nrm) * q_scale[lane];
}

void launch_muse_q_prep(const float* qraw, const float* q_scale, float* Q, int B, int H, int Nq, hipStream_t s) {
    const long total = (long)B * Nq * H;
    hipLaunchKernelGGL(muse_q_prep_kernel, dim3((int)((total + 3) / 4)), dim3(256), 0, s, qraw, q_scale, Q, H, Nq, total);
    LAUNCH_CHECK();
}

__global__ __launch_bounds__(256) void muse_kv_prep_kernel(const float* __restrict__ kvraw, const float* __restrict__ null_kv, const float* __restrict__ k_scale,
                                                           float* __restrict__ K, float* __restrict__ V, int H, int Nk, int Nk_pad, long total) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b*(Nk+1) + j)*H + h ; j = 0 is the null key/value
    if (w >= total) return;
    const int h = (int)(w % H);
    const long bj = w / H;
    const int j = (int)(bj % (Nk + 1));
    const long b = bj / (Nk + 1);
    float kv, vv;
    if (j == 0) {
        kv = null_kv[h * 64 + lane];
        vv = null_kv[(long)H * 64 + h * 64 + lane];
    } else {
        const float* src = kvraw + (b * Nk + (j - 1)) * (2L * H * 64);
        kv = src[h * 64 + lane];
        vv = src[H * 64 + h * 64 + lane];
    }
    const float nrm = fmaxf(sqrtf(wave_sum(kv * kv)), 1e-12f);
    const long dst = ((b * H + h) * Nk_pad + j) * 64 + lane;
    K[dst] = (kv / nrm) * k_scale[lane];
    V[dst] = vv;
}

void launch_muse_kv_prep(const float* kvraw, const float* null_kv, const float* k_scale, float* K, float* V, int B, int H, int Nk, int Nk_pad, hipStream_t s) {
    const long total = (long)B * (Nk + 1) * H;
    hipLaunchKernelGGL(muse_kv_prep_kernel, dim3((int)((total + 3) / 4)), dim3(256), 0, s, kvraw, null_kv, k_scale, K, V, H, Nk, Nk_pad, total);
    LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------- Route A q/k/v scatter into the KV cache
__device__ __forceinline__ void cache_store(void* cache, int kv_dtype, long idx, float v, unsigned& bad) {
    if (kv_dtype == 0) reinterpret_cast<float*>(cache)[idx] = v;
    else {
        const _Float16 h = (_Float16)v;   // fp16 storage (round to nearest even)
        guard_half(h, bad);               // (a key / value outside the fp16 range: BG_ST_F16_RANGE)
        reinterpret_cast<_Float16*>(cache)[idx] = h;
    }
}

__global__ __launch_bounds__(256) void ar_qkv_scatter_kernel(const float* __restrict__ qkv, float* __restrict__ Q, void* kcache, void* vcache, int kv_dtype,
                                                             int H, int n, int pos0, const int* __restrict__ d_pos, int Lmax, long total, unsigned* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // (b*n + i)*H + h
    if (w >= total) return;
    const int h = (int)(w % H);
    const long bi = w / H;
    const int i = (int)(bi % n);
    const long b = bi / n;
    const float* src = qkv + bi * (3L * H * 64) + h * 64 + lane;
    const int pos = (d_pos ? *d_pos : 0) + pos0 + i;
    if (Q) Q[((b * H + h) * n + i) * 64 + lane] = src[0];
    const long dst = ((b * H + h) * Lmax + pos) * 64 + lane;
    unsigned bad = 0;
    cache_store(kcache, kv_dtype, dst, src[H * 64], bad);
    cache_store(vcache, kv_dtype, dst, src[2 * H * 64], bad);
    if (bad) status_raise(status, BG_ST_F16_RANGE);
}

void launch_ar_qkv_scatter(const float* qkv, float* Q, void* kcache, void* vcache, int kv_dtype, int B, int H, int n, int pos0, int Lmax, hipStream_t s) {
    const long total = (long)B * n * H;
    hipLaunchKernelGGL(ar_qkv_scatter_kernel, dim3((int)((total + 3) / 4)), dim3(256), 0, s, qkv, Q, kcache, vcache, kv_dtype, H, n, pos0, (const int*)nullptr, Lmax, total, status_current());
    LAUNCH_CHECK();
}

void launch_ar_kv_append(const float* qkv, void* kcache, void* vcache, int kv_dtype, int B, int H, int pos0, const int* d_pos, int Lmax, hipStream_t s) {
    const long total = (long)B * H;
    hipLaunchKernelGGL(ar_qkv_scatter_kernel, dim3((int)((total + 3) / 4)), dim3(256), 0, s, qkv, (float*)nullptr, kcache, vcache, kv_dtype, H, 1, pos0, d_pos, Lmax, total, status_current());
    LAUNCH_CHECK();
}

}  // namespace bevgen
