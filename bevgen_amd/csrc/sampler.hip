// Token sampling kernels.
//
//   MaskGit (Route M): re-masking by critic score, top-k filtered gumbel-argmax, self-critic scores
//                      muse_maskgit_pytorch.py:443-458 (helpers), :564-611 (generate loop)
//   Autoregressive (Route A): temperature, top-k (ties kept), softmax, greedy / inverse-CDF draw
//                      cond_transformer_multi_view.py:138-142, 204-219
// One wave per vocabulary row; the k-th largest logit is found by a 32-step bitwise search on the order-preserving integer image
// of the floats (no sort, no LDS), counting with wave reductions.
#include "common.h"
#include "kernels.h"

namespace bevgen {

constexpr int VPL_MAX = 16;  // vocabulary <= 1024 (64 lanes x 16)

__device__ __forceinline__ uint32_t ordered_key(float x) {  // monotone float -> uint32
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// k-th largest key of the wave's values (ties counted): largest t with count(key >= t) >= k
__device__ __forceinline__ uint32_t kth_largest_key(const uint32_t (&key)[VPL_MAX], const bool (&valid)[VPL_MAX], int k) {
    uint32_t t = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = t | (1u << bit);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < VPL_MAX; ++j) cnt += (valid[j] && key[j] >= cand) ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        if (cnt >= k) t = cand;
    }
    return t;
}

__device__ __forceinline__ void wave_argmax(float& v, int& idx) {  // largest value, lowest index on ties
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(idx, o, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
}

// ---------------------------------------------------------------------------------------------- re-masking
__global__ __launch_bounds__(256) void remask_kernel(int64_t* __restrict__ ids, const float* __restrict__ scores, const int64_t* __restrict__ init_ids,
                                                     int T, int n_mask, int64_t mask_id) {
    extern __shared__ float sc[];
    const int row = blockIdx.x;
    for (int i = threadIdx.x; i < T; i += blockDim.x) sc[i] = scores[(long)row * T + i];
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        const float s = sc[i];
        int rank = 0;
        for (int j = 0; j < T; ++j) rank += (sc[j] > s || (sc[j] == s && j < i)) ? 1 : 0;
        int64_t v = ids[(long)row * T + i];
        if (rank < n_mask) v = mask_id;
        if (init_ids) {
            const int64_t iv = init_ids[(long)row * T + i];
            if (iv != mask_id) v = iv;
        }
        ids[(long)row * T + i] = v;
    }
}

void launch_remask(int64_t* ids, const float* scores, const int64_t* init_ids, int rows, int T, int n_mask, int64_t mask_id, hipStream_t s) {
    hipLaunchKernelGGL(remask_kernel, dim3(rows), dim3(256), T * sizeof(float), s, ids, scores, init_ids, T, n_mask, mask_id);
    LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------- MaskGit token pick
// conf_mode (scores without a token critic, muse_net:611-622): 0 = off; 1 = scores[row] = 1 - softmax(logits)[pred] at the positions that were masked, -1e5 elsewhere
// (can_remask_prev_masked = False); 2 = 1 - softmax(logits)[pred] everywhere, pred drawn for EVERY position (can_remask_prev_masked = True)
__global__ __launch_bounds__(256) void maskgit_pick_kernel(int64_t* __restrict__ ids, const float* __restrict__ logits, int ldl, const float* __restrict__ gumbel_u,
                                                           long rows, int V, int k, float temp_div, int64_t mask_id, unsigned long long seed, unsigned iter,
                                                           float* __restrict__ conf_scores, int conf_mode, unsigned* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const bool was_mask = ids[row] == mask_id;
    if (!was_mask) {  // only masked positions are replaced (muse_net:593-599)
        if (conf_mode == 1 && lane == 0) conf_scores[row] = -1e5f;
        if (conf_mode != 2) return;
    }
    const float* lr = logits + row * ldl;
    float x[VPL_MAX];
    uint32_t key[VPL_MAX];
    bool valid[VPL_MAX];
#pragma unroll
    for (int j = 0; j < VPL_MAX; ++j) {
        const int i = lane + 64 * j;
        valid[j] = i < V;
        x[j] = valid[j] ? lr[i] : 0.f;
        key[j] = ordered_key(x[j]);
    }
    {   // the reference's Route A asserts finite logits on the host every step (ar_lm:202, gpt:388); here the sampler that reads every logit of the row anyway flags it
        bool nf = false;
#pragma unroll
        for (int j = 0; j < VPL_MAX; ++j) nf |= nonfinite(x[j]);
        if (nf) status_raise(status, BG_ST_NONFINITE_LOGITS);
    }
    const bool noisy = gumbel_u != nullptr || seed != 0;   // explicit uniforms, or drawn in registers (Philox keyed by seed / iteration / element)
    uint32_t thr = 0;
    if (noisy) thr = kth_largest_key(key, valid, k);  // without noise the arg-max is unaffected by the top-k filter
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    Philox4 ph{};
#pragma unroll
    for (int j = 0; j < VPL_MAX; ++j) {
        const int i = lane + 64 * j;
        if (!valid[j]) continue;
        if (!gumbel_u && seed && (j & 3) == 0) ph = philox4(seed, (unsigned long long)(row * V + lane + 256 * (j >> 2)), iter, 0u);   // philox_gumbel_block(row, V, i)
        if (key[j] < thr) continue;
        float v = x[j] / temp_div;
        if (noisy) {
            const float u = gumbel_u ? gumbel_u[row * V + i] : philox_to_unit(ph.v[j & 3]);
            const float l1 = logf(fmaxf(u, 1e-20f));
            v += -logf(fmaxf(-l1, 1e-20f));
        }
        if (v > best || (v == best && i < bidx)) { best = v; bidx = i; }
    }
    wave_argmax(best, bidx);
    if (lane == 0 && was_mask) ids[row] = bidx;
    if (conf_mode) {   // probs_without_temperature = logits.softmax(-1) over the UNfiltered logits; score = 1 - p[pred] (muse_net:612-615)
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < VPL_MAX; ++j)
            if (valid[j]) mx = fmaxf(mx, x[j]);
        mx = wave_max(mx);
        float sum = 0.f, xp = 0.f;
#pragma unroll
        for (int j = 0; j < VPL_MAX; ++j) {
            if (!valid[j]) continue;
            sum += expf(x[j] - mx);
            if (lane + 64 * j == bidx) xp = x[j];
        }
        sum = wave_sum(sum);
        xp = wave_sum(xp);   // exactly one lane holds the picked logit
        if (lane == 0) conf_scores[row] = 1.f - expf(xp - mx) / sum;
    }
}

void launch_maskgit_pick(int64_t* ids, const float* logits, int ldl, const float* gumbel_u, int rows, int V, int k, float temperature, int64_t mask_id, hipStream_t s,
                         unsigned long long seed, unsigned iter, float* conf_scores, int conf_mode) {
    BG_REQUIRE(V <= 64 * VPL_MAX, "maskgit_pick: vocabulary %d > %d", V, 64 * VPL_MAX);
    BG_REQUIRE(conf_mode == 0 || conf_scores, "maskgit_pick: confidence scores requested without an output buffer");
    const float temp_div = fmaxf(temperature, 1e-10f);
    hipLaunchKernelGGL(maskgit_pick_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, ids, logits, ldl, gumbel_u, (long)rows, V, k, temp_div, mask_id, seed, iter, conf_scores,
                       conf_mode, status_current());
    LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------- self-critic scores
__global__ __launch_bounds__(256) void critic_scores_kernel(const float* __restrict__ embed, int lde, const float* __restrict__ w, const float* __restrict__ b,
                                                            const float* __restrict__ u, float noise_scale, float frac, float* __restrict__ scores, long rows, int D,
                                                            unsigned long long seed, unsigned iter, unsigned* __restrict__ status) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* e = embed + row * lde;
    float acc = 0.f;
    for (int i = lane * 4; i < D; i += 256) {
        const float4 ev = *reinterpret_cast<const float4*>(e + i);
        const float4 wv = *reinterpret_cast<const float4*>(w + i);
        acc = fmaf(ev.x, wv.x, acc); acc = fmaf(ev.y, wv.y, acc); acc = fmaf(ev.z, wv.z, acc); acc = fmaf(ev.w, wv.w, acc);
    }
    acc = wave_sum(acc);
    if (lane == 0) {
        float sc = acc + b[0];
        const float uu = u ? u[row] : (seed ? philox_uniform(seed, (unsigned long long)row, iter, 1u) : 0.5f);
        sc += ((uu - 0.5f) * noise_scale) * frac;
        if (nonfinite(sc)) status_raise(status, BG_ST_NONFINITE_LOGITS);   // (a NaN score would silently rank first in the re-masking)
        scores[row] = sc;
    }
}

void launch_critic_scores(const float* embed, int lde, const float* w, const float* b, const float* u, float noise_scale, float frac, float* scores, int rows, int D, hipStream_t s,
                          unsigned long long seed, unsigned iter) {
    BG_REQUIRE(D % 4 == 0, "critic_scores: D must be a multiple of 4");
    hipLaunchKernelGGL(critic_scores_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, embed, lde, w, b, u, noise_scale, frac, scores, (long)rows, D, seed, iter, status_current());
    LAUNCH_CHECK();
}

// ---------------------------------------------------------------------------------------------- Route A token pick
// lane owns the contiguous index range [lane*VPL, lane*VPL + VPL) so that the cumulative distribution runs in index order
// what follows the pick in a decode step (ArPickTail): every lane holds the row's token
__device__ __forceinline__ void ar_pick_tail(const ArPickTail& t, const int* d_step, long row, long token, int lane) {
    if (!t.out_all) return;
    const long j = t.fwd_idx[*d_step];
    if (lane == 0) t.out_all[row * t.N + j] = token;
    if (!t.x) return;
    const long id = token < 0 ? 0 : (token >= t.vocab_rows ? t.vocab_rows - 1 : token);
    // (one wave writes the D-wide row: 16-byte loads, four vectors per lane in flight - a scalar loop is 16 dependent L2 round trips and costs more than the launch saved)
    const float4* te = reinterpret_cast<const float4*>(t.tok_emb + id * t.D);
    const float4* ie = reinterpret_cast<const float4*>(t.img_embed ? t.img_embed + (row * t.C * t.T + j) * t.D : t.tok_emb);
    const float4* pe = reinterpret_cast<const float4*>(t.pos_emb + j * t.D);
    float4* xo = reinterpret_cast<float4*>(t.x + row * t.D);
    const int nv = t.D >> 2;   // launcher: D % 4 == 0
    for (int o0 = 0; o0 < nv; o0 += 256) {
        float4 a[4], b[4], c[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int o = min(o0 + lane + 64 * k, nv - 1);
            a[k] = te[o]; b[k] = ie[o]; c[k] = pe[o];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int o = o0 + lane + 64 * k;
            if (o < nv) {
                const float f = t.img_embed ? 1.f : 0.f;
                xo[o] = make_float4((a[k].x + f * b[k].x) + c[k].x, (a[k].y + f * b[k].y) + c[k].y, (a[k].z + f * b[k].z) + c[k].z, (a[k].w + f * b[k].w) + c[k].w);
            }
        }
    }
}

__global__ __launch_bounds__(64) void ar_pick_kernel(const float* __restrict__ logits, int ldl, const float* __restrict__ u_base, const int* __restrict__ d_step,
                                                     const int64_t* __restrict__ forced, int64_t* __restrict__ out, int V, int top_k, float temperature, ArPickTail tail,
                                                     unsigned* __restrict__ status) {
    const int lane = threadIdx.x;
    const long row = blockIdx.x;
    // partial decoding (ar_lm:161-165, 181-182): positions of the fixed cameras keep their given token, laid out [steps, rows] like the noise
    if (forced) {
        const int64_t f = forced[(d_step ? (long)(*d_step) * gridDim.x : 0) + row];
        if (f >= 0) {
            if (lane == 0) out[row] = f;
            ar_pick_tail(tail, d_step, row, f, lane);
            return;
        }
    }
    // explicit uniforms are laid out [steps, rows]; the step comes from the device counter so that one captured launch serves every step
    const float* u = u_base ? u_base + (d_step ? (long)(*d_step) * gridDim.x : 0) : nullptr;
    const float* lr = logits + row * ldl;
    const int per = (V + 63) / 64;
    float x[VPL_MAX];
    uint32_t key[VPL_MAX];
    bool valid[VPL_MAX];
#pragma unroll
    for (int j = 0; j < VPL_MAX; ++j) {
        const int i = lane * per + j;
        valid[j] = j < per && i < V;
        const float raw = valid[j] ? lr[i] : 0.f;
        if (nonfinite(raw)) status_raise(status, BG_ST_NONFINITE_LOGITS);   // ar_lm:202 / gpt:388: assert (~logits.isfinite()).sum() == 0 (forced positions: their head row is
        x[j] = valid[j] ? raw / temperature : -INFINITY;                    // not read here; a non-finite activation still reaches the next drawn position through the KV cache)
        key[j] = ordered_key(x[j]);
    }
    if (top_k > 0 && top_k < V) {
        const uint32_t thr = kth_largest_key(key, valid, top_k);
#pragma unroll
        for (int j = 0; j < VPL_MAX; ++j)
            if (valid[j] && key[j] < thr) x[j] = -INFINITY;  // values below the k-th largest -> -inf, ties kept (ar_lm:141)
    }
    float mx = -INFINITY;
    int midx = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < VPL_MAX; ++j)
        if (valid[j] && x[j] > mx) { mx = x[j]; midx = lane * per + j; }
    wave_argmax(mx, midx);
    if (!u) {
        if (lane == 0) out[row] = midx;
        ar_pick_tail(tail, d_step, row, midx, lane);
        return;
    }
    // inverse-CDF draw: first index whose cumulative probability exceeds u * total
    float e[VPL_MAX], local = 0.f;
#pragma unroll
    for (int j = 0; j < VPL_MAX; ++j) {
        e[j] = valid[j] ? expf(x[j] - mx) : 0.f;
        local += e[j];
    }
    float incl = local;  // inclusive scan across lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    const float total = __shfl(incl, 63, 64);
    const float target = u[row] * total;
    // first index (in vocabulary order) that carries probability AND whose cumulative sum exceeds the target.  The running sum is rebuilt per lane from
    // a cross-lane scan, so it need not be monotone across lane boundaries to the last bit: selecting by "first crossing with e > 0" (per-lane candidate,
    // wave minimum) can never land on a top-k-filtered token, unlike counting the entries below the target.
    float run = incl - local;
    int cand = 0x7fffffff, last = -1;
#pragma unroll
    for (int j = 0; j < VPL_MAX; ++j) {
        if (!valid[j]) continue;
        run += e[j];
        if (e[j] > 0.f) {
            last = lane * per + j;
            if (run > target && cand == 0x7fffffff) cand = lane * per + j;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        cand = min(cand, __shfl_xor(cand, o, 64));
        last = max(last, __shfl_xor(last, o, 64));
    }
    const int picked = cand != 0x7fffffff ? cand : last;   // u * total rounding up to the total: the last token with probability
    if (lane == 0) out[row] = picked;
    ar_pick_tail(tail, d_step, row, picked, lane);
}

void launch_ar_pick(const float* logits, int ldl, const float* u, const int* d_step, const int64_t* forced, int64_t* out, int rows, int V, int top_k, float temperature,
                    hipStream_t s, const ArPickTail* tail) {
    BG_REQUIRE(V <= 64 * VPL_MAX, "ar_pick: vocabulary %d > %d", V, 64 * VPL_MAX);
    BG_REQUIRE(!tail || !tail->out_all || d_step, "ar_pick: the fused tail reads the step from the device counter");
    BG_REQUIRE(!tail || !tail->x || tail->D % 4 == 0, "ar_pick: the fused embedding needs D %% 4 == 0");
    hipLaunchKernelGGL(ar_pick_kernel, dim3(rows), dim3(64), 0, s, logits, ldl, u, d_step, forced, out, V, top_k, temperature, tail ? *tail : ArPickTail{}, status_current());
    LAUNCH_CHECK();
}

// the uniforms the samplers draw in registers, written out (tests: explicit-noise run == seeded run)
__global__ void philox_fill_kernel(float* __restrict__ out, long n, unsigned long long seed, unsigned iter, unsigned stream, int V) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        if (stream == 0 && V > 0) {   // gumbel stream: the per-row block layout of maskgit_pick_kernel
            unsigned long long block; int word;
            philox_gumbel_block(i / V, V, (int)(i % V), block, word);
            out[i] = philox_to_unit(philox4(seed, block, iter, 0u).v[word]);
        } else {
            out[i] = philox_uniform(seed, (unsigned long long)i, iter, stream);
        }
    }
}
void launch_philox_fill(float* out, long n, unsigned long long seed, unsigned iter, unsigned stream, int V, hipStream_t s) {
    hipLaunchKernelGGL(philox_fill_kernel, dim3((int)std::min<long>((n + 255) / 256, 8192)), dim3(256), 0, s, out, n, seed, iter, stream, V);
    LAUNCH_CHECK();
}

}  // namespace bevgen
