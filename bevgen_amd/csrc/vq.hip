// Small VQGAN-decode helpers: codebook lookup (VectorQuantizer2.get_codebook_entry, stage1/quantize.py:314-329) and the final
// layout change NHWC -> NCHW fused with util.denormalize_tensor (bev_utils/util.py:97-118: x*std+mean per channel, clamp [0,1]).
#include "common.h"
#include "kernels.h"

namespace bevgen {

__global__ __launch_bounds__(256) void codebook_gather_kernel(const int64_t* __restrict__ ids, const float* __restrict__ codebook, float* __restrict__ out,
                                                              long rows, int dim4, int n_embed) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * dim4; i += (long)gridDim.x * blockDim.x) {
        const long r = i / dim4;
        const int c = (int)(i - r * dim4);
        long id = ids[r];
        id = id < 0 ? 0 : (id >= n_embed ? n_embed - 1 : id);
        reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(codebook)[id * dim4 + c];
    }
}

void launch_codebook_gather(const int64_t* ids, const float* codebook, float* out, int rows, int dim, int n_embed, hipStream_t s) {
    BG_REQUIRE(dim % 4 == 0, "codebook_gather: embed dim must be a multiple of 4");
    const long total = (long)rows * (dim / 4);
    hipLaunchKernelGGL(codebook_gather_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, s, ids, codebook, out, (long)rows, dim / 4, n_embed);
    LAUNCH_CHECK();
}

// x [n, hw, ldc] (first C channels used) -> y [n, C, hw]
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, long total, int hw, int C, int ldc,
                                                           const float* __restrict__ mean, const float* __restrict__ stdv, int clamp01, uint8_t* __restrict__ y8,
                                                           unsigned* __restrict__ status) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % hw);
        const long nc = i / hw;
        const int c = (int)(nc % C);
        const long n = nc / C;
        float v = x[(n * hw + p) * ldc + c];
        if (nonfinite(v)) status_raise(status, BG_ST_NONFINITE_PIXELS);   // (the clamp below would turn a NaN into a valid-looking pixel)
        if (mean) v = v * stdv[c] + mean[c];
        if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        if (y8) y8[i] = (uint8_t)fminf(fmaxf(rintf(v * 255.0f), 0.f), 255.f);   // round(x*255) (half to even, like torch.round), the storage format of the images
        else y[i] = v;
    }
}

void launch_nhwc_to_nchw(const float* x, float* y, int n, int hw, int C, int ldc, const float* mean, const float* stdv, int clamp01, hipStream_t s, uint8_t* y8) {
    const long total = (long)n * C * hw;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((int)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, s, x, y, total, hw, C, ldc, mean, stdv, clamp01, y8, status_current());
    LAUNCH_CHECK();
}

// ------------------------------------------------------------------------------------------------ decoder tail in one kernel
// Decoder.forward's last three operators (stage1/model.py:532-536: norm_out -> nonlinearity -> conv_out) + denormalize_tensor (bev_utils/util.py:97-118) + the NHWC -> NCHW
// (/ uint8) write: GroupNorm-apply + swish on the way INTO LDS, a direct 3x3 convolution 128 -> 3 on the vector ALUs, per-channel x std + mean, clamp, store.
// As an implicit GEMM the 3-channel convolution filled 3 of the 128 columns of an MFMA tile (1.8-3.7 ms per 48-image pass, + 1.2 ms for the GroupNorm-apply pass that wrote
// the 1.6 GB plane image it read); each output pixel is 1152 MACs x 3 channels over data that is read once: bandwidth work.
// One workgroup = 16 x 16 output pixels of one image; the 18 x 18 halo tile goes through LDS in chunks of 32 channels (pixel stride 36 floats: ds_read_b128 of 16 consecutive
// pixels is bank-conflict free), zero outside the image (the convolution pads the ACTIVATED tensor); thread = pixel, 3 accumulators; weights are wave-uniform (scalar loads).
constexpr int OC_T = 16, OC_HALO = OC_T + 2, OC_CC = 32, OC_PS = 36;
template <int COUT>
__global__ __launch_bounds__(256) void vq_out_conv_kernel(const float* __restrict__ x, const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ wgt, const float* __restrict__ bias, const float* __restrict__ mean, const float* __restrict__ stdv,
                                                          int clamp01, float* __restrict__ y, uint8_t* __restrict__ y8, int H, int W, int C, unsigned* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) float tile[OC_HALO * OC_HALO * OC_PS];
    const int tid = threadIdx.x, px = tid & 15, py = tid >> 4;
    const int x0 = blockIdx.x * OC_T, y0 = blockIdx.y * OC_T, n = blockIdx.z;
    const int cpg = C / 32;   // GroupNorm(32 groups)
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    for (int c0 = 0; c0 < C; c0 += OC_CC) {
        // ---- halo tile of this channel chunk, normalised + activated: one float4 (4 channels of one pixel) per thread and pass
        for (int i = tid; i < OC_HALO * OC_HALO * (OC_CC / 4); i += 256) {
            const int q = i & (OC_CC / 4 - 1), p = i / (OC_CC / 4);
            const int hy = p / OC_HALO, hx = p - hy * OC_HALO;
            const int gy = y0 + hy - 1, gx = x0 + hx - 1;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const int c = c0 + 4 * q;
                const float4 v = *reinterpret_cast<const float4*>(x + (((long)n * H + gy) * W + gx) * C + c);
                const float in[4] = {v.x, v.y, v.z, v.w};
                float r[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float* st = stats + ((long)n * 32 + (c + k) / cpg) * 2;
                    const float t = (in[k] - st[0]) * st[1] * gamma[c + k] + beta[c + k];
                    r[k] = t / (1.f + expf(-t));
                }
                o = make_float4(r[0], r[1], r[2], r[3]);
            }
            *reinterpret_cast<float4*>(&tile[p * OC_PS + 4 * q]) = o;
        }
        __syncthreads();
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float* tp = &tile[((py + dy) * OC_HALO + px + dx) * OC_PS];
                const float* wp = wgt + (long)(dy * 3 + dx) * C + c0;   // [Cout][kh][kw][Cin]
#pragma unroll
                for (int c = 0; c < OC_CC; c += 4) {
                    const float4 v = *reinterpret_cast<const float4*>(tp + c);
#pragma unroll
                    for (int co = 0; co < COUT; ++co) {
                        const float* w = wp + (long)co * 9 * C + c;
                        acc[co] = fmaf(v.x, w[0], acc[co]);
                        acc[co] = fmaf(v.y, w[1], acc[co]);
                        acc[co] = fmaf(v.z, w[2], acc[co]);
                        acc[co] = fmaf(v.w, w[3], acc[co]);
                    }
                }
            }
        __syncthreads();
    }
    const int oy = y0 + py, ox = x0 + px;
    if (oy < H && ox < W) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            float v = acc[co] + (bias ? bias[co] : 0.f);
            if (nonfinite(v)) status_raise(status, BG_ST_NONFINITE_PIXELS);   // (the clamp below would turn a NaN into a valid-looking pixel)
            if (mean) v = v * stdv[co] + mean[co];
            if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
            const long o = (((long)n * COUT + co) * H + oy) * W + ox;
            if (y8) y8[o] = (uint8_t)fminf(fmaxf(rintf(v * 255.0f), 0.f), 255.f);   // round(x*255), half to even like torch.round
            else y[o] = v;
        }
    }
}

bool vq_out_conv_supported(int C, int cout) { return cout == 3 && C % 32 == 0 && C >= 32; }

void launch_vq_out_conv(const float* x, const float* stats, const float* gamma, const float* beta, const float* wgt, const float* bias, const float* mean, const float* stdv, int clamp01,
                        float* y, uint8_t* y8, int n, int H, int W, int C, int cout, hipStream_t s) {
    BG_REQUIRE(vq_out_conv_supported(C, cout), "vq_out_conv: C=%d cout=%d", C, cout);
    hipLaunchKernelGGL((vq_out_conv_kernel<3>), dim3(cdiv(W, OC_T), cdiv(H, OC_T), n), dim3(256), 0, s, x, stats, gamma, beta, wgt, bias, mean, stdv, clamp01, y, y8, H, W, C, status_current());
    LAUNCH_CHECK();
}

// x [n, C, hw] -> y [n, hw, C]
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, long total, int hw, int C) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const long np = i / C;
        const int p = (int)(np % hw);
        const long n = np / hw;
        y[i] = x[(n * C + c) * hw + p];
    }
}

void launch_nchw_to_nhwc(const float* x, float* y, int n, int hw, int C, hipStream_t s) {
    const long total = (long)n * C * hw;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((int)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, s, x, y, total, hw, C);
    LAUNCH_CHECK();
}

}  // namespace bevgen
