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
                                                           const float* __restrict__ mean, const float* __restrict__ stdv, int clamp01, uint8_t* __restrict__ y8) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int p = (int)(i % hw);
        const long nc = i / hw;
        const int c = (int)(nc % C);
        const long n = nc / C;
        float v = x[(n * hw + p) * ldc + c];
        if (mean) v = v * stdv[c] + mean[c];
        if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
        if (y8) y8[i] = (uint8_t)fminf(fmaxf(rintf(v * 255.0f), 0.f), 255.f);   // round(x*255) (half to even, like torch.round), the storage format of the images
        else y[i] = v;
    }
}

void launch_nhwc_to_nchw(const float* x, float* y, int n, int hw, int C, int ldc, const float* mean, const float* stdv, int clamp01, hipStream_t s, uint8_t* y8) {
    const long total = (long)n * C * hw;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((int)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, s, x, y, total, hw, C, ldc, mean, stdv, clamp01, y8);
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
