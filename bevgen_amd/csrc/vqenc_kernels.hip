// Kernels of the VQGAN *encode* side (the step before the sampling path: BEV segmentation -> condition token ids, images -> z ids):
//   VectorQuantizer2.forward   stage1/quantize.py:271-312   argmin_j ( |z|^2 + |e_j|^2 - 2 z.e_j )
//   input layout change NCHW -> NHWC with the channel axis zero-padded to a multiple of 32 (implicit-GEMM convolutions need Cin % 32 == 0)
#include "common.h"
#include "kernels.h"

namespace bevgen {

// x [n, C, hw] -> y [n, hw, Cpad] (channels >= C are zero)
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* __restrict__ x, float* __restrict__ y, long total, int hw, int C, int Cpad) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cpad);
        const long np = i / Cpad;
        const int p = (int)(np % hw);
        const long n = np / hw;
        y[i] = c < C ? x[(n * C + c) * hw + p] : 0.f;
    }
}
void launch_nchw_to_nhwc_pad(const float* x, float* y, int n, int hw, int C, int Cpad, hipStream_t s) {
    const long total = (long)n * hw * Cpad;
    hipLaunchKernelGGL(nchw_to_nhwc_pad_kernel, dim3((int)std::min<long>((total + 255) / 256, 8192)), dim3(256), 0, s, x, y, total, hw, C, Cpad);
    LAUNCH_CHECK();
}

// conv weight [Cout][Cin][kh][kw] -> [Cout][kh][kw][CinPad] (zero padded)
__global__ void relayout_conv_weight_pad_kernel(const float* __restrict__ w, float* __restrict__ o, int cout, int cin, int cin_pad, int kh, int kw) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)cout * cin_pad * kh * kw;
    if (i >= total) return;
    const int ci = (int)(i % cin_pad);
    long t = i / cin_pad;
    const int x = (int)(t % kw); t /= kw;
    const int y = (int)(t % kh);
    const int co = (int)(t / kh);
    o[i] = ci < cin ? w[(((long)co * cin + ci) * kh + y) * kw + x] : 0.f;
}
void launch_relayout_conv_weight_pad(const float* w, float* o, int cout, int cin, int cin_pad, int kh, int kw, hipStream_t s) {
    hipLaunchKernelGGL(relayout_conv_weight_pad_kernel, dim3(cdiv((long)cout * cin_pad * kh * kw, 256)), dim3(256), 0, s, w, o, cout, cin, cin_pad, kh, kw);
    LAUNCH_CHECK();
}

// out[row] = sum_k x[row,k]^2   (one wave per row)
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* __restrict__ x, float* __restrict__ out, long rows, int D) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) { const float v = x[row * D + i]; s = fmaf(v, v, s); }
    s = wave_sum(s);
    if (lane == 0) out[row] = s;
}
void launch_row_sqnorm(const float* x, float* out, long rows, int D, hipStream_t s) {
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, s, x, out, rows, D);
    LAUNCH_CHECK();
}

// ids[row] = argmin_j (zz[row] + ee[j]) - 2 * dots[row, j]; ties -> lowest index (torch.argmin)
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ dots, const float* __restrict__ zz, const float* __restrict__ ee, int64_t* __restrict__ ids,
                                                        long rows, int n_e) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float z2 = zz[row];
    float best = INFINITY;
    int bidx = 0x7fffffff;
    for (int j = lane; j < n_e; j += 64) {
        const float d = (z2 + ee[j]) - 2.f * dots[row * n_e + j];
        if (d < best) { best = d; bidx = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        if (ob < best || (ob == best && oi < bidx)) { best = ob; bidx = oi; }
    }
    if (lane == 0) ids[row] = bidx;
}
void launch_vq_argmin(const float* dots, const float* zz, const float* ee, int64_t* ids, long rows, int n_e, hipStream_t s) {
    hipLaunchKernelGGL(vq_argmin_kernel, dim3((int)((rows + 3) / 4)), dim3(256), 0, s, dots, zz, ee, ids, rows, n_e);
    LAUNCH_CHECK();
}

}  // namespace bevgen
