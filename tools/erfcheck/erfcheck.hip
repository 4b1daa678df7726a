// erf_ocml (common.h) against the device library's erff for ALL 2^32 float bit patterns (NaN results compared as NaN == NaN).
// build: hipcc -O3 --offload-arch=gfx950 -I bevgen_amd/csrc tools/erfcheck/erfcheck.hip -o tools/erfcheck/erfcheck ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include "common.h"
__global__ void check(unsigned long long base, unsigned long long* mismatches, unsigned* first_bad) {
    const unsigned long long i = base + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned bits = (unsigned)i;
    const float x = __uint_as_float(bits);
    const float a = erff(x), b = bevgen::erf_ocml(x);
    const bool same = (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b);
    const float ga = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)), gb = bevgen::gelu_erf(x);
    const bool gsame = (__float_as_uint(ga) == __float_as_uint(gb)) || (ga != ga && gb != gb);
    if (!same || !gsame) { atomicAdd(mismatches, 1ull); atomicMin(first_bad, bits); }
}
__global__ void show(float x, float* out) { out[0] = erff(x); out[1] = bevgen::erf_ocml(x); out[2] = 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); out[3] = bevgen::gelu_erf(x); }
int main() {
    { float* d; hipMalloc(&d, 16); float h[4];
      for (float x : {1.0f, 1.5f, 2.0f, 3.7f, -1.25f, 0.5f, 9.0f}) { hipLaunchKernelGGL(show, dim3(1), dim3(1), 0, 0, x, d); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("x=%g erff %a mine %a | gelu %a mine %a\n", x, h[0], h[1], h[2], h[3]); } }
    unsigned long long* d_mis; unsigned* d_first;
    hipMalloc(&d_mis, 8); hipMalloc(&d_first, 4);
    hipMemset(d_mis, 0, 8); hipMemset(d_first, 0xFF, 4);
    for (unsigned long long base = 0; base < (1ull << 32); base += (1ull << 28))
        hipLaunchKernelGGL(check, dim3((1u << 28) / 256), dim3(256), 0, 0, base, d_mis, d_first);
    unsigned long long mis; unsigned first;
    hipMemcpy(&mis, d_mis, 8, hipMemcpyDeviceToHost); hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
    printf("erf_ocml / gelu_erf vs erff over 2^32 inputs: %llu mismatches (first bits 0x%08x)\n", mis, first);
    return mis != 0;
}
