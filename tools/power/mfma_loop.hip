// Register-only f16 MFMA loop (v_mfma_f32_32x32x16_f16 with random, non-trivial operands): what clock and power does the chip hold when every matrix pipe
// issues back to back and nothing else happens?  Reference point for the "power envelope" reading of the LDS-DMA GEMM.
//   ./mfma_loop [seconds] [independent accumulators per wave: 4] [waves per SIMD: 2] [fresh operands: 0 | 16]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// NOPS distinct A and NOPS distinct B operand register sets, walked so that EVERY instruction sees operands that differ from the previous instruction's on both ports
// (the original loop below cycles through 4 x 4 fixed pairs: the same operand values come back every fourth instruction).  What a GEMM's matrix pipe sees is fresh data
// on every instruction; if the switching activity of the multiplier array matters for power, this loop - not the 4 x 4 one - is the register-only reference point.
template <int NACC, int NOPS>
__global__ __launch_bounds__(512) void mfma_loop_fresh(const half8* __restrict__ src, float* sink, int iters) {
    half8 a[NOPS], b[NOPS];
    f32x16 acc[NACC];
    for (int i = 0; i < NOPS; ++i) {
        a[i] = src[(threadIdx.x + 512 * i) & 4095];
        b[i] = src[(threadIdx.x * 7 + 131 * i + 2048) & 4095];
    }
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[u % NOPS], b[(u * 5 + 3) % NOPS], acc[u % NACC], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 123.456f) sink[0] = s;
}

template <int NACC>
__global__ __launch_bounds__(512) void mfma_loop(const half8* __restrict__ src, float* sink, int iters) {
    half8 a[NACC], b[NACC];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) {
        a[i] = src[(threadIdx.x + 512 * i) & 4095];
        b[i] = src[(threadIdx.x * 7 + 131 * i) & 4095];
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[(i + u) % NACC], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 123.456f) sink[0] = s;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 3.0;
    const int wps = argc > 3 ? atoi(argv[3]) : 2;
    const int fresh = argc > 4 ? atoi(argv[4]) : 0;   // 16: the loop with 16 + 16 operand sets (fresh operands on every instruction)
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    half8* src; float* sink;
    CK(hipMalloc(&src, 4096 * sizeof(half8))); CK(hipMalloc(&sink, 64));
    half8* h = (half8*)malloc(4096 * sizeof(half8));
    srand(1);
    for (int i = 0; i < 4096; ++i) for (int j = 0; j < 8; ++j) h[i][j] = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);   // random mantissas: realistic toggling
    CK(hipMemcpy(src, h, 4096 * sizeof(half8), hipMemcpyHostToDevice));
    const int blocks = cus * wps / 2;   // 512 threads = 8 waves = 2 per SIMD per block
    const int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto t0 = std::chrono::steady_clock::now();
    int launches = 0; double ms_total = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0));
        if (fresh) hipLaunchKernelGGL((mfma_loop_fresh<4, 16>), dim3(blocks), dim3(512), 0, 0, src, sink, iters);
        else hipLaunchKernelGGL((mfma_loop<4>), dim3(blocks), dim3(512), 0, 0, src, sink, iters);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ms_total += ms; ++launches;
    }
    const double flop = (double)launches * blocks * 8.0 /*waves*/ * iters * 8 * 4 * (2.0 * 32 * 32 * 16);
    printf("mfma_loop%s: %d CUs, %d blocks x 8 waves, %d launches, %.1f ms busy, %.1f TFLOP/s f16 dense (peak 2500 at 2.4 GHz) => %.2f GHz-equivalent of back-to-back issue\n",
           fresh ? " (16 + 16 operand sets: fresh operands on every instruction)" : "", cus, blocks, launches, ms_total, flop / (ms_total * 1e-3) / 1e12,
           flop / (ms_total * 1e-3) / 1e12 / 2500.0 * 2.4);
    return 0;
}
