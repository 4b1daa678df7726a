// Which XCD does workgroup i of a launch run on?  Every workgroup reads HW_REG_XCC_ID; the probe prints the map for the grid shapes of the decode kernels
// (the placement is used for speed only - tile orders, which workgroups share an L2 - never for correctness).  build: hipcc --offload-arch=gfx950 -O2 -o xcc_probe xcc_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(int* out) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) {
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        out[blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)] = (int)(x & 0xf);
    }
    if (threadIdx.x == 1) lds[0] = 1;
}
static void run(const char* what, dim3 grid, int threads, int lds) {
    const int n = grid.x * grid.y * grid.z;
    int* d; hipMalloc(&d, n * sizeof(int));
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024);
    hipLaunchKernelGGL(probe, grid, dim3(threads), lds, 0, d);
    std::vector<int> h(n); hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
    int rr = 0; for (int i = 0; i < n; ++i) rr += h[i] == (h[0] + i) % 8;
    printf("%s: grid (%u,%u,%u) x %d threads, %d B LDS: %d of %d workgroups on XCD (xcd(0) + i) %% 8; first 24:", what, grid.x, grid.y, grid.z, threads, lds, rr, n);
    for (int i = 0; i < 24 && i < n; ++i) printf(" %d", h[i]);
    printf("\n");
    hipFree(d);
}
int main() {
    run("skinny / MLP launch", dim3(256, 1, 1), 512, 0);
    run("MLP-down (tiles x K slices)", dim3(64, 4, 1), 512, 0);
    run("fused decode attention (head, sequence)", dim3(16, 16, 1), 1024, 158 * 1024);
    run("GEMM-like", dim3(8, 96, 1), 512, 96 * 1024);
    run("attention_split-like", dim3(6, 16, 16), 512, 36 * 1024);
    run("5 workgroups (not a multiple of 8)", dim3(5, 1, 1), 256, 0);
    run("... and the launch after it", dim3(16, 1, 1), 256, 0);
    return 0;
}
