// Shared declarations for libbevgen_hip (gfx950 / CDNA4 only).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <stdexcept>
#include <cstdio>
#include <cstdarg>

namespace bevgen {

// hipFuncSetAttribute (the > 64 KB dynamic-LDS opt-in) is PER DEVICE state and the launchers may be entered from several host threads with contexts on several GPUs:
// their one-time flags are indexed by the current device and atomic (setting an attribute twice is harmless)
constexpr int kMaxDevices = 64;
inline int device_slot() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    return dev >= 0 && dev < kMaxDevices ? dev : 0;
}

// -------- error handling: C++ exceptions inside, translated to int codes at the C-ABI --------
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

[[noreturn]] inline void fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw Error(code, buf);
}

#define HIP_CHECK(expr)                                                                            \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess) ::bevgen::fail(-2, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
    } while (0)

#define LAUNCH_CHECK() HIP_CHECK(hipGetLastError())

#define BG_REQUIRE(cond, ...)                      \
    do {                                           \
        if (!(cond)) ::bevgen::fail(-1, __VA_ARGS__); \
    } while (0)

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
inline long round_up(long a, long b) { return (a + b - 1) / b * b; }

// XCD-aware tile order for 2-D tiled GEMMs.  Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, observed, used
// for speed only), each with a private L2.  With the natural order the gx column tiles that share one A row-panel land on 8 different L2s
// and the panel is fetched from HBM 8 times (measured: 4.5x the algorithmic traffic).  Re-labelling so that every XCD walks a CONTIGUOUS
// range of tile ids keeps a panel's column tiles on one XCD, back to back.  Bijective for any grid size (MI355X guide, T1).
__device__ __forceinline__ void xcd_tile_lin(int gx, int gy, int lin, int& tx, int& ty) {   // lin = dispatch-order index of the workgroup (or of a persistent workgroup's turn)
    const int total = gx * gy;
    const int xcd = lin & 7, k = lin >> 3;
    const int q = total >> 3, r = total & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const int logical = base + k;
    ty = logical / gx;
    tx = logical - ty * gx;
}

__device__ __forceinline__ void xcd_tile(int gx, int gy, int& tx, int& ty) { xcd_tile_lin(gx, gy, blockIdx.y * gx + blockIdx.x, tx, ty); }

// Same, but the contiguous id range of an XCD walks bands of R row-tiles column by column, so that the ~32 workgroups resident on one XCD form
// an (R x 32/R) patch of the output: the patch's R A-panels and 32/R B-panels are each fetched once per k-slice and shared through that L2
// (R = 4 with 256x128 tiles = 1024 + 1024 operand rows per patch, the minimum for 32 tiles), and the A band stays warm while B streams past.
__device__ __forceinline__ void xcd_tile_banded_lin(int gx, int gy, int R, int lin, int& tx, int& ty) {
    int lx, ly;
    xcd_tile_lin(gx, gy, lin, lx, ly);
    const int logical = ly * gx + lx;
    const int band = logical / (R * gx);
    const int within = logical - band * (R * gx);
    const int h = min(R, gy - band * R);
    tx = within / h;
    ty = band * R + (within - tx * h);
}
__device__ __forceinline__ void xcd_tile_banded(int gx, int gy, int R, int& tx, int& ty) { xcd_tile_banded_lin(gx, gy, R, blockIdx.y * gx + blockIdx.x, tx, ty); }

// -------- device status word: what the reference asserts on the host every step (gpt:383,388 finite inputs / logits, ar_lm:202 finite logits) is flagged on the DEVICE
// here - no host synchronisation on the sampling path - into one host-visible word per context (mapped host memory, Ctx::status_host), read by the host wherever
// it synchronises anyway (bevgen_synchronize, the entry of every C-ABI call, bevgen_destroy).  A raised bit costs one system-scope atomic; the normal case costs the compare.
enum : unsigned {
    BG_ST_MLP_BARRIER = 1u,      // ar_mlp_fused_kernel: an XCD-local barrier timed out (the launch did not have the GPU to itself)
    BG_ST_MLP_PLACEMENT = 2u,    // ar_mlp_fused_kernel: a workgroup was not placed on the XCD its index implies
    BG_ST_NONFINITE_LOGITS = 4u, // a sampler saw a NaN / inf logit or critic score (the reference: assert (~logits.isfinite()).sum() == 0)
    BG_ST_F16_RANGE = 8u,        // a value written as an f16 operand (hi/lo planes of precision = f16x3, fp16 KV cache, fp16 decode activations) was NaN or |v| >= 65520:
                                 // the f16 image is inf / NaN where the reference's bf16 / fp32 arithmetic has an 8-bit exponent
    BG_ST_NONFINITE_PIXELS = 16u // the VQGAN decoder produced a NaN / inf pixel
};
// device address of the status word of the context whose C-ABI call is executing on this host thread (null outside a call): the launchers of the flagged kernels read it
unsigned* status_current();
unsigned* status_set_current(unsigned* dev);   // returns the previous one
__device__ __forceinline__ void status_raise(unsigned* st, unsigned bits) {
    if (st) __hip_atomic_fetch_or(st, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// both halves of a packed f16 pair: non-zero where the exponent field is all ones (inf / NaN) - three integer operations per pair
__device__ __forceinline__ unsigned f16x2_nonfinite(unsigned packed) { return ((packed & 0x7C007C00u) + 0x04000400u) & 0x80008000u; }
__device__ __forceinline__ bool nonfinite(float v) { return (__float_as_uint(v) & 0x7F800000u) == 0x7F800000u; }

// -------- split precision: v = hi + lo * 2^-11 with hi, lo in f16 (gemm_split.hip); interleaved plane layout [row][k/32][hi 32 | lo 32]
__device__ __forceinline__ _Float16 split_hi(float v) { return (_Float16)v; }
__device__ __forceinline__ _Float16 split_lo(float v, _Float16 hi) { return (_Float16)((v - (float)hi) * 2048.f); }
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half8_guard_t __attribute__((ext_vector_type(8)));
// range guard of the plane writers: `bad` collects (per thread) whether any hi half written so far is inf / NaN; the kernel raises BG_ST_F16_RANGE once at its end
__device__ __forceinline__ void guard_half4(half4_t h, unsigned& bad) {
    const uint2 w = __builtin_bit_cast(uint2, h);
    bad |= f16x2_nonfinite(w.x) | f16x2_nonfinite(w.y);
}
__device__ __forceinline__ void guard_half8(half8_guard_t h, unsigned& bad) {
    const uint4 w = __builtin_bit_cast(uint4, h);
    bad |= f16x2_nonfinite(w.x) | f16x2_nonfinite(w.y) | f16x2_nonfinite(w.z) | f16x2_nonfinite(w.w);
}
__device__ __forceinline__ void guard_half(_Float16 h, unsigned& bad) { bad |= ((unsigned)__builtin_bit_cast(unsigned short, h) & 0x7C00u) == 0x7C00u ? 1u : 0u; }
// store 4 (2) consecutive elements starting at column `col` (a multiple of 4 (2)) of one plane row
__device__ __forceinline__ void store_planes4(_Float16* row, int col, float4 v, unsigned& bad) {
    half4_t h, l;
    h[0] = split_hi(v.x); h[1] = split_hi(v.y); h[2] = split_hi(v.z); h[3] = split_hi(v.w);
    l[0] = split_lo(v.x, h[0]); l[1] = split_lo(v.y, h[1]); l[2] = split_lo(v.z, h[2]); l[3] = split_lo(v.w, h[3]);
    guard_half4(h, bad);
    _Float16* p = row + (col >> 5) * 64 + (col & 31);
    *reinterpret_cast<half4_t*>(p) = h;
    *reinterpret_cast<half4_t*>(p + 32) = l;
}
__device__ __forceinline__ void store_planes2(_Float16* row, int col, float2 v, unsigned& bad) {
    half2_t h, l;
    h[0] = split_hi(v.x); h[1] = split_hi(v.y);
    l[0] = split_lo(v.x, h[0]); l[1] = split_lo(v.y, h[1]);
    bad |= f16x2_nonfinite(__builtin_bit_cast(unsigned, h));
    _Float16* p = row + (col >> 5) * 64 + (col & 31);
    *reinterpret_cast<half2_t*>(p) = h;
    *reinterpret_cast<half2_t*>(p + 32) = l;
}

constexpr float kNegBig = -1.0e30f;  // "minus infinity" that never produces inf-inf NaNs in online softmax

// -------- wave-level helpers (wave = 64 lanes) --------
// value held by lane ^ 32: v_permlane32_swap (VALU, gfx950) instead of a ds_bpermute round trip through the LDS crossbar
__device__ __forceinline__ float xor32(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
// value held by lane ^ 16, same idea: v_permlane16_swap exchanges the odd 16-lane rows of one register with the even rows of the other
__device__ __forceinline__ float xor16(float x) {
    const unsigned u = __float_as_uint(x);
    const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 16) ? r[0] : r[1]);
}
// a + (a of lane ^ 32) in lanes 0..31 and c + (c of lane ^ 32) in lanes 32..63: ONE swap folds two values over the wave halves
__device__ __forceinline__ float fold32(float a, float c) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(c), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// the same over neighbouring 16-lane rows: a's sum in the even rows, b's in the odd rows
__device__ __forceinline__ float fold16(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// exp(x) for the online softmax: 2^n * 2^f with n = rint(x log2e) and f = x log2e - n evaluated with a two-term log2e (fma), so the argument
// of v_exp_f32 is exact to ~2^-25 and |f| <= 0.5: ~1.5 ulp overall in 7 instructions (the library expf is ~12 with its range checks).
// x <= 0 in every use; x = -1e30 (masked score) gives n -> INT_MIN and the result 0.
__device__ __forceinline__ float exp_softmax(float x) {
    const float t = x * 1.44269502162933349609375f;
    const float n = __builtin_rintf(t);
    float f = __builtin_fmaf(x, 1.44269502162933349609375f, -n);
    f = __builtin_fmaf(x, 1.92596299112661746e-8f, f);
    return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(f), (int)fmaxf(n, -300.f));
}

// erf-based GELU (torch nn.GELU() / F.gelu default)
// erff of the device library (OCML), branch-free.  The library form is two branches (|y| < 1: an odd polynomial; else 1 - exp(-p(|y|))) - under divergence a wave runs both
// plus the exec-mask bookkeeping, 32 times per lane in the GEGLU epilogue of the LDS-DMA GEMM.  Same coefficients and the same operation order here, both sides computed
// unconditionally and selected; of the two range checks of the library's expf the overflow one is dropped (p is never negative).  Bit-identical to
// erff for every one of the 2^32 float inputs (tools/erfcheck, run on the MI355X).
__device__ __forceinline__ float erf_ocml(float y) {
#pragma clang fp contract(off)   // (the library's operation sequence exactly: the explicit fmaf calls stay fused, nothing else is)
    const float t = fabsf(y);
    // |y| >= 1
    float p = fmaf(t, __uint_as_float(0x378e98abu), __uint_as_float(0xb9c68948u));
    p = fmaf(t, p, __uint_as_float(0x3b7cd369u));
    p = fmaf(t, p, __uint_as_float(0xbcc618b2u));
    p = fmaf(t, p, __uint_as_float(0x3dda74e4u));
    p = fmaf(t, p, __uint_as_float(0x3f228afdu));
    p = fmaf(t, p, __uint_as_float(0x3e03c728u));
    p = fmaf(t, p, t);
    const float nl2e = __uint_as_float(0xbfb8aa3bu);              // -log2(e), high part
    const float ph = p * nl2e;
    float pl = fmaf(p, nl2e, -ph);
    pl = fmaf(p, __uint_as_float(0xb2a5705fu), pl);               // + p * (-log2(e), low part)
    const float pn = __builtin_rintf(ph);
    float e = ldexpf(__builtin_amdgcn_exp2f((ph - pn) + pl), (int)pn);
    e = p > __uint_as_float(0x42ce8ed0u) ? 0.0f : e;               // (the library's underflow check; also the polynomial's overflow for |y| > 4e6: 1 - 0)
    const float big = 1.0f - e;
    // |y| < 1
    const float s2 = y * y;
    float q = fmaf(s2, __uint_as_float(0xba1345e1u), __uint_as_float(0x3ba10414u));
    q = fmaf(s2, q, __uint_as_float(0xbcdac9b8u));
    q = fmaf(s2, q, __uint_as_float(0x3de703beu));
    q = fmaf(s2, q, __uint_as_float(0xbec09330u));
    q = fmaf(s2, q, __uint_as_float(0x3e0375d0u));
    const float small = fmaf(t, q, t);
    return copysignf(t < 1.0f ? small : big, y);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_ocml(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float swish(float x) { return x / (1.0f + __expf(-x)); }

// Philox4x32-10 counter-based generator (Salmon et al. 2011): uniform in [0,1) with 24 random bits for element `idx` of noise stream (iter, stream) under `seed`.
// Stateless: the MaskGit samplers draw their gumbel / critic uniforms in registers instead of reading [timesteps, rows, T, V] tensors (1.8 GB at the bench size).
__host__ __device__ inline unsigned philox_mulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
struct Philox4 { unsigned v[4]; };
__host__ __device__ inline Philox4 philox4(unsigned long long seed, unsigned long long idx, unsigned iter, unsigned stream) {
    unsigned c0 = (unsigned)idx, c1 = (unsigned)(idx >> 32), c2 = iter, c3 = stream;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned h0 = philox_mulhi(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const unsigned h1 = philox_mulhi(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        const unsigned n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{{c0, c1, c2, c3}};
}
__host__ __device__ inline float philox_to_unit(unsigned x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }
// stream 1 (critic noise, one value per row): element idx -> word 0 of block idx
__host__ __device__ inline float philox_uniform(unsigned long long seed, unsigned long long idx, unsigned iter, unsigned stream) {
    return philox_to_unit(philox4(seed, idx, iter, stream).v[0]);
}
// stream 0 (gumbel noise, V values per row, consumed by a wave whose lane l holds elements l + 64 j): element (row, i) -> word (j & 3) of block
// row * V + l + 256 (j >> 2), l = i % 64, j = i / 64: one Philox block serves four of a lane's values
__host__ __device__ inline void philox_gumbel_block(long row, int V, int i, unsigned long long& block, int& word) {
    const int l = i & 63, j = i >> 6;
    block = (unsigned long long)(row * V + l + 256 * (j >> 2));
    word = j & 3;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// LDS-DMA request the compiler does not see (inline asm): 16 bytes per lane from `src` (per-lane address) to LDS byte address `lds_base` + lane * 16 (wave-uniform base).
// With the builtin, LLVM's wait-count pass counts the request like a load and - not knowing which LDS bytes it writes - puts `s_waitcnt vmcnt(0)` in front of later LDS
// writes (in the fused decode kernel the xn / bias row stores of ln1 then waited for every K/V piece: statistics done at 10.0 instead of 5.8 us).  Hidden requests only ever make the compiler's counted
// waits stricter (loads return in issue order; an uncounted later request means it waits for a few more of the older ones), never weaker; the one reader of the staged bytes
// (Attend::run_staged) orders itself by an explicit vmcnt wait.
__device__ __forceinline__ void glds16_hidden(const void* src, unsigned lds_base) {
    unsigned keep;   // (m0 is the compiler's: saved and restored around the request)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_base) : "memory");
}
// the same with the non-temporal hint (streamed once: should not displace what other kernels parked in L2)
__device__ __forceinline__ void glds16_hidden_nt(const void* src, unsigned lds_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_base) : "memory");
}
__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}


}  // namespace bevgen
