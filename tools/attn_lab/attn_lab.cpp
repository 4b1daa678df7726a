// Attention lab: times the Route M attention kernel of the product (bevgen_amd/csrc/attention_split.hip compiled with -DBEVGEN_ATTN_LAB) against the
// round-1 kernel (attention_split_r1.inc) on the self-attention shape of BASELINE configs[1] (B=16 scenes, 16 heads, Nq=1536, 1568 keys) and on the
// cross-attention shape (288 keys), checks both against an fp64 evaluation of twelve output rows, and prints the phase timestamps of one
// workgroup of the product kernel.   build: bash tools/attn_lab/build.sh (here, cross-compiles);   run on the GPU box: bash tools/attn_lab/run.sh
// The variant sweeps this tool was used for are recorded in profiles/r02_attention_lab.txt.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../bevgen_amd/csrc/common.h"
#include "../../bevgen_amd/csrc/kernels.h"
#include "../../bevgen_amd/csrc/profiler.h"

namespace bevgen {
extern int g_attn_variant;   // 0 = round-1 kernel, 1 = product kernel, 2 = product kernel with phase stamps
void attn_lab_read_trace(unsigned long long* out);
}
using namespace bevgen;

static unsigned long long rng_state = 0x9E3779B97F4A7C15ull;
static float urand() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return (float)((rng_state >> 40) * (1.0 / 16777216.0));
}

static void fill_planes(std::vector<_Float16>& hi, std::vector<_Float16>& lo, size_t n, float amp) {
    hi.resize(n); lo.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const float v = (urand() * 2.f - 1.f) * amp;
        hi[i] = (_Float16)v;
        lo[i] = (_Float16)((v - (float)hi[i]) * 2048.f);
    }
}

template <class T>
static T* to_dev(const std::vector<T>& v) {
    T* d;
    HIP_CHECK(hipMalloc(&d, v.size() * sizeof(T)));
    HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return d;
}

int main() {
    const int B = 16, H = 16, Nq = 1536;
    for (int shape = 0; shape < 2; ++shape) {
        const int Nk_pad = shape == 0 ? 1568 : 288;
        const size_t nq = (size_t)B * H * Nq * 64, nk = (size_t)B * H * Nk_pad * 64;
        std::vector<_Float16> qh, ql, kh, kl, vh, vl;
        fill_planes(qh, ql, nq, 3.0f);   // l2-normalised q, k times scales: |score| of a few units in the base-2 domain
        fill_planes(kh, kl, nk, 0.4f);
        fill_planes(vh, vl, nk, 1.0f);
        std::vector<float> bias((size_t)Nq * Nk_pad);
        for (auto& x : bias) x = (urand() - 0.5f) * 4.f;
        for (int q = 0; q < Nq; ++q)
            for (int k = Nk_pad - 13; k < Nk_pad; ++k) bias[(size_t)q * Nk_pad + k] = -1.0e30f;   // padded keys
        AttnSplitArgs a{};
        a.Qh = to_dev(qh); a.Ql = to_dev(ql); a.Kh = to_dev(kh); a.Kl = to_dev(kl); a.VTh = to_dev(vh); a.VTl = to_dev(vl);
        a.bias = to_dev(bias); a.ldbias = Nk_pad; a.bias_head_stride = 0;
        float* bpk;
        HIP_CHECK(hipMalloc(&bpk, attn_bias_packed_floats(Nq, Nk_pad) * 4));
        launch_pack_attn_bias(a.bias, Nk_pad, Nq, Nk_pad, bpk, 0);
        a.bias_pk = bpk;
        a.B = B; a.H = H; a.Nq = Nq; a.Nk_pad = Nk_pad; a.scale = 1.f;
        a.o_bstride = (long)Nq * H * 64; a.o_qstride = (long)H * 64; a.o_hstride = 64;
        const size_t no = (size_t)B * Nq * H * 64;
        float* O;
        HIP_CHECK(hipMalloc(&O, no * 4));
        a.O = O; a.Op = nullptr;
        std::vector<float> out(no);
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        const double flop = 4.0 * B * H * (double)Nq * Nk_pad * 64;
        double best[2] = {1e30, 1e30};
        for (int round = 0; round < 4; ++round)
            for (int v = 0; v < 2; ++v) {
                g_attn_variant = v;
                launch_attention_split(a, 0);
                HIP_CHECK(hipEventRecord(e0, 0));
                const int reps = 8;
                for (int i = 0; i < reps; ++i) launch_attention_split(a, 0);
                HIP_CHECK(hipEventRecord(e1, 0));
                HIP_CHECK(hipEventSynchronize(e1));
                float ms = 0;
                HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
                best[v] = std::fmin(best[v], ms * 1e3 / reps);
            }
        for (int v = 0; v < 2; ++v) {
            g_attn_variant = v;
            HIP_CHECK(hipMemset(O, 0, no * 4));
            launch_attention_split(a, 0);
            HIP_CHECK(hipMemcpy(out.data(), O, no * 4, hipMemcpyDeviceToHost));
            double err64 = 0;   // fp64 evaluation of twelve output rows from the same planes
            for (int ri = 0; ri < 12; ++ri) {
                const int b = (ri * 7) % B, hd = (ri * 5) % H, q = ri == 11 ? Nq - 1 : (ri * 131) % Nq;
                const size_t qo = (((size_t)b * H + hd) * Nq + q) * 64, ko = ((size_t)b * H + hd) * Nk_pad * 64;
                std::vector<double> sc(Nk_pad);
                double m = -1e300;
                for (int k = 0; k < Nk_pad; ++k) {
                    double acc = 0;
                    for (int d = 0; d < 64; ++d)
                        acc += ((double)qh[qo + d] + (double)ql[qo + d] / 2048.0) * ((double)kh[ko + (size_t)k * 64 + d] + (double)kl[ko + (size_t)k * 64 + d] / 2048.0);
                    sc[k] = acc + bias[(size_t)q * Nk_pad + k];
                    m = std::fmax(m, sc[k]);
                }
                double l = 0;
                std::vector<double> od(64, 0.0);
                for (int k = 0; k < Nk_pad; ++k) {
                    const double pk = std::exp2(sc[k] - m);
                    l += pk;
                    for (int d = 0; d < 64; ++d) od[d] += pk * ((double)vh[ko + (size_t)d * Nk_pad + k] + (double)vl[ko + (size_t)d * Nk_pad + k] / 2048.0);
                }
                for (int d = 0; d < 64; ++d) err64 = std::fmax(err64, std::fabs(od[d] / l - out[((size_t)b * Nq + q) * H * 64 + hd * 64 + d]));
            }
            printf("Nk=%4d %s: %8.1f us  %6.1f TF-equiv  max err vs fp64 (12 rows) %.3g\n", Nk_pad, v == 0 ? "round-1 kernel (4 waves)        " : "product kernel (8-wave pingpong)",
                   best[v], flop / best[v] * 1e-6, err64);
            fflush(stdout);
        }
        {   // phase trace of one workgroup (waves 0 and 4 = the two waves of SIMD 0)
            g_attn_variant = 2;
            launch_attention_split(a, 0);
            HIP_CHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> tr(8 * 16 * 8);
            attn_lab_read_trace(tr.data());
            const int nt = Nk_pad / 32;
            for (int w : {0, 4}) {
                printf("trace Nk=%d wave %d: per tile [M-phase | wait at barrier | LDS store | softmax + loads | wait at barrier] cycles\n", Nk_pad, w);
                for (int t = (nt > 14 ? 10 : 2); t < (nt > 14 ? 14 : (nt < 6 ? nt : 6)); ++t) {
                    const unsigned long long* e = &tr[(w * 16 + t) * 8];
                    printf("   t=%2d  start %8lld  M %5llu  bar %5llu  store %5llu  softmax %5llu (max+bias req %5llu | exp+split+K/V req %5llu | rescale %5llu)  bar %5llu\n", t,
                           (long long)(e[0] - tr[(0 * 16 + 2) * 8]), e[1] - e[0], e[2] - e[1], e[3] - e[2], e[4] - e[3], e[6] - e[3], e[7] - e[6], e[4] - e[7], e[5] - e[4]);
                }
            }
        }
        for (const void* ptr : {(const void*)a.Qh, (const void*)a.Ql, (const void*)a.Kh, (const void*)a.Kl, (const void*)a.VTh, (const void*)a.VTl, (const void*)a.bias, (const void*)bpk,
                                (const void*)O})
            HIP_CHECK(hipFree(const_cast<void*>(ptr)));
    }
    return 0;
}
