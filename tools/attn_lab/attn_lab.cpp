// Attention lab: times variants of attention_split_kernel (bevgen_amd/csrc/attention_split.hip compiled with -DBEVGEN_ATTN_LAB) on the Route M
// self-attention shape of BASELINE configs[1] (B=16 scenes, 16 heads, Nq=1536, 1568 keys) and on the cross-attention shape (288 keys), and checks
// every non-diagnostic variant against variant 0.   build + run: tools/attn_lab/run.sh [variants...]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../bevgen_amd/csrc/common.h"
#include "../../bevgen_amd/csrc/kernels.h"
#include "../../bevgen_amd/csrc/profiler.h"

namespace bevgen {
extern int g_attn_variant, g_attn_extra_lds;
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

int main(int argc, char** argv) {
    const int B = 16, H = 16, Nq = 1536;
    std::vector<int> variants;
    for (int i = 1; i < argc; ++i) variants.push_back(atoi(argv[i]));
    if (variants.empty()) variants = {0, 9, 1024 + 4096 + 256 + 10, 1024 + 65536 + 4096 + 256 + 10};
    for (int shape = 0; shape < 2; ++shape) {
        const int Nk_pad = shape == 0 ? 1568 : 288;
        const size_t nq = (size_t)B * H * Nq * 64, nk = (size_t)B * H * Nk_pad * 64;
        std::vector<_Float16> qh, ql, kh, kl, vh, vl;
        fill_planes(qh, ql, nq, 3.0f);   // l2-normalised q,k times scales: |score| of a few units in the base-2 domain
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
        // tile-major copies of the V^T planes: [b,h][tile][64][32]
        std::vector<_Float16> vth(nk), vtl(nk);
        for (size_t bh = 0; bh < (size_t)B * H; ++bh)
            for (int d = 0; d < 64; ++d)
                for (int j = 0; j < Nk_pad; ++j) {
                    const size_t src = bh * 64 * Nk_pad + (size_t)d * Nk_pad + j, dst = bh * 64 * Nk_pad + (size_t)(j / 32) * 2048 + d * 32 + (j % 32);
                    vth[dst] = vh[src]; vtl[dst] = vl[src];
                }
        const _Float16 *VTh_row = a.VTh, *VTl_row = a.VTl, *VTh_tile = to_dev(vth), *VTl_tile = to_dev(vtl);
        a.B = B; a.H = H; a.Nq = Nq; a.Nk_pad = Nk_pad; a.scale = 1.f;
        a.o_bstride = (long)Nq * H * 64; a.o_qstride = (long)H * 64; a.o_hstride = 64;
        const size_t no = (size_t)B * Nq * H * 64;
        float *O, *Oref;
        HIP_CHECK(hipMalloc(&O, no * 4));
        HIP_CHECK(hipMalloc(&Oref, no * 4));
        a.Op = nullptr;
        std::vector<float> ref(no), out(no);
        hipEvent_t e0, e1;
        HIP_CHECK(hipEventCreate(&e0));
        HIP_CHECK(hipEventCreate(&e1));
        const double flop = 4.0 * B * H * (double)Nq * Nk_pad * 64;
        for (int lds : {0}) {   // (48 KiB of extra dynamic LDS = one wave per SIMD: 951 us vs 761 us for variant 0)
            std::vector<double> best(variants.size(), 1e30);
            for (int round = 0; round < 4; ++round)
                for (size_t vi = 0; vi < variants.size(); ++vi) {
                    g_attn_variant = variants[vi]; g_attn_extra_lds = lds;
                    a.VTh = (variants[vi] & 16) ? VTh_tile : VTh_row; a.VTl = (variants[vi] & 16) ? VTl_tile : VTl_row;
                    a.O = O;
                    launch_attention_split(a, 0);
                    HIP_CHECK(hipEventRecord(e0, 0));
                    const int reps = 8;
                    for (int i = 0; i < reps; ++i) launch_attention_split(a, 0);
                    HIP_CHECK(hipEventRecord(e1, 0));
                    HIP_CHECK(hipEventSynchronize(e1));
                    float ms = 0;
                    HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
                    best[vi] = std::fmin(best[vi], ms * 1e3 / reps);
                }
            for (size_t vi = 0; vi < variants.size(); ++vi) {
                const int v = variants[vi];
                g_attn_variant = v; g_attn_extra_lds = lds;
                a.VTh = (v & 16) ? VTh_tile : VTh_row; a.VTl = (v & 16) ? VTl_tile : VTl_row;
                a.O = (v == 0 && lds == 0) ? Oref : O;
                launch_attention_split(a, 0);
                HIP_CHECK(hipDeviceSynchronize());
                const double us = best[vi];
                double maxdiff = -1;
                if ((v & 255 & ~0) < 32 || v >= 1024) {
                    if (v == 0 && lds == 0) HIP_CHECK(hipMemcpy(ref.data(), Oref, no * 4, hipMemcpyDeviceToHost));
                    else {
                        HIP_CHECK(hipMemcpy(out.data(), O, no * 4, hipMemcpyDeviceToHost));
                        maxdiff = 0;
                        for (size_t i = 0; i < no; ++i) maxdiff = std::fmax(maxdiff, std::fabs((double)out[i] - ref[i]));
                    }
                }
                double err64 = -1;
                if ((v & 255 & ~0) < 32 || v >= 1024) {   // fp64 evaluation of three output rows from the same planes
                    const std::vector<float>& o = (v == 0 && lds == 0) ? ref : out;
                    err64 = 0;
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
                        for (int d = 0; d < 64; ++d) err64 = std::fmax(err64, std::fabs(od[d] / l - o[((size_t)b * Nq + q) * H * 64 + hd * 64 + d]));
                    }
                }
                printf("Nk=%4d extra_lds=%5d variant=%3d: %8.1f us  %6.1f TF-equiv  maxdiff vs v0 %.3g  max err vs fp64 (12 rows) %.3g\n", Nk_pad, lds, v, us, flop / us * 1e-6, maxdiff,
                       err64);
                fflush(stdout);
            }
        }
        for (int tv : {1024 + 4096 + 256 + 10 + (1 << 20), 1024 + 65536 + 4096 + 256 + 10 + (1 << 20)}) {   // phase trace of one workgroup (waves 0 and 4 = the two waves of SIMD 0)
            g_attn_variant = tv;
            a.VTh = (tv & 16) ? VTh_tile : VTh_row; a.VTl = (tv & 16) ? VTl_tile : VTl_row; a.O = O;
            launch_attention_split(a, 0);
            HIP_CHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> tr(8 * 16 * 8);
            attn_lab_read_trace(tr.data());
            const int nt = Nk_pad / 32;
            for (int w : {0, 4}) {
                printf("trace variant %d Nk=%d wave %d: per tile [M-phase | wait at barrier | staging | softmax | wait at barrier] cycles\n", tv & 0xFFFFF, Nk_pad, w);
                for (int t = 2; t < (nt < 9 ? nt : 9); ++t) {
                    const unsigned long long* e = &tr[(w * 16 + t) * 8];
                    printf("   t=%2d  start %8lld  M %5llu  bar %5llu  stage %5llu  softmax %5llu  bar %5llu\n", t, (long long)(e[0] - tr[(0 * 16 + 2) * 8]), e[1] - e[0], e[2] - e[1],
                           e[3] - e[2], e[4] - e[3], e[5] - e[4]);
                }
            }
        }
        for (const void* ptr : {(const void*)a.Qh, (const void*)a.Ql, (const void*)a.Kh, (const void*)a.Kl, (const void*)VTh_row, (const void*)VTl_row, (const void*)VTh_tile,
                                (const void*)VTl_tile, (const void*)a.bias, (const void*)bpk, (const void*)O, (const void*)Oref})
            HIP_CHECK(hipFree(const_cast<void*>(ptr)));
    }
    return 0;
}
