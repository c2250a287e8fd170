// Stand-alone bench of the Winograd F(4,5) conv kernel at the postnet's 512 -> 512 shape (32 utterances x 1000 frames), with
// cycle stamps of one step of workgroup (0,0) wave 0: [0] step start, [1] requests issued, [2] MFMAs issued, [3] next slice
// transformed + stored, [4] past the barrier.  (The stamps of the LAST executed step survive.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGT_WINO_STAMPS -o tools/wino_bench tools/wino_bench.hip && tools/wino_bench
#include "../gst_tacotron_amd/csrc/gemm_conv.hip"
#include "wino_w4.h"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
int main() {
    const int B = 32, T = 1000, C = 512, N = 512;
    std::vector<float> hx((size_t)B * T * C), hu((size_t)8 * C * N);
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : hx) v = nd(rng);
    for (auto& v : hu) v = nd(rng) * 0.02f;
    float *x, *u, *out; CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&u, hu.size() * 4)); CK(hipMalloc(&out, (size_t)B * T * N * 4));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(u, hu.data(), hu.size() * 4, hipMemcpyHostToDevice));
    ConvGemmArgs a{}; a.x = x; a.out = out; a.ldo = N; a.B = B; a.T = T; a.Cin = C; a.N = N; a.taps = 5; a.pad_before = 2; a.act = ACT_NONE;
    a.wino_u4 = u; a.wino_u = u; a.wino_cin = C;
    const int P4 = B * ((T + 3) / 4);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(gt_conv_wino5_kernel<4>, dim3(8 * (((P4 + 63) / 64 + 7) / 8) * 4), dim3(WT), 0, 0, a, a.wino_u4);
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("F(4,5) 512->512, 32 x 1000 frames: %.1f us / launch = %.1f TF of MFMA work (%.1f TF direct-equivalent)\n", ms * 100, 33.55e9 / (ms * 1e-4) * 1e-12,
               83.9e9 / (ms * 1e-4) * 1e-12);
    }
    // the 4-wave x 512-register form (one workgroup per CU, two accumulator tiles per wave): bitwise the 8-wave kernel, and its time
    {
        std::vector<float> r8((size_t)B * T * N), r4(r8.size());
        CK(hipMemset(out, 0, r8.size() * 4));
        hipLaunchKernelGGL(gt_conv_wino5_kernel<4>, dim3(8 * (((P4 + 63) / 64 + 7) / 8) * 4), dim3(WT), 0, 0, a, a.wino_u4);
        CK(hipMemcpy(r8.data(), out, r8.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(out, 0, r8.size() * 4));
        CK(gt_conv_wino5_w4_init());
        const int Q4 = w4_tiles(a, 4);
        const dim3 g4(8 * (((Q4 + 63) / 64 + 7) / 8) * 4);
        hipLaunchKernelGGL(gt_conv_wino5_w4_kernel<4>, g4, dim3(WT4), w4_lds_bytes(), 0, a, a.wino_u4);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(r4.data(), out, r4.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0; double md = 0;
        for (size_t i = 0; i < r8.size(); ++i) { if (memcmp(&r8[i], &r4[i], 4)) ++bad; md = std::max(md, (double)fabsf(r8[i] - r4[i])); }
        printf("4-wave form vs 8-wave form: %zu of %zu values differ (max abs %.3g)\n", bad, r8.size(), md);
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(gt_conv_wino5_w4_kernel<4>, g4, dim3(WT4), w4_lds_bytes(), 0, a, a.wino_u4);
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("F(4,5) 4 waves x 512 registers: %.1f us / launch = %.1f TF of MFMA work (%.1f TF direct-equivalent)\n", ms * 100, 33.55e9 / (ms * 1e-4) * 1e-12,
                   83.9e9 / (ms * 1e-4) * 1e-12);
        }
#ifdef GT_WINO_STAMPS
    {   // (the 4-wave F(4,5) kernel ran last: its stamps, one per PAIR step = two transform-domain GEMMs on two tiles = 64 MFMAs per wave)
        static unsigned long long st[1024]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(gt_wino_stamp), sizeof(st)));
        std::vector<long long> dt;
        for (int i = 1; i < 63; ++i) dt.push_back((long long)(st[i] - st[i - 1]));
        std::sort(dt.begin(), dt.end());
        printf("4-wave kernel, s_memtime ticks per pair step (64 MFMAs = 4096 cycles of the pipe), WG 0 wave 0: min %lld median %lld p90 %lld max %lld; first->last %lld over 62 steps\n",
               dt[0], dt[dt.size() / 2], dt[dt.size() * 9 / 10], dt.back(), (long long)(st[62] - st[0]));
#ifdef GT_W4_FINE
        // inside pair steps 20..27, ticks between: step start (past the barrier) -> DMA + first operand reads + row requests issued ->
        // region 0 -> region 1 -> region 2 -> region 3 done -> vmcnt(0) passed | -> past the barrier
        for (int stp = 20; stp < 28; ++stp) {
            printf("  pair step %d (J = %d):", stp, stp % 4);
            for (int k = 1; k < 7; ++k) printf(" %5lld", (long long)(st[512 + 8 * stp + k] - st[512 + 8 * stp + k - 1]));
            printf("  | barrier %5lld\n", (long long)(st[512 + 8 * (stp + 1)] - st[512 + 8 * stp + 6]));
        }
#endif
    }
#endif
        // F(2,5), N = 80 (the last postnet layer's shape)
        ConvGemmArgs b2 = a; b2.N = 80; b2.ldo = 80;
        const int P2 = B * ((T + 1) / 2), Q2 = w4_tiles(b2, 2);
        CK(hipMemset(out, 0, r8.size() * 4));
        hipLaunchKernelGGL(gt_conv_wino5_kernel<2>, dim3(8 * (((P2 + 63) / 64 + 7) / 8)), dim3(WT), 0, 0, b2, b2.wino_u);
        CK(hipMemcpy(r8.data(), out, (size_t)B * T * 80 * 4, hipMemcpyDeviceToHost));
        CK(hipMemset(out, 0, r8.size() * 4));
        hipLaunchKernelGGL(gt_conv_wino5_w4_kernel<2>, dim3(8 * (((Q2 + 63) / 64 + 7) / 8)), dim3(WT4), w4_lds_bytes(), 0, b2, b2.wino_u);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(r4.data(), out, (size_t)B * T * 80 * 4, hipMemcpyDeviceToHost));
        bad = 0;
        for (size_t i = 0; i < (size_t)B * T * 80; ++i) if (memcmp(&r8[i], &r4[i], 4)) ++bad;
        printf("F(2,5) 512 -> 80, 4-wave vs 8-wave: %zu values differ\n", bad);
        for (int which = 0; which < 2; ++which) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 10; ++i) {
                if (which) hipLaunchKernelGGL(gt_conv_wino5_w4_kernel<2>, dim3(8 * (((Q2 + 63) / 64 + 7) / 8)), dim3(WT4), w4_lds_bytes(), 0, b2, b2.wino_u);
                else hipLaunchKernelGGL(gt_conv_wino5_kernel<2>, dim3(8 * (((P2 + 63) / 64 + 7) / 8)), dim3(WT), 0, 0, b2, b2.wino_u);
            }
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("F(2,5) 512 -> 80 %s: %.1f us / launch\n", which ? "4-wave" : "8-wave", ms * 100);
        }
    }
    return 0;
}
