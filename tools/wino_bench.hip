// Stand-alone bench of the Winograd F(4,5) conv kernel at the postnet's 512 -> 512 shape (32 utterances x 1000 frames), with
// cycle stamps of one step of workgroup (0,0) wave 0: [0] step start, [1] requests issued, [2] MFMAs issued, [3] next slice
// transformed + stored, [4] past the barrier.  (The stamps of the LAST executed step survive.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGT_WINO_STAMPS -o tools/wino_bench tools/wino_bench.hip && tools/wino_bench
#include "../gst_tacotron_amd/csrc/gemm_conv.hip"
#include <cstdio>
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
#ifdef GT_WINO_STAMPS
    static unsigned long long st[1024]; CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(gt_wino_stamp), sizeof(st)));
    std::vector<long long> dt;
    for (int i = 1; i < 127; ++i) dt.push_back((long long)(st[i] - st[i - 1]));
    std::sort(dt.begin(), dt.end());
    printf("cycles per step (s_memtime ticks), WG 0 wave 0: min %lld median %lld p90 %lld max %lld; first->last %lld over 126 steps\n", dt[0], dt[dt.size() / 2],
           dt[dt.size() * 9 / 10], dt.back(), (long long)(st[126] - st[0]));
#endif
    return 0;
}
