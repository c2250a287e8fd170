// Stand-alone bench of the split-bf16 Winograd kernel (conv_wino_split.hip) at the postnet's 512 -> 512 shape (32 utterances x 1000
// frames) beside the fp32-MFMA kernel, with ablations: -DWS_NO_MFMA (everything but the matrix instructions), -DWS_NO_SPLIT (the V planes
// without the split arithmetic: wrong results, same traffic).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DWS_NO_MFMA] -o tools/wino_split_bench tools/wino_split_bench.hip && tools/wino_split_bench
#include "../gst_tacotron_amd/csrc/gemm_conv.hip"
#include "../gst_tacotron_amd/csrc/conv_wino_split.hip"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static uint16_t bfb(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
int main() {
    const int B = 32, T = 1000, C = 512, N = 512;
    std::vector<float> hx((size_t)B * T * C), hu((size_t)8 * C * N);
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : hx) v = nd(rng);
    for (auto& v : hu) v = nd(rng) * 0.02f;
    std::vector<uint16_t> hs((size_t)8 * 3 * N * C);
    for (int xi = 0; xi < 8; ++xi) for (int k = 0; k < C; ++k) for (int n = 0; n < N; ++n) {
        double v = hu[((size_t)xi * C + k) * N + n];
        for (int p = 0; p < 3; ++p) { const uint16_t b = bfb((float)v); const uint32_t w = (uint32_t)b << 16; float f; memcpy(&f, &w, 4); v -= f; hs[(((size_t)xi * 3 + p) * N + n) * C + k] = b; }
    }
    float *x, *u, *out, *out2; void* us;
    CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&u, hu.size() * 4)); CK(hipMalloc(&out, (size_t)B * T * N * 4)); CK(hipMalloc(&out2, (size_t)B * T * N * 4)); CK(hipMalloc(&us, hs.size() * 2));
    CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(u, hu.data(), hu.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(us, hs.data(), hs.size() * 2, hipMemcpyHostToDevice));
    ConvGemmArgs a{}; a.x = x; a.out = out; a.ldo = N; a.B = B; a.T = T; a.Cin = C; a.N = N; a.taps = 5; a.pad_before = 2; a.act = ACT_NONE;
    a.wino_u4 = u; a.wino_u = u; a.wino_cin = C; a.wino_s4 = us; a.wino_s = us; a.wino_npad = N; a.wino_x3 = getenv("WINO_X3") ? 1 : 0;
    CK(gt_conv_wino5s_init());
    const int P4 = B * ((T + 3) / 4);
    const dim3 grid(8 * (((P4 + 63) / 64 + 7) / 8) * 4);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(gt_conv_wino5_kernel<4>, grid, dim3(WT), 0, 0, a, a.wino_u4);
    ConvGemmArgs a2 = a; a2.out = out2;
    CK(gt_launch_conv_wino5s(a2, 4, 0));
    CK(hipDeviceSynchronize());
    std::vector<float> r0((size_t)B * T * N), r1(r0.size());
    CK(hipMemcpy(r0.data(), out, r0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(r1.data(), out2, r1.size() * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0; for (size_t i = 0; i < r0.size(); ++i) { md = std::max(md, (double)fabsf(r0[i] - r1[i])); mx = std::max(mx, (double)fabsf(r0[i])); }
    printf("split-bf16 x6 vs fp32 MFMA Winograd: max abs difference %.3g at max |y| %.3g\n", md, mx);
    for (int which = 0; which < 2; ++which)
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 10; ++i) {
                if (which) CK(gt_launch_conv_wino5s(a2, 4, 0));
                else hipLaunchKernelGGL(gt_conv_wino5_kernel<4>, grid, dim3(WT), 0, 0, a, a.wino_u4);
            }
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("F(4,5) 512->512, 32 x 1000 frames, %s: %.1f us / launch (%.1f TF direct-equivalent)\n", which ? "split-bf16 x6" : "fp32 MFMA    ", ms * 100, 83.9e9 / (ms * 1e-4) * 1e-12);
        }
    return 0;
}
