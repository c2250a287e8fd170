// Stand-alone bench of the bf16 five-tap Conv1D kernel (gemm_conv.hip gt_conv5_bf16_kernel) at the mixed-precision postnet's 512 -> 512
// shape (64 utterances x 1000 frames, bf16 activations in and out), with ablations: -DC5_NO_MFMA (everything but the matrix instructions),
// -DC5_NO_READS_A / -DC5_NO_READS_B (one fragment read per step instead of one per 16 k: wrong results, same staging traffic).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DC5_NO_MFMA] -o tools/conv5_bench tools/conv5_bench.hip && tools/conv5_bench
#include "../gst_tacotron_amd/csrc/gemm_conv.hip"
#include "../gst_tacotron_amd/csrc/conv_wino_split.hip"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static uint16_t bfb(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16); }
static float bff(uint16_t b) { const uint32_t w = (uint32_t)b << 16; float f; memcpy(&f, &w, 4); return f; }
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, T = 1000, C = 512, N = 512, ldk = 5 * C;
    std::vector<uint16_t> hx((size_t)B * T * C), hw((size_t)N * ldk);
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : hx) v = bfb(nd(rng));
    for (auto& v : hw) v = bfb(nd(rng) * 0.02f);          // [n][tap][k]
    void *x, *w, *out;
    CK(hipMalloc(&x, hx.size() * 2)); CK(hipMalloc(&w, hw.size() * 2)); CK(hipMalloc(&out, (size_t)B * T * N * 2));
    CK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    ConvGemmArgs a{}; a.x = reinterpret_cast<const float*>(x); a.out = reinterpret_cast<float*>(out); a.ldo = N; a.B = B; a.T = T; a.Cin = C; a.N = N; a.taps = 5; a.pad_before = 2;
    a.act = ACT_NONE; a.wt_bf16 = w; a.ldk = ldk; a.x_bf16 = 1; a.out_bf16 = 1;
    CK(gt_conv5_bf16_init());
    CK(gt_launch_conv_gemm(a, 0));
    CK(hipDeviceSynchronize());
    std::vector<uint16_t> ho((size_t)B * T * N);
    CK(hipMemcpy(ho.data(), out, ho.size() * 2, hipMemcpyDeviceToHost));
    // spot check against a double sum of the same bf16 operands
    double md = 0, mx = 0;
    for (int q = 0; q < 64; ++q) {
        const int b = (q * 7) % B, t = (q * 131) % T, n = (q * 37) % N;
        double s = 0;
        for (int tap = 0; tap < 5; ++tap) { const int tt = t + tap - 2; if (tt < 0 || tt >= T) continue;
            for (int k = 0; k < C; ++k) s += (double)bff(hx[((size_t)b * T + tt) * C + k]) * bff(hw[(size_t)n * ldk + tap * C + k]); }
        md = std::max(md, fabs(s - bff(ho[((size_t)b * T + t) * N + n]))); mx = std::max(mx, fabs(s));
    }
    printf("spot check: max abs difference %.3g at max |y| %.3g\n", md, mx);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) CK(gt_launch_conv_gemm(a, 0));
        CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("five-tap bf16 512->512, %d x 1000 frames: %.1f us / launch (%.0f TF)\n", B, ms * 100, 2.0 * B * T * 5 * C * N / (ms * 1e-4) * 1e-12);
    }
    return 0;
}
