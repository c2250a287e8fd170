// Recurrent-half worker jobs at batches above 32 rows: the K-split multi-chunk body (lean_body.h gt_lean_partial_mc) against the M-split
// body (tools/msplit_body.h gt_msplit_partial), same operands, one job (pair of tiles, all rows) per workgroup, 256 workgroups = one per CU.
// Prints the time per launch (= per job) and compares the outputs bitwise.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/msplit_bench tools/msplit_bench.hip && tools/msplit_bench [rows]
//   (-DGT_MSPLIT_NO_X: the activation loads taken out -- what the matrix pipe alone sustains; -DGT_X_SC1=0: plain instead of sc1 activation loads)
#include "msplit_body.h"
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NW, int KPW>
__global__ __launch_bounds__(NW * 64) void k_ksplit(LeanPartialArgs A, int mchunks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    gt_lean_partial_mc<NW, KPW, 2>(A, 2 * blockIdx.x, 2, 0, mchunks, lds);
}
__device__ unsigned long long g_clk[4];
__device__ int g_reps = 1;        // jobs per workgroup and launch (-> how long one launch keeps the matrix pipe busy)
template <int NWAVES, int ORDER, int NKB, int TPW>
__global__ __launch_bounds__(NWAVES * 64) void k_msplit(LeanPartialArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int rep = 0; rep < g_reps; ++rep) {
        gt_msplit_partial<NWAVES, ORDER, NKB, TPW>(A, 2 * blockIdx.x, 2, 0, A.MT, lds);
        __syncthreads();
    }
    if (blockIdx.x == 7 && threadIdx.x == 0) { g_clk[0] = __builtin_amdgcn_s_memtime() - c0; g_clk[1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 128, NKB = 64, MT = (M + 15) / 16, NJOBS = 256, NT = 2 * NJOBS;
    std::vector<float> hw((size_t)NT * NKB * 256), hx((size_t)NKB * MT * 256), hb(NT * 16);
    std::mt19937 rng(1); std::normal_distribution<float> nd(0.f, 1.f);
    for (auto& v : hw) v = nd(rng) * 0.03f;
    for (auto& v : hx) v = nd(rng);
    for (auto& v : hb) v = nd(rng);
    float *w, *x, *b, *o1, *o2;
    const size_t on = (size_t)NT * MT * 16 * 16;
    CK(hipMalloc(&w, hw.size() * 4)); CK(hipMalloc(&x, hx.size() * 4)); CK(hipMalloc(&b, hb.size() * 4)); CK(hipMalloc(&o1, on * 4)); CK(hipMalloc(&o2, on * 4));
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), hx.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    LeanPartialArgs A{w, b, x, o1, MT};
    const int mchunks = (M + 31) / 32;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> r1(on), r2(on);
    auto timeit = [&](const char* name, auto launch, float* out, std::vector<float>& res) {
        A.partial_out = out;
        CK(hipMemset(out, 0, on * 4));
        launch(); CK(hipDeviceSynchronize());
        CK(hipMemcpy(res.data(), out, on * 4, hipMemcpyDeviceToHost));
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, 0));
            for (int i = 0; i < 20; ++i) launch();
            CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = std::min(best, ms * 50.f);
        }
        const double flop = 2.0 * M * 1024 * 32 * NJOBS;
        printf("%-58s %7.2f us per job (256 jobs, one per CU) = %5.1f TFLOP/s\n", name, best, flop / (best * 1e-6) * 1e-12);
    };
    auto cmp = [&](const char* what) {
        size_t bad = 0;
        for (size_t i = 0; i < on; ++i) if (memcmp(&r1[i], &r2[i], 4)) ++bad;
        printf("  %s: %zu of %zu outputs differ\n", what, bad, on);
    };
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_ksplit<16, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msplit<8, 8, 64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_msplit<16, 16, 64, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    constexpr size_t L8 = LeanLds<8, 2>::kFloats * 4, L16 = LeanLds<16, 2>::kFloats * 4, LM = MSplitLds<64>::kFloats * 4;
    const int reps = argc > 2 ? atoi(argv[2]) : 1;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_reps), &reps, sizeof(int)));
    if (reps > 1) printf("(the M-split launches repeat their job %d times: their 'us per job' lines are per LAUNCH)\n", reps);
    printf("rows %d (%d M-tiles, %d chunks), K = 1024, fp32\n", M, MT, mchunks);
    timeit("K-split, 8 waves x 8 k-blocks (LSTM / projection launches)", [&] { hipLaunchKernelGGL((k_ksplit<8, 8>), dim3(NJOBS), dim3(512), L8, 0, A, mchunks); }, o1, r1);
    timeit("M-split, 8 waves, both tiles per wave, order of 8", [&] { hipLaunchKernelGGL((k_msplit<8, 8, 64, 2>), dim3(NJOBS), dim3(512), LM, 0, A); }, o2, r2);
    cmp("M-split vs K-split (8)");
    { unsigned long long c[4]; CK(hipMemcpyFromSymbol(c, HIP_SYMBOL(g_clk), sizeof(c)));
      printf("  workgroup 7, wave 0 of the M-split kernel: %llu shader clocks in %.2f us = %.2f GHz\n", c[0], c[1] / 100.0, c[0] / (c[1] * 10.0)); }
    timeit("K-split, 16 waves x 4 k-blocks (front launch workers)", [&] { hipLaunchKernelGGL((k_ksplit<16, 4>), dim3(NJOBS), dim3(1024), L16, 0, A, mchunks); }, o1, r1);
    timeit("M-split, 16 waves, one tile per wave, order of 16", [&] { hipLaunchKernelGGL((k_msplit<16, 16, 64, 1>), dim3(NJOBS), dim3(1024), LM, 0, A); }, o2, r2);
    cmp("M-split vs K-split (16)");
    return 0;
}
