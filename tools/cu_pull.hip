// Micro-benchmark: how fast can one CU pull a few hundred KB that every step re-reads (L2 / Infinity Cache),
// with and without a 58 MB weight stream between the reads (what the decode LSTM kernels do).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NL, int THREADS>
__global__ __launch_bounds__(THREADS) void reader(const f32x4* buf, size_t wg_stride_f4, float* out) {
    const f32x4* p = buf + (size_t)blockIdx.x * wg_stride_f4 + threadIdx.x;
    f32x4 r[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) r[i] = p[(size_t)i * THREADS];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i) s += r[i][0] + r[i][1] + r[i][2] + r[i][3];
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <bool NT>
__global__ __launch_bounds__(512) void streamer(const f32x4* buf, int n_per_thread, float* out) {
    const f32x4* p = buf + (size_t)blockIdx.x * 512 * n_per_thread + threadIdx.x;
    float s = 0.f;
    for (int i = 0; i < n_per_thread; i += 8) {
        f32x4 r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = NT ? __builtin_nontemporal_load(p + (size_t)(i + j) * 512) : p[(size_t)(i + j) * 512];
#pragma unroll
        for (int j = 0; j < 8; ++j) s += r[j][0] + r[j][3];
    }
    if (s == 12345.678f) out[threadIdx.x] = s;
}
template <typename F> double run_graph(hipStream_t s, int n, F body) {
    hipGraph_t g; hipGraphExec_t ge;
    (void)hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
    for (int i = 0; i < n; ++i) body();
    (void)hipStreamEndCapture(s, &g);
    (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
    double best = 1e9;
    for (int r = 0; r < 5; ++r) {
        auto t0 = std::chrono::high_resolution_clock::now();
        (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
        double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
        if (us < best) best = us;
    }
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    return best / n;
}
int main() {
    hipStream_t s; (void)hipStreamCreate(&s);
    f32x4 *small, *big; float* out;
    const size_t small_bytes = 160u << 20, big_bytes = 64u << 20;
    (void)hipMalloc(&small, small_bytes); (void)hipMalloc(&big, big_bytes); (void)hipMalloc(&out, 1 << 16);
    (void)hipMemset(small, 0, small_bytes); (void)hipMemset(big, 0, big_bytes);
    const int N = 300;
    constexpr int NL = 29;                       // 29 x 16 B x 1024 threads = 475 KB per workgroup
    const size_t wg_f4 = (size_t)NL * 1024;
    const int npt = 58 * 1024 * 1024 / 16 / (256 * 512) / 8 * 8;   // ~58 MB streamed by 256 workgroups
    auto rd = [&](int nwg, bool shared) { hipLaunchKernelGGL((reader<NL, 1024>), dim3(nwg), dim3(1024), 0, s, small, shared ? 0 : wg_f4, out); };
    auto st = [&](bool nt) { if (nt) hipLaunchKernelGGL((streamer<true>), dim3(256), dim3(512), 0, s, big, npt, out);
                             else hipLaunchKernelGGL((streamer<false>), dim3(256), dim3(512), 0, s, big, npt, out); };
    double t_st = run_graph(s, N, [&] { st(false); });
    double t_stnt = run_graph(s, N, [&] { st(true); });
    printf("streamer 58MB default: %.2f us (%.2f TB/s)   nt: %.2f us\n", t_st, 58.0 * 1.048576 / t_st, t_stnt);
    for (int nwg : {1, 32, 256}) {
        for (int shared = 1; shared >= 0; --shared) {
            if (!shared && nwg == 256 && wg_f4 * 16 * 256 > small_bytes) { }
            double a = run_graph(s, N, [&] { rd(nwg, shared); });
            double b = run_graph(s, N, [&] { st(false); rd(nwg, shared); }) - t_st;
            double c = run_graph(s, N, [&] { st(true); rd(nwg, shared); }) - t_stnt;
            printf("reader %3d WG x 475KB %s: alone %.2f us (%.0f GB/s/CU) | after 58MB stream %.2f us | after nt stream %.2f us\n",
                   nwg, shared ? "shared  " : "distinct", a, 475.0 * 1.024 / a / 1e0 * 1e-3 * 1e3 / 1e0, b, c);
        }
    }
    return 0;
}
