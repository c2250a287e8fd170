// Micro-benchmark: cost of a dependent kernel boundary on this box (eager vs hipGraph), for trivial kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
struct Big { char b[400]; };
__global__ void k_empty(float* p) { if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ void k_big(Big a, float* p) { if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] += a.b[3]; }
__global__ void k_touch(float* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <typename F> double run_graph(hipStream_t s, int n, F body) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
    for (int i = 0; i < n; ++i) body();
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    double best = 1e9;
    for (int r = 0; r < 5; ++r) {
        auto t0 = std::chrono::high_resolution_clock::now();
        hipGraphLaunch(ge, s); hipStreamSynchronize(s);
        double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
        if (us < best) best = us;
    }
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return best / n;
}
int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float* p; CK(hipMalloc(&p, 1 << 24)); CK(hipMemset(p, 0, 1 << 24));
    const int N = 2000;
    // warm the clocks
    for (int i = 0; i < 20000; ++i) hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, s, p, 1 << 20);
    CK(hipStreamSynchronize(s));
    printf("graph  empty<<<1,64>>>      : %.2f us/kernel\n", run_graph(s, N, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p); }));
    printf("graph  empty<<<256,256>>>   : %.2f us/kernel\n", run_graph(s, N, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, p); }));
    printf("graph  empty<<<256,512>>>   : %.2f us/kernel\n", run_graph(s, N, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, s, p); }));
    Big b{}; 
    printf("graph  bigargs<<<256,256>>> : %.2f us/kernel\n", run_graph(s, N, [&] { hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s, b, p); }));
    printf("graph  touch 1M floats      : %.2f us/kernel\n", run_graph(s, N, [&] { hipLaunchKernelGGL(k_touch, dim3(4096), dim3(256), 0, s, p, 1 << 20); }));
    // eager
    for (int rep = 0; rep < 2; ++rep) {
        auto t0 = std::chrono::high_resolution_clock::now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, p);
        CK(hipStreamSynchronize(s));
        double us = std::chrono::duration<double, std::micro>(std::chrono::high_resolution_clock::now() - t0).count();
        printf("eager  empty<<<256,256>>>   : %.2f us/kernel\n", us / N);
    }
    return 0;
}
