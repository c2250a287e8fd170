// Micro-benchmark 7 (round 2): what does a SIDE BRANCH cost in a hipGraph of short dependent kernels?
//
// Main chain per iteration: A(t) -> B(t) (each `main_us` of spinning on 256 workgroups).  Variants:
//   0  main chain only
//   1  + side kernel R(t) (side_us on `side_wgs` workgroups), captured on a second stream: R(t) depends on A(t-1), B(t) depends on R(t)
//      (the shape of "recurrent half of step t computed beside the rest of step t-1 / front of step t")
//   2  the same R(t) work serialised INTO the main chain (A -> R -> B), for reference
// Reports us per iteration.   hipcc --offload-arch=gfx950 -O3 -o tools/graph_branch tools/graph_branch.hip && tools/graph_branch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(512) void k_spin(int ticks, float* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < ticks) __builtin_amdgcn_s_sleep(1);
    if (ticks < 0) sink[0] = 1.f;
}

int main() {
    float* sink; CK(hipMalloc(&sink, 64));
    hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 300;
    hipEvent_t evA[iters + 1], evR[iters + 1];
    for (int i = 0; i <= iters; ++i) { CK(hipEventCreateWithFlags(&evA[i], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&evR[i], hipEventDisableTiming)); }
    for (int main_us : {3, 6})
        for (int side_us : {4, 8})
            for (int side_wgs : {64, 256})
                for (int variant = 0; variant < 3; ++variant) {
                    if (variant == 0 && (side_us != 4 || side_wgs != 64)) continue;
                    hipGraph_t g; hipGraphExec_t ge;
                    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
                    for (int t = 0; t < iters; ++t) {
                        hipLaunchKernelGGL(k_spin, dim3(256), dim3(512), 0, s0, main_us * 100, sink);              // A(t)
                        if (variant == 1) {
                            // R(t+1) may start once A(t) is done; B(t+1) needs it.  R(t): after A(t-1)
                            if (t > 0) {
                                CK(hipStreamWaitEvent(s0, evR[t], 0));                                                 // B(t) waits for R(t)
                            }
                            CK(hipEventRecord(evA[t], s0));
                            CK(hipStreamWaitEvent(s1, evA[t], 0));
                            hipLaunchKernelGGL(k_spin, dim3(side_wgs), dim3(512), 0, s1, side_us * 100, sink);      // R(t+1)
                            CK(hipEventRecord(evR[t + 1], s1));
                        } else if (variant == 2) {
                            hipLaunchKernelGGL(k_spin, dim3(side_wgs), dim3(512), 0, s0, side_us * 100, sink);
                        }
                        hipLaunchKernelGGL(k_spin, dim3(256), dim3(512), 0, s0, main_us * 100, sink);              // B(t)
                    }
                    if (variant == 1) CK(hipStreamWaitEvent(s0, evR[iters], 0));                                       // join
                    CK(hipStreamEndCapture(s0, &g));
                    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                    float ms = 0;
                    for (int rep = 0; rep < 3; ++rep) {
                        CK(hipEventRecord(e0, s0)); CK(hipGraphLaunch(ge, s0)); CK(hipEventRecord(e1, s0));
                        CK(hipStreamSynchronize(s0)); CK(hipEventElapsedTime(&ms, e0, e1));
                    }
                    const char* names[] = {"main chain only (A -> B)", "side branch R beside the chain", "R serialised into the chain"};
                    printf("A,B %d us x 256 WGs | R %d us x %3d WGs | %-32s: %.2f us / iteration\n", main_us, side_us, side_wgs, names[variant], ms * 1e3 / iters);
                    fflush(stdout);
                    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
                }
    return 0;
}
