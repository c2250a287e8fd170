// Micro-benchmark 8 (round 2): the kernel-boundary floor of a hipGraph chain as a function of the launch shape.
// 400 dependent launches of a kernel that does nothing (or writes one dirty line per workgroup), per (grid, block, kernarg bytes).
//   hipcc --offload-arch=gfx950 -O3 -o tools/launch_floor tools/launch_floor.hip && tools/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
struct Big { float* p; int dirty; int pad[120]; };
struct Small { float* p; int dirty; };
template <class A> __global__ void k_nop(A a) {
    if (a.dirty && threadIdx.x == 0) a.p[blockIdx.x * 32] = 1.f;
}
template <class A> float run(int grid, int block, int dirty, float* buf, hipStream_t st) {
    A a{}; a.p = buf; a.dirty = dirty;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(k_nop<A>, dim3(grid), dim3(block), 0, st, a);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, st)); CK(hipGraphLaunch(ge, st)); CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return ms * 1e3f / 400;
}
int main() {
    float* buf; CK(hipMalloc(&buf, 1 << 20));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int block : {64, 256, 512, 1024})
        for (int grid : {8, 32, 128, 256, 512, 1024})
            printf("grid %4d x block %4d: %.2f us / launch (16 B args, clean) | %.2f (496 B args) | %.2f (one dirty line per workgroup)\n", grid, block,
                   run<Small>(grid, block, 0, buf, st), run<Big>(grid, block, 0, buf, st), run<Small>(grid, block, 1, buf, st));
    return 0;
}
