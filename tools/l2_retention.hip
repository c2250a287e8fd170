// Micro-benchmark: does a weight tile read by a kernel stay in the reading XCD's L2 for the NEXT kernel of the stream?
// 256 blocks x 64 KB (16.8 MB, 2.1 MB per XCD: fits the 4 MiB L2s).  Block b reads region (b + shift) % 256 and records
// its elapsed time (s_memrealtime, 100 MHz) and the XCD it ran on.  Sequence, back to back on one stream (also captured in
// a graph): flush (512 MB write) | read shift 0 (cold) | read shift 0 (same block, same tile) | read shift 8 (a tile another
// block of the SAME XCD read, if blocks are dealt round-robin) | read shift 1 (a tile a block of another XCD read).
//   hipcc --offload-arch=gfx950 -O3 tools/l2_retention.hip -o tools/l2_retention && tools/l2_retention
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define NB 256
#define NL 16          // 16 x 16 B x 256 threads = 64 KB per block

__global__ __launch_bounds__(256) void reader(const f32x4* buf, int shift, unsigned long long* ticks, int* xcc, float* sink) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const f32x4* p = buf + (size_t)((blockIdx.x + shift) % NB) * (NL * 256) + threadIdx.x;
    f32x4 r[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) r[i] = p[(size_t)i * 256];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NL; ++i) s += r[i][0] + r[i][1] + r[i][2] + r[i][3];
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        ticks[blockIdx.x] = t1 - t0;
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[blockIdx.x] = (int)(id & 0xF);
    }
    if (s == 12345.678f) sink[threadIdx.x] = s;
}
__global__ void flusher(f32x4* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = f32x4{1.f, 2.f, 3.f, 4.f};
}
static double median_us(std::vector<unsigned long long> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2] / 100.0; }

int main() {
    f32x4 *buf, *big; unsigned long long* ticks; int* xcc; float* sink;
    const size_t nbuf = (size_t)NB * NL * 256, nbig = (size_t)512 << 20 >> 4;
    (void)hipMalloc(&buf, nbuf * 16); (void)hipMalloc(&big, nbig * 16);
    (void)hipMalloc(&ticks, 5 * NB * 8); (void)hipMalloc(&xcc, 5 * NB * 4); (void)hipMalloc(&sink, 1024 * 4);
    (void)hipMemset(buf, 0, nbuf * 16);
    hipStream_t s; (void)hipStreamCreate(&s);
    const int shifts[4] = {0, 0, 8, 1};
    const char* names[4] = {"cold (after a 512 MB flush)", "same block, same tile again", "tile read by another block of the same XCD (shift 8)",
                            "tile read by a block of another XCD (shift 1)"};
    for (int mode = 0; mode < 2; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        if (mode == 1) (void)hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed);
        hipLaunchKernelGGL(flusher, dim3(2048), dim3(256), 0, s, big, nbig);
        for (int k = 0; k < 4; ++k)
            hipLaunchKernelGGL(reader, dim3(NB), dim3(256), 0, s, buf, shifts[k], ticks + k * NB, xcc + k * NB, sink);
        if (mode == 1) { (void)hipStreamEndCapture(s, &g); (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0); (void)hipGraphLaunch(ge, s); (void)hipGraphLaunch(ge, s); }
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> t(4 * NB); std::vector<int> x(4 * NB);
        (void)hipMemcpy(t.data(), ticks, 4 * NB * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(x.data(), xcc, 4 * NB * 4, hipMemcpyDeviceToHost);
        printf("%s launches\n", mode ? "hipGraph" : "eager");
        for (int k = 0; k < 4; ++k)
            printf("  %-62s median %.2f us per block (64 KB)\n", names[k], median_us(std::vector<unsigned long long>(t.begin() + k * NB, t.begin() + (k + 1) * NB)));
        int same = 0, rr = 0;
        for (int b = 0; b < NB; ++b) { same += x[b] == x[NB + b] && x[b] == x[2 * NB + b]; rr += x[b] == x[(b + 8) % NB]; }
        printf("  blocks on the same XCD in three consecutive launches: %d / %d;  block b and b+8 on the same XCD: %d / %d;  XCD of blocks 0..9:", same, NB, rr, NB);
        for (int b = 0; b < 10; ++b) printf(" %d", x[b]);
        printf("\n");
    }
    return 0;
}
