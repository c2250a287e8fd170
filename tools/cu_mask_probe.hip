// Which XCDs do the workgroups of a CU-masked stream land on?  (hipExtStreamCreateWithCUMask; 256 CUs = 8 mask words.)  Prints, for a few
// mask patterns, the histogram of XCC_ID over 2048 workgroups -- the experiment behind running the GST branch on the XCDs the
// persistent BiLSTM leaves idle.   hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o tools/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(unsigned* hist) {
    if (threadIdx.x == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        atomicAdd(&hist[xcc & 15], 1u);
        for (int i = 0; i < 2000; ++i) __builtin_amdgcn_s_sleep(10);        // (hold the CU a little so that the grid spreads)
    }
}
int main() {
    unsigned* d;
    hipMalloc(&d, 64);
    const char* names[] = {"words 0..3 (CUs 0..127)", "words 4..7 (CUs 128..255)", "low half of every word", "high half of every word",
                           "bits with (i % 8) >= 4", "bits with (i / 8) % 2 == 1", "every CU"};
    for (int p = 0; p < 7; ++p) {
        uint32_t m[8];
        for (int w = 0; w < 8; ++w) {
            m[w] = 0;
            for (int b = 0; b < 32; ++b) {
                const int i = w * 32 + b;
                bool on = p == 0 ? w < 4 : p == 1 ? w >= 4 : p == 2 ? b < 16 : p == 3 ? b >= 16 : p == 4 ? (i % 8) >= 4 : p == 5 ? ((i / 8) % 2) == 1 : true;
                if (on) m[w] |= 1u << b;
            }
        }
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, m);
        if (e != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", names[p], hipGetErrorString(e)); continue; }
        hipMemsetAsync(d, 0, 64, s);
        hipLaunchKernelGGL(probe, dim3(2048), dim3(64), 0, s, d);
        unsigned h[16];
        hipMemcpyAsync(h, d, 64, hipMemcpyDeviceToHost, s);
        e = hipStreamSynchronize(s);
        printf("%-28s %s  XCC histogram:", names[p], hipGetErrorString(e));
        for (int x = 0; x < 8; ++x) printf(" %u", h[x]);
        printf("\n");
        hipStreamDestroy(s);
    }
    return 0;
}
