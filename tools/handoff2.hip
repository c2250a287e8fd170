// Micro-benchmark 5 (round 2): the producer -> consumer hand-off the merged "projection + front" decode launch needs.
//
// One launch per decode step (captured in a hipGraph, like the product): NP producer workgroups (the projection tiles)
// each compute for `work_us` and then publish their 16 x 16 block of z0 as 8-byte {value, tag} granules with write-through
// (sc1) stores; NC consumer workgroups (one per utterance) first issue a burst of weight loads (as the front end does), then
// poll THEIR 256 granules with sc1 loads until every tag equals the step number, and check the values.  The granule buffer
// is reused by every step (tag = step), so a stale copy anywhere would show as a wrong value or a spin that gives up.
// Reported: launch time per step, producer-store -> consumer-has-all latency (s_memrealtime stamps), wrong values, give-ups.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/handoff2 tools/handoff2.hip && tools/handoff2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Gran { float v; unsigned int tag; };

__device__ __forceinline__ void st_gran(Gran* p, float v, unsigned int tag) {
    const unsigned long long bits = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
    asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(bits) : "memory");
}
__device__ __forceinline__ unsigned long long ld_gran(const Gran* p) {
    unsigned long long bits;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(bits) : "v"(p) : "memory");
    return bits;
}

// rows = 32 utterances, 256 columns; producer p owns column tile (p % 16) of row tile (p / 16) for p < 32, the other
// producers (the frame tiles of the projection) publish nothing.
__global__ __launch_bounds__(1024) void k_step(Gran* z, const f32x4* weights, int NP, int NC, unsigned int step, int work_ticks,
                                               int burst16, unsigned long long* stamps, unsigned int* bad, unsigned int* gaveup,
                                               float* sink) {
    const int b = blockIdx.x;
    if (b < NP) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < work_ticks) __builtin_amdgcn_s_sleep(1);
        if (b < 32 && threadIdx.x < 256) {
            const int rt = b / 16, ct = b % 16;
            const int row = rt * 16 + (threadIdx.x >> 4), col = ct * 16 + (threadIdx.x & 15);
            st_gran(z + row * 256 + col, (float)(step * 1000 + row) + (float)col * 0.001f, step);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (threadIdx.x == 0) stamps[(size_t)step * 96 + b] = __builtin_amdgcn_s_memrealtime();     // store acknowledged
        return;
    }
    if (b >= NP + NC) return;
    const int u = b - NP;
    // the weight burst of the front end (W1 + Wq, half of the rows): burst16 16-byte loads per lane
    f32x4 acc = {0, 0, 0, 0};
    for (int i = 0; i < burst16; ++i) acc += weights[((size_t)i * 1024 + threadIdx.x + (size_t)u * 7) & 0xFFFFF];
    if (acc[0] == 1.2345f) sink[0] = acc[1];
    float val = 0.f;
    bool ok = true;
    if (threadIdx.x < 256) {
        unsigned int spins = 0;
        for (;;) {
            const unsigned long long g = ld_gran(z + u * 256 + threadIdx.x);
            if ((unsigned int)(g >> 32) == step) { val = __uint_as_float((unsigned int)g); break; }
            if (++spins > (1u << 18)) { ok = false; break; }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) stamps[(size_t)step * 96 + 64 + u] = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x < 256) {
        if (!ok) atomicAdd(gaveup, 1u);
        else if (val != (float)(step * 1000 + u) + (float)threadIdx.x * 0.001f) atomicAdd(bad, 1u);
    }
}

int main() {
    const int NP = 54, NC = 32, steps = 400;
    Gran* z; f32x4* w; unsigned long long* stamps; unsigned int *bad, *gaveup; float* sink;
    CK(hipMalloc(&z, 32 * 256 * sizeof(Gran))); CK(hipMalloc(&w, (size_t)(1 << 20) * 16)); CK(hipMalloc(&stamps, (size_t)steps * 96 * 8));
    CK(hipMalloc(&bad, 4)); CK(hipMalloc(&gaveup, 4)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(w, 0, (size_t)(1 << 20) * 16));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant = 0; variant < 4; ++variant) {
        const int work_ticks = variant < 2 ? 250 : 0;          // 2.5 us of "projection" before the publish (100 MHz ticks), or none
        const int burst16 = (variant & 1) ? 12 : 0;            // with / without the consumer's weight burst (12 x 16 KiB = 192 KiB per CU)
        CK(hipMemsetAsync(z, 0, 32 * 256 * sizeof(Gran), st)); CK(hipMemsetAsync(bad, 0, 4, st)); CK(hipMemsetAsync(gaveup, 0, 4, st));
        CK(hipMemsetAsync(stamps, 0, (size_t)steps * 96 * 8, st));
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        for (int s = 1; s <= steps - 1; ++s)
            hipLaunchKernelGGL(k_step, dim3(256), dim3(1024), 0, st, z, w, NP, NC, (unsigned int)s, work_ticks, burst16, stamps, bad, gaveup, sink);
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(z, 0, 32 * 256 * sizeof(Gran), st));       // tags of the previous replay must not match
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        std::vector<unsigned long long> h((size_t)steps * 96);
        CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        unsigned int hb, hg; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hg, gaveup, 4, hipMemcpyDeviceToHost));
        std::vector<double> lat;
        for (int s = 1; s <= steps - 1; ++s) {
            unsigned long long last_pub = 0, last_got = 0;
            for (int p = 0; p < 32; ++p) last_pub = std::max(last_pub, h[(size_t)s * 96 + p]);
            for (int u = 0; u < 32; ++u) last_got = std::max(last_got, h[(size_t)s * 96 + 64 + u]);
            lat.push_back(((double)last_got - (double)last_pub) / 100.0);
        }
        std::sort(lat.begin(), lat.end());
        printf("producers work %.1f us, consumer burst %3d KiB: %.2f us / launch | last publish -> last consumer has its row: median %.2f us, p90 %.2f, max %.2f | wrong %u, gave up %u\n",
               work_ticks / 100.0, burst16 * 16, ms * 1e3 / (steps - 1), lat[lat.size() / 2], lat[lat.size() * 9 / 10], lat.back(), hb, hg);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
