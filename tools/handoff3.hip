// Micro-benchmark 6 (round 2): an all-to-all hand-off among workgroups of ONE XCD, for a persistent BiLSTM direction.
//
// tools/handoff.hip / handoff2.hip priced hand-offs across the chip: consumers on other XCDs need agent-scope (sc1) loads,
// which are served from the fabric, not from their L2 (1.8 TB/s over all CUs; a 32 KiB state read by 128 workgroups = 2 us).
// Workgroups are dealt to the 8 XCDs round-robin by id, so the 32 workgroups with blockIdx % 8 == x share ONE L2, which is
// coherent for them: plain (write-through L1) stores land in it and a load only has to miss the reader's own L1.
// Here: grid of 256, the 32 workgroups of XCD `x` exchange a 32 KiB state per iteration (each publishes 1 KiB, waits for the
// 32 flags, reads all 32 KiB), two parity buffers RE-USED every other iteration (so stale L1 / L2 lines would show), with the
// load / store flavours below.  Reports us / iteration, wrong words, wait give-ups and the XCC ids the participants saw.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/handoff3 tools/handoff3.hip && tools/handoff3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int F> __device__ __forceinline__ f32x4 ld16(const f32x4* p) {
    f32x4 v;
    if (F == 0) asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 1) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 4) asm volatile("global_load_dwordx4 %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int F> __device__ __forceinline__ unsigned ld4(const unsigned* p) {
    unsigned v;
    if (F == 0) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    if (F == 4) asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int S> __device__ __forceinline__ void st4f(float* p, float v) {
    if (S == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (S == 1) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (S == 2) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
template <int S> __device__ __forceinline__ void st4u(unsigned* p, unsigned v) {
    if (S == 0) asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
    if (S == 1) asm volatile("global_store_dword %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
    if (S == 2) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// state: [2 parities][8192 floats]; flags: [2 parities][32]
template <int F, int S>
__global__ __launch_bounds__(512) void k_xcd(float* state, unsigned* flags, int iters, int xcd, int nwg, unsigned* bad, unsigned* gaveup,
                                             unsigned* xcc_seen, float* sink) {
    if ((int)(blockIdx.x & 7) != xcd) return;
    const int rank = blockIdx.x >> 3;
    if (rank >= nwg) return;
    const int tid = threadIdx.x, lane = tid & 63;
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        atomicOr(xcc_seen, 1u << (id & 15));
    }
    unsigned mism = 0;
    float acc = 0.f;
    const int per = 8192 / nwg;                                  // floats each workgroup publishes
    for (int it = 0; it < iters; ++it) {
        const int par = it & 1;
        float* st = state + par * 8192;
        if (tid < per) st4f<S>(st + rank * per + tid, (float)(it + 1) + (float)(rank * per + tid) * (1.f / 16384.f));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) st4u<S>(flags + par * 32 + rank, (unsigned)(it + 1));
        unsigned spins = 0;
        for (;;) {
            const unsigned v = lane < nwg ? ld4<F>(flags + par * 32 + lane) : 0xffffffffu;
            if (__builtin_amdgcn_readfirstlane(__popcll(__ballot(v >= (unsigned)(it + 1)))) == 64) break;
            if (++spins > (1u << 13)) { if (tid == 0) atomicAdd(gaveup, 1u); break; }
        }
        // 8192 floats = 2048 16-byte pieces: 4 per thread
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + j * 512;
            const f32x4 v = ld16<F>(reinterpret_cast<const f32x4*>(st) + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) mism += v[e] != (float)(it + 1) + (float)(i * 4 + e) * (1.f / 16384.f);
            acc += v[0];
        }
        __syncthreads();
    }
    if (mism) atomicAdd(bad, mism);
    if (acc == 1.2345f) sink[0] = acc;
}

template <int F, int S> void run(const char* name, int nwg) {
    float* state; unsigned *flags, *bad, *gaveup, *xcc; float* sink;
    CK(hipMalloc(&state, 2 * 8192 * 4)); CK(hipMalloc(&flags, 2 * 32 * 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&gaveup, 4)); CK(hipMalloc(&xcc, 4));
    CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int xcd = 0; xcd < 8; xcd += 3) {
        float ms = 0;
        unsigned hb = 0, hg = 0, hx = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(state, 0, 2 * 8192 * 4)); CK(hipMemset(flags, 0, 2 * 32 * 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(gaveup, 0, 4)); CK(hipMemset(xcc, 0, 4));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((k_xcd<F, S>), dim3(256), dim3(512), 0, 0, state, flags, iters, xcd, nwg, bad, gaveup, xcc, sink);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hg, gaveup, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hx, xcc, 4, hipMemcpyDeviceToHost));
        }
        printf("%-34s %2d workgroups with blockIdx %% 8 == %d: %.2f us / hand-off | wrong words %u, gave up %u, XCC ids seen mask 0x%x\n", name, nwg, xcd,
               ms * 1e3 / iters, hb, hg, hx);
        fflush(stdout);
    }
    CK(hipFree(state)); CK(hipFree(flags)); CK(hipFree(bad)); CK(hipFree(gaveup)); CK(hipFree(xcc)); CK(hipFree(sink));
}

int main() {
    for (int nwg : {32, 16}) {
        run<2, 2>("loads sc1, stores sc1", nwg);
        run<2, 0>("loads sc1, stores plain", nwg);
        run<1, 0>("loads sc0, stores plain", nwg);
        run<1, 1>("loads sc0, stores sc0", nwg);
        run<3, 0>("loads sc0 sc1, stores plain", nwg);
        run<4, 0>("loads nt, stores plain", nwg);
        run<0, 0>("loads plain, stores plain", nwg);
    }
    return 0;
}
