// Micro-benchmark 3: cost of an in-kernel grid-wide hand-off on gfx950 (256 workgroups, one per CU) versus the
// 1.5-1.7 us graph kernel boundary: (a) bare barrier, (b) with release/acquire fences, (c) with every workgroup
// publishing 512 B and then reading all 128 KiB (the h[32,1024] broadcast of an LSTM phase), verified.
// Spins are bounded: a barrier that does not complete sets an error flag and every later barrier falls through.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct Sync { unsigned int count; unsigned int error; };

__device__ __forceinline__ void grid_barrier(Sync* s, unsigned int target, bool fences) {
    __syncthreads();
    if (threadIdx.x == 0) {
        if (fences) __atomic_thread_fence(__ATOMIC_RELEASE);      // agent scope by default for HIP device code? use builtin below
        if (fences) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(&s->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned int spins = 0;
        while (__hip_atomic_load(&s->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 22) || __hip_atomic_load(&s->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(&s->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        if (fences) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// variant B: no sleep while polling, two-level arrive (per-XCD counter, last arriver of an XCD bumps the global one),
// data published with agent-scope atomic stores and read with agent-scope atomic loads (no L2 writeback/invalidate)
struct Sync2 { unsigned int xcd[8 * 32]; unsigned int top; unsigned int error; };   // counters 128 B apart
__device__ __forceinline__ void grid_barrier2(Sync2* s, unsigned int it, int nwg, int two_level) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_s_waitcnt(0);      // all of this wave's stores issued and acknowledged (vmcnt/lgkmcnt = 0)
        if (two_level) {
            const int x = blockIdx.x & 7, per = nwg >> 3;
            const unsigned int old = __hip_atomic_fetch_add(&s->xcd[x * 32], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (old + 1 == (it + 1) * per) __hip_atomic_fetch_add(&s->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_fetch_add(&s->top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned int target = two_level ? (it + 1) * 8 : (it + 1) * nwg;
        unsigned int spins = 0;
        while (__hip_atomic_load(&s->top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > (1u << 24)) { __hip_atomic_store(&s->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(512) void k_persist2(Sync2* s, float* buf, float* out, int iters, int two_level, int pub, int rd, unsigned int* bad) {
    const int nwg = gridDim.x;
    float acc = 0.f;
    unsigned int mism = 0;
    for (int it = 0; it < iters; ++it) {
        float* cur = buf + (size_t)(it & 1) * nwg * pub;
        for (int i = threadIdx.x; i < pub; i += blockDim.x)
            __hip_atomic_store(&cur[(size_t)blockIdx.x * pub + i], (float)(it * 7 + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        grid_barrier2(s, (unsigned int)it, nwg, two_level);
        for (int i = threadIdx.x; i < rd; i += blockDim.x) {
            const float v = __hip_atomic_load(&cur[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            mism += v != (float)(it * 7 + i / pub);
            acc += v;
        }
    }
    if (mism) atomicAdd(bad, mism);
    if (acc == 1.2345f) out[0] = acc;
}

// mode 0: bare barrier; 1: + fences; 2: + publish `pub` floats per WG and read `rd` floats of the shared buffer
__global__ __launch_bounds__(512) void k_persist(Sync* s, float* buf, float* out, int iters, int mode, int pub, int rd, unsigned int* bad) {
    const int nwg = gridDim.x;
    float acc = 0.f;
    unsigned int mism = 0;
    for (int it = 0; it < iters; ++it) {
        float* cur = buf + (size_t)(it & 1) * nwg * pub;           // ping-pong so a fast WG cannot overwrite what a slow one still reads
        if (mode >= 2)
            for (int i = threadIdx.x; i < pub; i += blockDim.x) cur[(size_t)blockIdx.x * pub + i] = (float)(it * 7 + blockIdx.x);
        grid_barrier(s, (unsigned int)(it + 1) * nwg, mode >= 1);
        if (mode >= 2) {
            const float4* c4 = reinterpret_cast<const float4*>(cur);
            for (int i = threadIdx.x; i < rd / 4; i += blockDim.x) {
                const float4 v = c4[i];
                const int wg = (i * 4) / pub;
                const float e = (float)(it * 7 + wg);
                mism += (v.x != e) + (v.y != e) + (v.z != e) + (v.w != e);
                acc += v.x + v.y + v.z + v.w;
            }
        }
    }
    if (mism) atomicAdd(bad, mism);
    if (acc == 1.2345f) out[0] = acc;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int nwg = prop.multiProcessorCount;
    printf("CUs %d\n", nwg);
    Sync* s; float *buf, *out; unsigned int* bad;
    CK(hipMalloc(&s, sizeof(Sync))); CK(hipMalloc(&buf, 8 << 20)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&bad, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    struct Case { const char* name; int mode, pub, rd, threads; } cases[] = {
        {"bare barrier                        ", 0, 0, 0, 512},
        {"barrier + release/acquire fences    ", 1, 0, 0, 512},
        {"publish 512 B, read 128 KiB (512 thr)", 2, 128, 128 * 256, 512},
        {"publish 512 B, read 128 KiB (256 thr)", 2, 128, 128 * 256, 256},
        {"publish 512 B, read  48 KiB (512 thr)", 2, 128, 48 * 256, 512},
        {"publish 512 B, read  16 KiB (512 thr)", 2, 128, 16 * 256, 512},
        {"publish 2 KiB, read 512 KiB (512 thr)", 2, 512, 512 * 256, 512},
    };
    for (const Case& c : cases) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(s, 0, sizeof(Sync))); CK(hipMemset(bad, 0, 4));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_persist, dim3(nwg), dim3(c.threads), 0, 0, s, buf, out, iters, c.mode, c.pub, c.rd, bad);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            Sync h; unsigned int hb; CK(hipMemcpy(&h, s, sizeof(h), hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            if (rep == 1) printf("%s: %.2f us / hand-off  (error flag %u, stale reads %u)\n", c.name, ms * 1e3 / iters, h.error, hb);
        }
    }
    Sync2* s2; CK(hipMalloc(&s2, sizeof(Sync2)));
    struct Case2 { const char* name; int two, pub, rd; } cases2[] = {
        {"B flat, no sleep, bare                 ", 0, 0, 0},
        {"B two-level, bare                      ", 1, 0, 0},
        {"B two-level, sc1 publish 512 B/read 16K", 1, 128, 16 * 256},
        {"B two-level, sc1 publish 512 B/read 128K", 1, 128, 128 * 256},
        {"B flat,      sc1 publish 512 B/read 128K", 0, 128, 128 * 256},
    };
    for (const Case2& c : cases2) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(s2, 0, sizeof(Sync2))); CK(hipMemset(bad, 0, 4));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_persist2, dim3(nwg), dim3(512), 0, 0, s2, buf, out, iters, c.two, c.pub, c.rd, bad);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            Sync2 h; unsigned int hb; CK(hipMemcpy(&h, s2, sizeof(h), hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            if (rep == 1) printf("%s: %.2f us / hand-off  (error flag %u, stale reads %u)\n", c.name, ms * 1e3 / iters, h.error, hb);
        }
    }
    return 0;
}
