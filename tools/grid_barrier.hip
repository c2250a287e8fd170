// Micro-benchmark 3: cost of an in-kernel grid-wide hand-off on gfx950 (256 workgroups, one per CU) versus the
// 1.5-1.7 us graph kernel boundary: (a) bare barrier, (b) with release/acquire fences, (c) with every workgroup
// publishing 512 B and then reading all 128 KiB (the h[32,1024] broadcast of an LSTM phase), verified.
// Spins are bounded: a barrier that does not complete sets an error flag and every later barrier falls through.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4_t __attribute__((ext_vector_type(4)));
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

// variant C: only the workgroups with blockIdx % 8 == 0 take part (32 of 256): if workgroups are dealt round-robin to
// the 8 XCDs they share ONE L2, so the counter can be a plain L2 atomic (workgroup scope) and the exchanged data needs
// no write-back / invalidate of L2 -- only the reader's L1 has to be bypassed.  xcc[] records HW_REG_XCC_ID per WG.
struct Sync3 { unsigned int count; unsigned int error; unsigned int xcc[64]; };
__global__ __launch_bounds__(256) void k_persist3(Sync3* s, float* buf, float* out, int iters, int scope_agent, int read_mode, int pub, unsigned int* bad) {
    if (blockIdx.x & 7) return;
    const int me = blockIdx.x >> 3, n = gridDim.x >> 3;
    if (threadIdx.x == 0) s->xcc[me] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 0xf;   // XCC_ID[3:0]
    float acc = 0.f;
    unsigned int mism = 0;
    for (int it = 0; it < iters; ++it) {
        float* cur = buf + (size_t)(it & 1) * n * pub;
        for (int i = threadIdx.x; i < pub; i += blockDim.x) {
            const float v = (float)(it * 7 + me);
            if (read_mode == 0) __hip_atomic_store(&cur[(size_t)me * pub + i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else cur[(size_t)me * pub + i] = v;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_s_waitcnt(0);
            const unsigned int target = (unsigned int)(it + 1) * n;
            if (scope_agent) __hip_atomic_fetch_add(&s->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(&s->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            unsigned int spins = 0;
            while ((scope_agent ? __hip_atomic_load(&s->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                : __hip_atomic_load(&s->count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < target) {
                if (++spins > (1u << 18) || __hip_atomic_load(&s->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(&s->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            }
        }
        __syncthreads();
        if (__hip_atomic_load(&s->error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        const int total = n * pub;
        if (read_mode == 0) {
            for (int i = threadIdx.x; i < total; i += blockDim.x) {
                const float v = __hip_atomic_load(&cur[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mism += v != (float)(it * 7 + i / pub); acc += v;
            }
        } else {
            if (read_mode == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            if (read_mode == 3) asm volatile("buffer_inv sc0" ::: "memory");
            const float4* c4 = reinterpret_cast<const float4*>(cur);
            for (int i = threadIdx.x; i < total / 4; i += blockDim.x) {
                float4 v;
                if (read_mode == 4) { const f32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(c4 + i)); v = make_float4(t[0], t[1], t[2], t[3]); }
                else v = c4[i];
                const float e = (float)(it * 7 + (i * 4) / pub);
                mism += (v.x != e) + (v.y != e) + (v.z != e) + (v.w != e); acc += v.x + v.y + v.z + v.w;
            }
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
    Sync3* s3; CK(hipMalloc(&s3, sizeof(Sync3)));
    struct Case3 { const char* name; int agent, read_mode, pub; } cases3[] = {
        {"C 32 WGs one XCD, agent atomics, bare        ", 1, 1, 0},
        {"C 32 WGs one XCD, L2 atomics, bare           ", 0, 1, 0},
        {"C L2 atomics, sc1 publish 1K / sc1 read 32K  ", 0, 0, 256},
        {"C L2 atomics, plain publish / plain read 32K ", 0, 1, 256},
        {"C L2 atomics, plain publish / wg-acq read 32K", 0, 2, 256},
        {"C L2 atomics, plain publish / inv sc0 read   ", 0, 3, 256},
        {"C L2 atomics, plain publish / nt read 32K    ", 0, 4, 256},
        {"C agent atomics, sc1 publish / sc1 read 32K  ", 1, 0, 256},
    };
    for (const Case3& c : cases3) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipMemset(s3, 0, sizeof(Sync3))); CK(hipMemset(bad, 0, 4));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k_persist3, dim3(nwg), dim3(256), 0, 0, s3, buf, out, iters, c.agent, c.read_mode, c.pub, bad);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            Sync3 h; unsigned int hb; CK(hipMemcpy(&h, s3, sizeof(h), hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            unsigned int xmask = 0; for (int i = 0; i < nwg / 8; ++i) xmask |= 1u << h.xcc[i];
            if (rep == 1) { printf("%s: %.2f us / hand-off  (error %u, stale reads %u, XCC mask 0x%x)\n", c.name, ms * 1e3 / iters, h.error, hb, xmask); fflush(stdout); }
        }
    }
    return 0;
}
