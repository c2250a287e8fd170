// Micro-benchmark 4 (round 2): an in-kernel all-to-all hand-off with FRESH ADDRESSES per iteration, against the same
// exchange done with dependent kernel launches in a hipGraph.
//
// tools/grid_barrier.hip (round 1) priced hand-offs that re-use two ping-pong buffers: the readers then need agent-scope
// (sc1) loads or an acquire fence to get past their stale L1 / L2 lines, and a 128 KiB read cost 16-20 us.  Here every
// iteration publishes into a slot that no CU has touched since the kernel started, so no cache can hold a stale copy of
// it: producers store write-through (sc1, 16 B per lane), drain (vmcnt(0)), one lane per workgroup adds to a per-XCD-
// sharded counter; consumers poll the 8 shards with sc1 loads and then read the slot with PLAIN 16-byte loads (the first
// CU of an XCD to touch a line pulls it from memory into that XCD's L2, the other 31 hit L2).
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/handoff tools/handoff.hip && tools/handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Ctl {
    unsigned int shard[8 * 32];     // per-XCD arrival counters, 128 B apart
    unsigned int error;
};

__device__ __forceinline__ void st_sc1(f32x4* p, f32x4 v) {
    // write-through store: the line leaves this XCD's L2 (MI355X_MICROARCH.md, store flavours)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned int ld_sc1(const unsigned int* p) {
    unsigned int v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// One hand-off: every workgroup has published; returns when all `nwg` workgroups of iteration `it` have arrived.
__device__ __forceinline__ void arrive_and_wait(Ctl* c, unsigned int it, int nwg, int xcd) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's write-through stores have been acknowledged
    __syncthreads();                                        // ... and every other wave's of this workgroup
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&c->shard[xcd * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int target = (it + 1) * (unsigned int)(nwg >> 3);
        const int lane = threadIdx.x & 7;                   // lanes 0..7 watch one shard each
        unsigned int spins = 0;
        for (;;) {
            const unsigned int v = ld_sc1(&c->shard[lane * 32]);
            const bool ok = v >= target;
            if (__builtin_amdgcn_readfirstlane(__popcll(__ballot(ok || threadIdx.x >= 8))) == 64) break;
            if (++spins > (1u << 20)) { if (threadIdx.x == 0) __hip_atomic_store(&c->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
}

// pub16: 16-byte pieces each workgroup publishes per iteration; rd16: 16-byte pieces every workgroup then reads (from
// the start of the slot).  The slot of iteration `it` is slots + it * slot16 (fresh every iteration).
__global__ __launch_bounds__(512) void k_fresh(Ctl* c, f32x4* slots, size_t slot16, int iters, int pub16, int rd16, int plain_read,
                                               unsigned int* bad, float* sink) {
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7;
    float acc = 0.f;
    unsigned int mism = 0;
    for (int it = 0; it < iters; ++it) {
        f32x4* slot = slots + (size_t)it * slot16;
        const float tag = (float)(it * 3 + 1);
        for (int i = threadIdx.x; i < pub16; i += blockDim.x) {
            const float e = tag + (float)blockIdx.x * 0.001f;
            st_sc1(slot + (size_t)blockIdx.x * pub16 + i, f32x4{e, e, e, e});
        }
        arrive_and_wait(c, (unsigned int)it, nwg, xcd);
        // 8 loads in flight per lane
        for (int i0 = threadIdx.x; i0 < rd16; i0 += blockDim.x * 8) {
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * blockDim.x;
                v[j] = f32x4{0, 0, 0, 0};
                if (i < rd16) {
                    if (plain_read) v[j] = slot[i];
                    else v[j] = __builtin_nontemporal_load(slot + i);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * blockDim.x;
                if (i < rd16) {
                    const float e = tag + (float)(i / pub16) * 0.001f;
                    mism += (v[j][0] != e) + (v[j][1] != e) + (v[j][2] != e) + (v[j][3] != e);
                    acc += v[j][0];
                }
            }
        }
    }
    if (mism) atomicAdd(bad, mism);
    if (acc == 1.2345f) sink[0] = acc;
}

// The same exchange with a kernel boundary per iteration (captured into one hipGraph).
__global__ __launch_bounds__(512) void k_step(f32x4* slot_out, const f32x4* slot_in, int pub16, int rd16, float tag_out, float tag_in,
                                              unsigned int* bad, float* sink) {
    float acc = 0.f;
    unsigned int mism = 0;
    if (slot_in) {
        for (int i0 = threadIdx.x; i0 < rd16; i0 += blockDim.x * 8) {
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * blockDim.x;
                v[j] = f32x4{0, 0, 0, 0};
                if (i < rd16) v[j] = slot_in[i];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * blockDim.x;
                if (i < rd16) {
                    const float e = tag_in + (float)(i / pub16) * 0.001f;
                    mism += (v[j][0] != e) + (v[j][1] != e) + (v[j][2] != e) + (v[j][3] != e);
                    acc += v[j][0];
                }
            }
        }
    }
    for (int i = threadIdx.x; i < pub16; i += blockDim.x) {
        const float e = tag_out + (float)blockIdx.x * 0.001f;
        slot_out[(size_t)blockIdx.x * pub16 + i] = f32x4{e, e, e, e};
    }
    if (mism) atomicAdd(bad, mism);
    if (acc == 1.2345f) sink[0] = acc;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int nwg = prop.multiProcessorCount;
    printf("CUs %d\n", nwg);
    const int iters = 400;
    Ctl* ctl; unsigned int* bad; float* sink; f32x4* slots;
    const size_t slot16 = (size_t)nwg * 512;                  // room for up to 8 KiB per workgroup per iteration
    CK(hipMalloc(&ctl, sizeof(Ctl))); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&slots, slot16 * 16 * iters));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipStream_t st; CK(hipStreamCreate(&st));
    struct Case { const char* name; int pub16, rd_kib, plain; } cases[] = {
        {"in-kernel, publish  16 B / read    0   ", 1, 0, 1},
        {"in-kernel, publish 512 B / read   16 KiB", 32, 16, 1},
        {"in-kernel, publish 512 B / read   48 KiB", 32, 48, 1},
        {"in-kernel, publish 512 B / read  128 KiB", 32, 128, 1},
        {"in-kernel, publish 512 B / read  128 KiB nt", 32, 128, 0},
        {"in-kernel, publish 2 KiB / read  512 KiB", 128, 512, 1},
    };
    for (const Case& c : cases) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(ctl, 0, sizeof(Ctl), st)); CK(hipMemsetAsync(bad, 0, 4, st));
            CK(hipMemsetAsync(slots, 0xff, slot16 * 16 * iters, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_fresh, dim3(nwg), dim3(512), 0, st, ctl, slots, slot16, iters, c.pub16, c.rd_kib * 64, c.plain, bad, sink);
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            Ctl h; unsigned int hb; CK(hipMemcpy(&h, ctl, sizeof(h), hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            if (rep == 2) printf("%s: %.2f us / hand-off  (error flag %u, wrong words %u)\n", c.name, ms * 1e3 / iters, h.error, hb);
            fflush(stdout);
        }
    }
    // kernel-boundary version of the same exchange
    for (const Case& c : cases) {
        if (!c.plain) continue;
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipMemsetAsync(bad, 0, 4, st));
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        for (int it = 0; it < iters; ++it) {
            f32x4* out = slots + (size_t)it * slot16;
            const f32x4* in = it ? slots + (size_t)(it - 1) * slot16 : nullptr;
            hipLaunchKernelGGL(k_step, dim3(nwg), dim3(512), 0, st, out, in, c.pub16, c.rd_kib * 64, (float)(it * 3 + 1), (float)((it - 1) * 3 + 1), bad, sink);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st));
            CK(hipGraphLaunch(ge, st));
            CK(hipEventRecord(e1, st));
            CK(hipStreamSynchronize(st));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        unsigned int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("graph launches, same exchange (%s): %.2f us / step  (wrong words %u)\n", c.name + 11, ms * 1e3 / iters, hb);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
