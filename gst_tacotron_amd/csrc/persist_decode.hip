// The whole decoder loop (reference Taco2.py:153-228: Max_Step // r iterations of Decoder_Step, Taco2.py:96-120, with the
// monotonic attention of Steps.py:107-229) as ONE PERSISTENT LAUNCH -- batches of up to 32 utterances of up to 128 tokens in
// fp32 at the reference's decoder sizes (prenet 256/256, attention 128, LSTM 1024/1024), i.e. the headline shape.
//
// Why (tools/persist_step.hip, profiles/r04_persist_step.txt, EXPERIMENTS round 4): the three-launch step costs 26.0 us, of
// which 6.6 us are kernel boundaries, ~4 us the weights' trip from the Infinity Cache behind each boundary and 9 us a
// per-utterance chain that starts from nothing every step.  Here 256 workgroups (one per CU, 512 threads) stay resident for
// all steps:
//   * EVERY workgroup owns gate tile `blockIdx.x` (4 hidden units x 4 gates) of BOTH LSTM cells, weights in registers;
//   * workgroups 0..31 also run utterance b's chain -- prenet-1 + dropout, query, scores, SMA / BMA alignment, context -- with
//     the utterance's processed memory in LDS for the whole loop and ALL of prenet-1's weights requested while the previous
//     step's projection is still running (rows the hashed dropout zeroes are never requested);
//   * the next workgroups also own a (projection tile, 16-row M-tile) with its 72 KB in registers;
//   * the chain workgroups' LSTM tiles still need their recurrent halves h . W_h + b (3.4 us of MFMA per step that would sit
//     between the chain and the cells): 64 HELPER workgroups compute them from the state fragments they hold anyway and hand
//     them back through memory.
// Hand-offs (all bounded; a give-up raises the host-mapped word the fused LSTM launch uses, gsttaco_synchronize reports it, the
// context falls back to launches):
//   S1  projection -> chains   prenet-0 pre-activations as 8-byte {value, step tag} granules, one row per utterance (1-to-1)
//   S2  chains -> everybody    prenet output (early: the LSTM-1 workgroups multiply it under the attention) and context,
//                              16-byte write-through stores, one flag per utterance
//   S3 / S4  h1 / h2 all-to-all  write-through stores, drained, 8-way sharded arrival counter, one sc1 poll in flight per
//                              workgroup (pipelined polls made it slower: the arrivals queue behind them), sc1 loads
//
// ARITHMETIC: bitwise the launch path's (gt_dec_front[_lean]_kernel, gt_lstm12_kernel, gt_proj_lean_kernel and their workers):
// the same k-block -> wave assignment and summation orders everywhere; where the launch path runs 1024 threads (the chain,
// the recurrent-half workers) every thread here plays two of them with separate accumulators.  tests/test_gpu_parity.py
// compares the two paths bitwise over hundreds of steps -- which is also the test that no hand-off ever delivers a stale word.
#include "chain_common.h"
#include "device_utils.h"
#include "kernels.h"
#include "../../include/gsttaco.h"

namespace {

constexpr int PD_NT = 512, PD_NW = 8, PD_NWG = 256, PD_UTT = 32, PD_HELP = 64;
constexpr int PD_P = 256, PD_A = 128, PD_H = 1024, PD_TVMAX = 256, PD_LDV = PD_A + 4;
constexpr int PD_GMAX = 4, PD_BMAX = 32 * PD_GMAX;     // groups of up to 32 rows through one set of resident weights (the _g kernels below)
constexpr int PD_KBP = PD_P / 16, PD_KBC = PD_A / 16, PD_KBH = PD_H / 16, PD_KBPJ = PD_KBH + PD_KBC;
constexpr uint32_t PD_SPIN_MAX = 1u << 20;

// control words (zeroed before every launch): a 128-byte line per counter shard and per chain-tile flag, two utterance flags per line
#ifndef PD_FS
#define PD_FS 16       // words between the per-utterance prenet / context flags.  Same-box A/B (profiles/r04_ab.txt): 32 (a line each, 32
                       // requests per poll) 20.5-20.7 us per step, 16 (two per line) 20.4, 8: 20.9-21.2, 1 (all 32 in ONE line, a poll is one
                       // request): 24.2 -- 32 writers and 256 pollers on one line queue at its memory channel
#endif
#ifndef PD_SLEEP_N
#define PD_SLEEP_N 0   // s_sleep between two polls of a flag / counter wait: 0 (none) 19.5-19.6 us per step, 1 (64 clocks): 20.3, 4: 20.3-20.4 (same-box
                       // A/B, profiles/r04_ab.txt) -- a poll is one read in flight and a round trip long: the pause only delays the next one
#endif
#define PD_SLEEP() do { if (PD_SLEEP_N > 0) __builtin_amdgcn_s_sleep(PD_SLEEP_N); } while (0)
// `zt`: a per-step opaque zero added into every per-thread global address of a body.  Without it the step loop's invariant
// addresses -- flags, counter shards and state rows of every group and parity: dozens of 64-bit pairs at four groups -- are hoisted
// out of the loop and live across the chain, which needs the whole register file: they spill, and a scratch reload waits for vmcnt(0).
#define PD_ZT(zt) int zt = 0; asm volatile("" : "+v"(zt))
#ifndef PD_X2_EARLY
#define PD_X2_EARLY 1   // the chain role requests W2x inside cell 1 (0: behind the h1 wait, with the h1 fragments: same-box A/B 19.8 against 19.4 us per step)
#endif
#ifndef PD_NSH
#define PD_NSH 64      // shards of an arrival counter (a 128-byte line each): arrivals per line = 256 / PD_NSH, lines per poll = PD_NSH.  The 256
                       // arrival atomics of an all-to-all are served one after the other per line: same-box A/B (profiles/r04_ab.txt) 8 shards
                       // 20.7-21.1 us per step, 16: 20.4-20.9, 32: 20.0, 64: 19.9-20.3, 128: 19.6-19.9, 256: 19.9-20.1
#endif
// (per utterance: a prenet flag and a context flag; per group of rows: the two arrival counters; per chain tile: two helper flags)
constexpr int PD_F_P = 0, PD_F_C = PD_BMAX * 32, PD_CNT3 = 2 * PD_BMAX * 32, PD_CNT4 = PD_CNT3 + PD_GMAX * PD_NSH * 32,
              PD_F_H = PD_CNT4 + PD_GMAX * PD_NSH * 32, PD_CTL_WORDS = PD_F_H + 128 * 32;     // (helper flags: 2 x 32 tiles, bf16 kernel 2 x 64)

__device__ __forceinline__ uint32_t pd_ld_sc1(const uint32_t* p) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ uint2 pd_ld2_sc1(const uint2* p) {
    uint2 v;
    asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ void pd_st4_sc1(float* p, float4 v) {
    f32x4 t = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(t) : "memory");
}
__device__ __forceinline__ void pd_st2_sc1(uint2* p, uint2 v) { asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void pd_st1_sc1(uint32_t* p, uint32_t v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void pd_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

#define PD_PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
// the accumulator is "used" here: the FMAs that feed it cannot sink below what follows (a sched_barrier orders the machine
// scheduler, not the IR passes before it)
#define PD_PIN4(a) do { asm volatile("" : "+v"((a).x), "+v"((a).y), "+v"((a).z), "+v"((a).w) : : "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

struct PdShared { int abort; };
// After a give-up (a bounded wait ran out, or found the launch's error word raised) the workgroup leaves at the top of its NEXT step:
// one LDS read of the flag per step instead of one behind every wait -- each was a read, a wait for it and a branch, ~60 ns, four or
// five of them on the step's critical path.  Until then the remaining phases of the step run on whatever the failed wait left
// (no address and no loop bound depends on data), and each of their waits gives up on the error word after 64 polls.
#define PD_PHASE_ABORT(sh) do { } while (0)
// a pointer argument into scalar registers NOW: placed in front of a wait, for what is dereferenced right behind it -- kernel
// arguments the compiler does not keep in registers are otherwise re-read (s_load + wait, ~0.1-0.2 us) between the hand-off and the
// first fragment request
#define PD_HOLD(p)                                                                                      \
    do {                                                                                                \
        const uint64_t u_ = reinterpret_cast<uint64_t>(p);                                              \
        uint32_t lo_ = __builtin_amdgcn_readfirstlane((uint32_t)u_), hi_ = __builtin_amdgcn_readfirstlane((uint32_t)(u_ >> 32)); \
        asm volatile("" : "+s"(lo_), "+s"(hi_));                                                        \
        p = reinterpret_cast<decltype(p)>(((uint64_t)hi_ << 32) | lo_);                                 \
    } while (0)

// diagnostic phase stamps (GSTTACO_STAMPS=1, tools/stamps_persist.py): thread 0 of workgroups 0 (chain), 32 (projection) and 255 (plain) at
// the middle step, slot = role * 32 + index (100 MHz ticks)
#define PD_STAMP(role, slot)                                                                                              \
    do {                                                                                                                  \
        if (A.dbg && t == (A.steps >> 1) && threadIdx.x == 0 && (blockIdx.x == 0 || (int)blockIdx.x == A.n_chain || blockIdx.x == PD_NWG - 1)) \
            A.dbg[(role) * 32 + (slot)] = __builtin_amdgcn_s_memrealtime();                                                 \
    } while (0)

// ---- bounded waits: one wave polls (ONE read in flight), the workgroup joins behind a barrier
__device__ __forceinline__ void pd_give_up(const PersistDecodeArgs& A, PdShared* sh, bool raise) {
    if (raise) atomicOr(A.err, 1u);         // (this launch's OWN give-up word: the fused LSTM launches poll theirs on any bit)
    sh->abort = 1;
}
// flags [n] (a line each) all >= want
template <int STRIDE = 32>
__device__ __forceinline__ void pd_wait_flags(const PersistDecodeArgs& A, const uint32_t* f, int n, uint32_t want, PdShared* sh) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        uint32_t spins = 0;
        for (;;) {
            const uint32_t v = lane < n ? pd_ld_sc1(f + lane * STRIDE) : want;
            if (__builtin_amdgcn_readfirstlane(__popcll(__ballot(v >= want))) == 64) break;
            if (++spins > PD_SPIN_MAX) { if (lane == 0) pd_give_up(A, sh, true); break; }
            if ((spins & 63u) == 0u && __builtin_amdgcn_readfirstlane(pd_ld_sc1(A.err)) != 0u) { if (lane == 0) pd_give_up(A, sh, false); break; }
            PD_SLEEP();
        }
    }
    __syncthreads();
}
// every one of the PD_NSH counter shards >= want.  Every workgroup arrives once per step on shard blockIdx.x % PD_NSH, so each shard's
// expected value is known (PD_WANT) and no cross-lane sum is needed: the poll is a load, a compare and a ballot (the sum was six
// dependent ds_bpermute round trips, ~0.2 us, inside every poll of every hand-off).
static_assert(PD_NWG % PD_NSH == 0 && PD_NSH % 64 == 0, "whole workgroups per shard, whole shards per lane");
#define PD_WANT(A, t) ((uint32_t)((t) + 1) * (uint32_t)(PD_NWG / PD_NSH) + (uint32_t)(A).expect_extra)
// `flag` (or NULL): one more word that must show `fwant` -- it rides in the same poll (a poll of its own in front of this one was a
// serial round trip of ~0.8 us even when long satisfied)
__device__ __forceinline__ void pd_wait_count(const PersistDecodeArgs& A, const uint32_t* c, uint32_t want, PdShared* sh, const uint32_t* flag = nullptr,
                                              uint32_t fwant = 0u) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        uint32_t spins = 0;
        for (;;) {
            uint32_t u[PD_NSH / 64];                    // (several shards per lane: all requested, then one wait)
            uint32_t uf = fwant;
#pragma unroll
            for (int k = 0; k < PD_NSH / 64; ++k) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(u[k]) : "v"(c + (lane + 64 * k) * 32) : "memory");
            if (flag) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(uf) : "v"(flag) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("" : "+v"(uf));
            bool ok = uf >= fwant;
#pragma unroll
            for (int k = 0; k < PD_NSH / 64; ++k) { asm volatile("" : "+v"(u[k])); ok = ok && u[k] >= want; }
            if (__builtin_amdgcn_readfirstlane(__popcll(__ballot(ok))) == 64) break;
            if (++spins > PD_SPIN_MAX) { if (lane == 0) pd_give_up(A, sh, true); break; }
            if ((spins & 63u) == 0u && __builtin_amdgcn_readfirstlane(pd_ld_sc1(A.err)) != 0u) { if (lane == 0) pd_give_up(A, sh, false); break; }
            PD_SLEEP();
        }
    }
    __syncthreads();
}
// every storing wave has drained; then one lane signals for the workgroup
__device__ __forceinline__ void pd_arrive(uint32_t* c) {
    pd_drain();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(c + (blockIdx.x & (PD_NSH - 1)) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- GEMM pieces (lean_body.h's arithmetic): wave w owns k-blocks w, w + 8, ... of a blocked A operand [kb][MT][64][4]
// fragments i0 <= i < i1 of the wave's k-blocks kb = wave + 8 i, both M-tiles (MT == 1: the second is the first again, never stored)
template <int I0, int I1, int NB>
__device__ __forceinline__ void pd_xload(const float* base, int MT, float4 (&x0)[NB], float4 (&x1)[NB]) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const auto rs = gt_rsrc(base, 0x7FFFF000u);
    const uint32_t m1 = (uint32_t)min(1, MT - 1) * 1024u;
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        const uint32_t so = (uint32_t)((wave + i * PD_NW) * MT) * 1024u;
        x0[i] = gt_bload4_sc1(rs, (uint32_t)lane * 16u, so);
        x1[i] = gt_bload4_sc1(rs, (uint32_t)lane * 16u, so + m1);
    }
}
// the same for group g of MTG M-tiles (M-tiles MTG g, MTG g + 1; MTG == 1: only x0)
template <int MTG, int I0, int I1, int NB>
__device__ __forceinline__ void pd_g_xload(const float* base, int MT, int g, float4 (&x0)[NB], float4 (&x1)[NB]) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const auto rs = gt_rsrc(base, 0x7FFFF000u);
    const uint32_t m0 = (uint32_t)(MTG * g) * 1024u, m1 = (uint32_t)min(MTG * g + 1, MT - 1) * 1024u;
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        const uint32_t so = (uint32_t)((wave + i * PD_NW) * MT) * 1024u;
        x0[i] = gt_bload4_sc1(rs, (uint32_t)lane * 16u, so + m0);
        if (MTG == 2) x1[i] = gt_bload4_sc1(rs, (uint32_t)lane * 16u, so + m1);
    }
}
// b[OFF + i * STRIDE], i < KPW (compile-time indices only: the weight fragments must stay registers)
template <int KPW, int OFF, int STRIDE, int NB>
__device__ __forceinline__ void pd_mma(const float4 (&x0)[NB], const float4 (&x1)[NB], const float4 (&b)[NB], f32x4& a0, f32x4& a1) {
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int k = OFF + i * STRIDE;
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[k].x, b[k].x, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[k].x, b[k].x, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[k].y, b[k].y, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[k].y, b[k].y, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[k].z, b[k].z, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[k].z, b[k].z, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[k].w, b[k].w, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[k].w, b[k].w, a1, 0, 0, 0);
    }
}
template <int MTG, int KPW, int OFF, int STRIDE, int NB>
__device__ __forceinline__ void pd_g_mma(const float4 (&x0)[NB], const float4 (&x1)[NB], const float4 (&b)[NB], f32x4& a0, f32x4& a1) {
    if constexpr (MTG == 2) {
        pd_mma<KPW, OFF, STRIDE, NB>(x0, x1, b, a0, a1);
    } else {
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            const int k = OFF + i * STRIDE;
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[k].x, b[k].x, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[k].y, b[k].y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[k].z, b[k].z, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[k].w, b[k].w, a0, 0, 0, 0);
        }
    }
}
__device__ __forceinline__ void pd_spill(float* lds, int slab, const f32x4& a0, const f32x4& a1) {
    float (*part)[32][17] = reinterpret_cast<float (*)[32][17]>(lds);
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int v = 0; v < 4; ++v) { part[slab][q * 4 + v][r] = a0[v]; part[slab][16 + q * 4 + v][r] = a1[v]; }
}
// element (row = tid >> 4, col = tid & 15) = base + sum over the first `nslab` slabs (ascending)
// TRAIL = false: without the closing barrier -- for the caller whose next spill into the slabs lies behind another workgroup barrier
// anyway (an arrival, a wait): the epilogue's stores then leave one barrier earlier
template <int NSLAB, bool TRAIL = true>
__device__ __forceinline__ float pd_reduce(float* lds, float base) {
    __syncthreads();
    const float (*part)[32][17] = reinterpret_cast<const float (*)[32][17]>(lds);
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    float z = base;
#pragma unroll
    for (int w = 0; w < NSLAB; ++w) z += part[w][row][col];
    if (TRAIL) __syncthreads();            // the slabs are re-used by the next reduction
    return z;
}
// the same, and one lane signals an arrival counter behind the first barrier (`cnt` != NULL): the DEFERRED arrival of an earlier
// store -- every wave has drained its stores before it got here (pd_drain in front of the spill)
template <int NSLAB>
__device__ __forceinline__ float pd_reduce_arrive(float* lds, float base, uint32_t* cnt) {
    __syncthreads();
    if (cnt && threadIdx.x == 0) __hip_atomic_fetch_add(cnt + (blockIdx.x & (PD_NSH - 1)) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float (*part)[32][17] = reinterpret_cast<const float (*)[32][17]>(lds);
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    float z = base;
#pragma unroll
    for (int w = 0; w < NSLAB; ++w) z += part[w][row][col];
    __syncthreads();
    return z;
}
// h . W_h (+ bias) of one tile from the fragments x0 / x1 of the whole state (8 k-blocks per wave).  ORDER16: the summation order
// of the launch path's 16-wave workers (gt_lean_partial<16, 4, 2>: virtual wave v owns k-blocks v, v + 16, v + 32, v + 48; this
// wave plays v = wave and v = wave + 8), else the 8-wave order of the projection launch's co-workers (gt_lean_partial<8, 8, 1>).
template <bool ORDER16>
__device__ __forceinline__ float pd_rec_tile(const float4 (&x0)[8], const float4 (&x1)[8], const float4 (&wh)[8], float bias, float* lds) {
    const int wave = threadIdx.x >> 6;
    if (ORDER16) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
        pd_mma<4, 0, 2, 8>(x0, x1, wh, a0, a1);         // k-blocks wave + 16 j
        pd_mma<4, 1, 2, 8>(x0, x1, wh, b0, b1);         // k-blocks wave + 8 + 16 j
        pd_spill(lds, wave, a0, a1);
        pd_spill(lds, wave + 8, b0, b1);
        return pd_reduce<16>(lds, bias);
    } else {
        f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
        pd_mma<8, 0, 1, 8>(x0, x1, wh, a0, a1);
        pd_spill(lds, wave, a0, a1);
        return pd_reduce<8>(lds, bias);
    }
}
// The same for a group of MTG M-tiles.  TWOPASS: the 16-slab sum through EIGHT slabs -- virtual waves 0..7, then 8..15 onto the running
// sum: the same sequence of additions (chain workgroups, whose LDS holds the utterance's processed memory).
template <bool ORDER16, int MTG, bool TWOPASS>
__device__ __forceinline__ float pd_g_rec_tile(const float4 (&x0)[8], const float4 (&x1)[8], const float4 (&wh)[8], float bias, float* lds) {
    const int wave = threadIdx.x >> 6;
    if (ORDER16) {
        f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
        pd_g_mma<MTG, 4, 0, 2, 8>(x0, x1, wh, a0, a1);
        pd_g_mma<MTG, 4, 1, 2, 8>(x0, x1, wh, b0, b1);
        if (TWOPASS) {
            pd_spill(lds, wave, a0, a1);
            const float z = pd_reduce<8>(lds, bias);
            pd_spill(lds, wave, b0, b1);
            return pd_reduce<8>(lds, z);
        }
        pd_spill(lds, wave, a0, a1);
        pd_spill(lds, wave + 8, b0, b1);
        return pd_reduce<16>(lds, bias);
    } else {
        f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
        pd_g_mma<MTG, 8, 0, 1, 8>(x0, x1, wh, a0, a1);
        pd_spill(lds, wave, a0, a1);
        return pd_reduce<8>(lds, bias);
    }
}
// gates of tile-local column g*4+u (i, f, c~, o of unit tile*4+u; Appendix A.6, reference Taco2.py:79-85): lanes col < 4 own a unit;
// h of the tile's 4 units leaves as ONE 16-byte write-through store per row (rows >= M are never stored)
__device__ __forceinline__ void pd_gates_store(float z, float& c, float* hdst, int tile, int M, int MT, int row0 = 0, int rows = 32) {
    const int lrow = threadIdx.x >> 4, row = row0 + lrow, col = threadIdx.x & 15;       // (a group: rows row0 .. row0 + rows - 1)
    const float zf = gt_row_down<4>(z), zg = gt_row_down<8>(z), zo = gt_row_down<12>(z);
    float hv = 0.f;
    if (lrow >= rows) return;               // (16-row groups: the slab's upper half is unused; whole 16-lane segments leave together)
    if (col < 4 && row < M) {
        const float gi = gt_sigmoid(z), gf = gt_sigmoid(zf), gg = gt_tanh(zg), go = gt_sigmoid(zo);
        c = __builtin_fmaf(gf, c, gi * gg);
        hv = go * gt_tanh(c);
    }
    const float h1v = gt_row_down<1>(hv), h2v = gt_row_down<2>(hv), h3v = gt_row_down<3>(hv);
    if (col == 0 && row < M) pd_st4_sc1(hdst + gt_blk_off(row, tile * 4, MT), make_float4(hv, h1v, h2v, h3v));
}
template <int KPW>
__device__ __forceinline__ void pd_load_tile(const float* wp, int tile, float4 (&dst)[KPW]) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const float4* wl = reinterpret_cast<const float4*>(wp) + ((size_t)tile * (KPW * PD_NW) + wave) * 64 + lane;
#pragma unroll
    for (int i = 0; i < KPW; ++i) dst[i] = wl[(size_t)i * PD_NW * 64];
}
// a [32][16] tile of recurrent-half sums (thread = element) for another workgroup: 16-byte write-through stores, drained, one flag
__device__ __forceinline__ void pd_publish_part(float v, float* dst, uint32_t* flag, uint32_t tag) {
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    const float v1 = gt_row_down<1>(v), v2 = gt_row_down<2>(v), v3 = gt_row_down<3>(v);
    if ((col & 3) == 0) pd_st4_sc1(dst + row * 16 + col, make_float4(v, v1, v2, v3));
    pd_drain();
    __syncthreads();
    if (threadIdx.x == 0) pd_st1_sc1(flag, tag);
}

struct PdW { float4 x1[3], h1[8], x2[8], h2[8]; };

// ====================================================================================================================== phases
// LSTM cell 1: z = [p | ctx] . W1x + part1 (= h1_{t-1} . W1h + b1).  split: the prenet part is multiplied as soon as the chains
// have published it, the context part when it arrives (same accumulator, same k-block order as gt_lstm_x_kernel<8, 3>).
__device__ __forceinline__ void pd_cell1(const PersistDecodeArgs& A, PdW& W, int t, int tile, float* lds, float& c1v, float p1v, PdShared* sh, bool split,
                                        bool stream_x1 = false, int role = 2, bool stream_x2 = false) {
    const int par = t & 1, MT = A.MT;
    if (stream_x1) pd_load_tile<3>(A.w1x, tile, W.x1);          // (the chain role: arrives while the other chains finish)
    const float* xa = A.xa[par];
    float* h1d = A.h1[par];
    PD_HOLD(xa); PD_HOLD(h1d);
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    float4 x0[3], x1[3];
    if (split) {
        pd_wait_flags<PD_FS>(A, A.ctl + PD_F_P, A.B, (uint32_t)t + 1u, sh);
        PD_PHASE_ABORT(sh);
        PD_STAMP(role, 1);
        pd_xload<0, 2, 3>(xa, MT, x0, x1);
        PD_PIN();
        pd_mma<2, 0, 1, 3>(x0, x1, W.x1, a0, a1);
        PD_STAMP(role, 2);
        pd_wait_flags<PD_FS>(A, A.ctl + PD_F_C, A.B, (uint32_t)t + 1u, sh);
        PD_PHASE_ABORT(sh);
        PD_STAMP(role, 3);
        pd_xload<2, 3, 3>(xa, MT, x0, x1);
        PD_PIN();
        pd_mma<1, 2, 1, 3>(x0, x1, W.x1, a0, a1);
    } else {
        pd_wait_flags<PD_FS>(A, A.ctl + PD_F_C, A.B, (uint32_t)t + 1u, sh);      // (a chain's context flag is set after its prenet flag)
        PD_PHASE_ABORT(sh);
        PD_STAMP(role, 3);
        pd_xload<0, 3, 3>(xa, MT, x0, x1);
        // (the chain role: cell 2's input half, behind this cell's fragments -- requested at cell 2's entry, the poll of the h1 arrivals
        // queued behind its 64 KB, loads returning in order, and this role, which everybody waits for, saw the hand-off ~1 us late)
        // (the tile index through an opaque register: its load addresses are otherwise hoisted out of the step loop, live across the
        // chain, and the allocator -- at 250 of 256 registers there -- spills three dozen other loop invariants)
        if (stream_x2) { int tz = tile; asm volatile("" : "+s"(tz)); pd_load_tile<8>(A.w2x, tz, W.x2); }
        PD_PIN();
        pd_mma<3, 0, 1, 3>(x0, x1, W.x1, a0, a1);
    }
    pd_spill(lds, threadIdx.x >> 6, a0, a1);
    const float z = pd_reduce<8, false>(lds, p1v);          // (pd_arrive's barrier below closes the slabs)
    pd_gates_store(z, c1v, h1d, tile, A.B, MT);
    PD_STAMP(role, 4);
    pd_arrive(A.ctl + PD_CNT3);
}

// LSTM cell 2: z = h1_t . W2x + part2; then, from the same fragments, recurrent halves of cell 1 for the NEXT step: a chain
// workgroup's (help_tile >= 0, published for it) and this workgroup's own (with_rec1)
// helped_p2 >= 0 (chain role, t > 0): this tile's layer-2 recurrent half comes from its helper -- the helper's flag rides in the poll
// of the h1 arrivals, the sums (slot helped_p2 of hpart) are requested with the h1 fragments
__device__ __forceinline__ void pd_cell2(const PersistDecodeArgs& A, PdW& W, int t, int tile, float* lds, float& c2v, float& p2v, float& p1_next, bool with_rec1,
                                        PdShared* sh, int help_tile, bool stream_h1, bool stream_x2, int role = 2, int helped_p2 = -1, bool helped_role = false) {
    const int par = t & 1, MT = A.MT;
    // (the chain role streams W2x: these registers belong to the chain's operands until cell 1.  Requested BEHIND the wait, with the
    // h1 fragments: in front of it the poll queued behind its 64 KB -- loads return in order -- and this role, which everybody waits
    // for, saw the hand-off ~1 us after the others)
    float4 wu[8];
    if (help_tile >= 0) pd_load_tile<8>(A.w1h, help_tile, wu);  // (a chain workgroup's W1h tile, streamed: arrives during the wait)
    if (stream_h1) pd_load_tile<8>(A.w1h, tile, W.h1);          // (layer-2 helpers keep W2h resident and stream their own W1h)
    const float* h1p = A.h1[par];
    float* h2d = A.h2[par];
    PD_HOLD(h1p); PD_HOLD(h2d);
    pd_wait_count(A, A.ctl + PD_CNT3, PD_WANT(A, t), sh, helped_p2 >= 0 ? A.ctl + PD_F_H + helped_p2 * 32 : nullptr, (uint32_t)t);
    PD_PHASE_ABORT(sh);
    PD_STAMP(role, 5);
    float4 x0[8], x1[8];
    if (stream_x2) pd_load_tile<8>(A.w2x, tile, W.x2);
    pd_xload<0, 8, 8>(h1p, MT, x0, x1);
    // (the helper's sums: requested with the fragments -- out of range when there are none: no branch around the request -- and looked
    // at BEHIND the MFMAs: a use right here, or inside a branch, is a wait for every fragment before the first MFMA instead of a
    // counted wait per fragment)
    float4 d2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (helped_role) {
        const auto rh = gt_rsrc(A.hpart, 2u * 32u * 512u * 4u);
        d2 = gt_bload4_sc1(rh, helped_p2 >= 0 ? (uint32_t)((helped_p2 * 512 + ((int)threadIdx.x & ~3)) * 4) : GT_OOB, 0u);
    }
    PD_PIN();
    PD_STAMP(role, 21);         // (fragments requested)
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
    pd_mma<8, 0, 1, 8>(x0, x1, W.x2, a0, a1);
    PD_STAMP(role, 6);
    if (helped_role && helped_p2 >= 0) {
        const int e = threadIdx.x & 3;
        p2v = e == 0 ? d2.x : e == 1 ? d2.y : e == 2 ? d2.z : d2.w;
    }
    pd_spill(lds, threadIdx.x >> 6, a0, a1);
    const float z = pd_reduce<8, false>(lds, p2v);
    PD_STAMP(role, 22);         // (reduced)
    pd_gates_store(z, c2v, h2d, tile, A.B, MT);
    PD_STAMP(role, 7);
    pd_arrive(A.ctl + PD_CNT4);
    PD_STAMP(role, 8);
    const int col = threadIdx.x & 15;
    if (help_tile >= 0) {       // (the chain workgroup's half first: it is waited for sooner than this workgroup's own)
        const float v = pd_rec_tile<true>(x0, x1, wu, A.b1h[help_tile * 16 + col], lds);
        pd_publish_part(v, A.hpart + (size_t)help_tile * 512, A.ctl + PD_F_H + help_tile * 32, (uint32_t)t + 1u);
    }
    if (with_rec1) p1_next = pd_rec_tile<true>(x0, x1, W.h1, A.b1h[tile * 16 + col], lds);
}

// layer-2 recurrent half of this workgroup's tile from h2 in memory (+ a chain workgroup's, for helpers); tiles below co_tiles sum
// in the projection launch's co-workers' order (8 waves), the others in the front launch's workers' (16 waves)
__device__ __forceinline__ float pd_rec2(const PersistDecodeArgs& A, int t, const float* hbuf, const float4 (&wh)[8], int tile, int help_tile, float* lds) {
    float4 wu[8];
    if (help_tile >= 0) pd_load_tile<8>(A.w2h, help_tile, wu);
    float4 x0[8], x1[8];
    pd_xload<0, 8, 8>(hbuf, A.MT, x0, x1);
    PD_PIN();
    const int col = threadIdx.x & 15;
    if (help_tile >= 0) {
        const float v = help_tile < A.co_tiles ? pd_rec_tile<false>(x0, x1, wu, A.b2h[help_tile * 16 + col], lds)
                                               : pd_rec_tile<true>(x0, x1, wu, A.b2h[help_tile * 16 + col], lds);
        pd_publish_part(v, A.hpart + (size_t)(32 + help_tile) * 512, A.ctl + PD_F_H + (32 + help_tile) * 32, (uint32_t)t + 1u);
    }
    return tile < A.co_tiles ? pd_rec_tile<false>(x0, x1, wh, A.b2h[tile * 16 + col], lds) : pd_rec_tile<true>(x0, x1, wh, A.b2h[tile * 16 + col], lds);
}
__device__ __forceinline__ float pd_rec1_mem(const PersistDecodeArgs& A, const float* hbuf, const float4 (&wh)[8], int tile, float* lds) {
    float4 x0[8], x1[8];
    pd_xload<0, 8, 8>(hbuf, A.MT, x0, x1);
    PD_PIN();
    return pd_rec_tile<true>(x0, x1, wh, A.b1h[tile * 16 + (threadIdx.x & 15)], lds);
}

// Projection tile `ptile`, M-tile `pmt` (Taco2.py:112-118: r mel frames | stop logit) + the next step's prenet-0 pre-activations
// (both layers are linear: gsttaco.cpp proj_z), gt_proj_lean_kernel's arithmetic
template <bool GK = false>     // GK: called by a group kernel (its stamp slots)
__device__ __forceinline__ void pd_proj(const PersistDecodeArgs& A, const float4 (&wp)[9], int t, int ptile, int pmt, float* lds, PdShared* sh, int g = 0) {
    const int par = t & 1, MT = A.MT;
    const float* h2p = A.h2[par];
    const float* xap = A.xa[par];
    uint2* z0g = A.z0g;
    PD_HOLD(h2p); PD_HOLD(xap); PD_HOLD(z0g);
    if (!GK) {          // (a group kernel has waited for every group's arrivals at once)
        pd_wait_count(A, A.ctl + PD_CNT4, PD_WANT(A, t), sh);
        PD_PHASE_ABORT(sh);
    }
    if (!GK) PD_STAMP(1, 9);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const auto rh = gt_rsrc(h2p, 0x7FFFF000u);
    const auto rx = gt_rsrc(xap, 0x7FFFF000u);
    float4 x[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int kb = wave + i * PD_NW;                                // wave-uniform; k-blocks [0, 64) = h2, [64, 72) = context
        x[i] = kb < PD_KBH ? gt_bload4_sc1(rh, (uint32_t)lane * 16u, (uint32_t)((kb * MT + pmt) * 1024))
                           : gt_bload4_sc1(rx, (uint32_t)lane * 16u, (uint32_t)(((kb - PD_KBH + PD_KBP) * MT + pmt) * 1024));
    }
    PD_PIN();
    f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i].x, wp[i].x, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i].y, wp[i].y, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i].z, wp[i].z, a0, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i].w, wp[i].w, a0, 0, 0, 0);
    }
    if (!GK) PD_STAMP(1, 10);
    pd_spill(lds, threadIdx.x >> 6, a0, a1);
    // (one-group kernel: the projection role's next spill -- the next step's cell 1 -- lies behind that phase's flag wait and its barrier)
    const float v = GK ? pd_reduce<8>(lds, A.bp[ptile * 16 + (threadIdx.x & 15)]) : pd_reduce<8, false>(lds, A.bp[ptile * 16 + (threadIdx.x & 15)]);
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    const int grow = pmt * 16 + row, gcol = ptile * 16 + col;
    if (row < 16 && grow < A.B) {
        if (gcol >= A.z_col0) {
            if (gcol < A.z_col0 + PD_P) {
                uint2 g;
                g.x = __builtin_bit_cast(uint32_t, v); g.y = (uint32_t)t + 1u;
                pd_st2_sc1(z0g + (size_t)grow * PD_P + (gcol - A.z_col0), g);
            }
        } else if (gcol < A.n_split) {
            A.pre[(size_t)grow * A.ld_pre + (size_t)t * A.n_split + gcol] = v;
        } else if (gcol < A.n_out) {
            A.stop[(size_t)grow * A.steps + t] = v;
        }
    }
    if (!GK) PD_STAMP(1, 11);
    else PD_STAMP(1, 6 + 7 * g);
}

// ---- the per-utterance chain (front_lean.h's arithmetic; every thread plays launch-path threads tid and tid + 512)
struct PdChainLds {
    float *y0, *y1, *qs, *vs, *sc, *nz, *pv, *al, *ks1, *partial, *red, *tile;
    float *lfeat, *lpack;       // LSA (one-group kernel, tvp = 128): location features [128][LFS] and the weight image (kernels.h LsaPack) behind the tile
};
// LDS of a chain workgroup (floats): [8 reduction slabs of its LSTM tile | small vectors | the utterance's processed memory, tvp rows].
// The chain's GEMV partials (16 x 256) and the context's row-group sums (8 x 128) ALIAS the slabs: the chain is over (a barrier
// behind its last LDS read) before the cells spill, and the cells' last reduction ends in a barrier before the next step's chain
// writes.  tvp = T_v rounded up to the 64 rows of a score pass: 157 KB at 256 tokens -- the CU's 160 KB hold one utterance.
// The other roles use 16 slabs (the 16-wave summation order of the recurrent halves) and nothing else.
constexpr int PD_SLAB = 32 * 17;
constexpr int PD_OTHER_FLOATS = 16 * PD_SLAB + 4 * PD_GMAX * PD_NT;      // (+ the group kernels' projection role: its per-group state)
__host__ __device__ constexpr int pd_chain_floats(int tvp, int nslab) { return nslab * PD_SLAB + 3 * PD_P + 2 * PD_A + 4 * tvp + tvp * PD_LDV; }
// the group kernels' chain workgroups also sum their own recurrent halves (16 slabs in the launch path's 16-wave order): with 16
// slabs while they fit beside the processed memory (T_v <= 192), else in two passes over 8
__host__ __device__ constexpr int pd_chain_slabs(int tvp, bool group_kernel) { return group_kernel && pd_chain_floats(tvp, 16) * 4 + 64 <= 160 * 1024 ? 16 : 8; }
__host__ __device__ constexpr int pd_lds_floats(int tvp, bool group_kernel = false) {
    return pd_chain_floats(tvp, pd_chain_slabs(tvp, group_kernel)) > PD_OTHER_FLOATS ? pd_chain_floats(tvp, pd_chain_slabs(tvp, group_kernel)) : PD_OTHER_FLOATS;
}
static_assert(16 * PD_P <= 8 * PD_SLAB && 8 * PD_A <= 8 * PD_SLAB, "the chain's partial sums alias the reduction slabs");
static_assert(pd_lds_floats(PD_TVMAX, true) * 4 + 64 <= 160 * 1024 && pd_lds_floats(PD_TVMAX, false) * 4 + 64 <= 160 * 1024, "one utterance's chain state must fit a CU's LDS");
__device__ __forceinline__ PdChainLds pd_carve(float* smem, int tvp, int nslab = 8, int slab_floats = PD_SLAB) {
    PdChainLds L;
    L.partial = smem; L.red = smem;
    L.y0 = smem + nslab * slab_floats; L.y1 = L.y0 + PD_P; L.qs = L.y1 + PD_P; L.vs = L.qs + PD_A; L.sc = L.vs + PD_A; L.nz = L.sc + tvp; L.pv = L.nz + tvp;
    L.al = L.pv + tvp; L.ks1 = L.al + tvp; L.tile = L.ks1 + PD_P;
    L.lfeat = L.tile + tvp * PD_LDV; L.lpack = L.lfeat;        // (pd_lsa_carve)
    return L;
}

struct PdChainRegs { float bias1, biasq, sbias; int Tv; uint64_t seed; bool hashed, drop, noisy; };
// the LSA chain's LDS behind the 128-row tile: location features of the tile's rows + the weight image
__host__ __device__ inline int pd_lsa_floats(int loc_f, int loc_k) { const LsaPack lp = gt_lsa_pack(PD_A, loc_f, loc_k); return 128 * lp.LFS + lp.total; }

// ---- ALL of prenet-1's weights for step t, requested before the projection's hand-off is even looked at (the bf16 kernel: a phase
// earlier still, in front of the recurrent half of the step before).  Launch-path wave kp (of 16) owns rows 16 kp .. 16 kp + 15,
// lane = 4 output columns; this wave plays kp = wave and kp = wave + 8.  Hashed dropout (throughput mode): rows whose input the keep
// decisions zero are not requested (they would multiply an exact zero).
// WHICH: 3 = both halves, 1 = the first (virtual wave kp = wave), 2 = the second (kp = wave + 8)
template <int WHICH = 3>
__device__ __forceinline__ void pd_chain_issue(const PersistDecodeArgs& A, const PdChainRegs& R, int t, int b, int zt, float4 (&ra)[16], float4 (&rb)[16]) {
    const int tid = threadIdx.x + zt, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const auto rsW1 = gt_rsrc(A.W1, (uint32_t)(PD_P * PD_P) * 4u);
    uint32_t rba = 0xFFFFu, rbb = 0xFFFFu;
    if (R.hashed) {
        const uint32_t w0 = gt_keep_word(R.seed, (uint32_t)t, 0u, (uint32_t)b, (uint32_t)wave >> 1);
        const uint32_t w1 = gt_keep_word(R.seed, (uint32_t)t, 0u, (uint32_t)b, ((uint32_t)wave >> 1) + 4u);
        rba = (w0 >> ((wave & 1) * 16)) & 0xFFFFu;
        rbb = (w1 >> ((wave & 1) * 16)) & 0xFFFFu;
    }
    // (a dropped row is skipped by a wave-uniform branch, not requested out of range: an out-of-range buffer load still costs the
    // CU's address pipe 85 % of a real one -- tools/oob_cost.hip -- and the pipe time of these 32 requests per wave is what decides
    // whether the weights are there when the projection's hand-off arrives.  The matching FMAs are skipped too: they would add
    // x = 0 times anything.)
    if (WHICH & 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((rba >> i) & 1u) ra[i] = gt_bload4(rsW1, (uint32_t)lane * 16u, (uint32_t)((16 * wave + i) * PD_P * 4));
        }
    }
    if (WHICH & 2) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((rbb >> i) & 1u) rb[i] = gt_bload4(rsW1, (uint32_t)lane * 16u, (uint32_t)((16 * (wave + 8) + i) * PD_P * 4));
        }
    }
}

// the chain behind its weight requests (pd_chain_issue)
// TV128: at most 128 tokens (the headline shape: two score passes, one context chunk -- as compile-time constants they are worth
// ~0.2 us per step, same-box A/B profiles/r05_ab.txt)
// LSA chain: location features of the 128 rows from the state in L.pv = Conv1D(state) + bias as a Toeplitz product (front_body.h), for the
// score pass of the step: called at the step's start, in front of the poll for the projection's hand-off (the state was updated at the end
// of the previous step's chain, barriers ago)
__device__ __forceinline__ void pd_lsa_features(const PersistDecodeArgs& A, const PdChainLds& L, int Tv) {
    const LsaPack lp = gt_lsa_pack(PD_A, A.loc_f, A.loc_k);
    const int LK = A.loc_k, LFS = lp.LFS, LCS = lp.LCS;
    const float* lcw = L.lpack + lp.off_cw;
    const float* lcbs = L.lpack + lp.off_cb;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, lq = lane >> 4, lpad = (LK - 1) / 2;
    const int NF = lp.LFc >> 4;
    for (int it = wave; it < 8 * NF; it += PD_NW) {
        const int m = it / NF, n = it - m * NF;
        const float cb = lcbs[16 * n + l15];
        f32x4 acc = {cb, cb, cb, cb};
#pragma unroll 2
        for (int ks = 0; ks < (lp.LKp >> 2); ++ks) {
            const int j = 4 * ks + lq, ts = 16 * m + l15 + j - lpad;
            const float xv = (j < LK && ts >= 0 && ts < Tv) ? L.pv[ts] : 0.f;
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xv, lcw[j * LCS + 16 * n + l15], acc, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) L.lfeat[(16 * m + 4 * lq + i) * LFS + 16 * n + l15] = acc[i];
    }
}

// (give-up invariant, PD_PHASE_ABORT: the LSA chain's loop bounds -- filter / tap counts, Tv -- are launch arguments or the caller's token
// lengths, its LDS indices functions of the thread index: nothing behind a wait is data-dependent)
// LSA (with TV128, tvp = 128; the one-group kernel): the step-wise location-sensitive extension in the chain -- dec_front_lsa.hip's two
// MFMA products on 8 waves instead of 16, each score row's two channel halves summed as the 16-wave kernel's two waves sum them
// (bitwise that kernel); the weight image stays in LDS for the whole launch, the cumulative alignment lives in L.pv
template <bool HELPED, bool MIRROR, bool TV128 = false, bool LSA = false>
__device__ __forceinline__ void pd_chain_rest(const PersistDecodeArgs& A, const PdChainLds& L, const PdChainRegs& R, int t, int b, PdShared* sh, float4& hraw, int zt,
                                              float4 (&ra)[16], float4 (&rb)[16]) {
    const int tid = threadIdx.x + zt, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int par = t & 1, Tv = R.Tv, TvFull = A.Tv, MT = A.MT;
    const auto rsWq = gt_rsrc(A.Wq, (uint32_t)(PD_P * PD_A) * 4u);
    // small per-step operands: noise row, injected keep masks (parity mode)
    float nzv = 0.f, k0 = 1.f, k1 = 1.f;
    if (R.noisy && tid < Tv) nzv = A.noise[((size_t)t * A.B + b) * TvFull + tid];
    if (R.drop && !R.hashed && tid < PD_P) {
        const float* m = A.masks + (size_t)t * A.B * (2 * PD_P);
        k0 = m[(size_t)b * PD_P + tid];
        k1 = m[(size_t)A.B * PD_P + (size_t)b * PD_P + tid];
    }
    // (which query rows this thread requests below -- launch-path thread = (4 output columns cgq, k-part kp of 8 rows), this thread
    // plays kp = tid / 32 and 16 + tid / 32 -- derived here, in front of the poll, with the other keep decisions)
    const int cgq = tid & 31, kpa = tid >> 5, kpb = 16 + (tid >> 5);
    uint32_t qba = 0xFFu, qbb = 0xFFu;
    if (R.hashed) {
        qba = (gt_keep_word(R.seed, (uint32_t)t, 1u, (uint32_t)b, (uint32_t)kpa >> 2) >> ((8 * kpa) & 31)) & 0xFFu;
        qbb = (gt_keep_word(R.seed, (uint32_t)t, 1u, (uint32_t)b, (uint32_t)kpb >> 2) >> ((8 * kpb) & 31)) & 0xFFu;
    }
    // (a row is requested when either half of the wave keeps it: the two halves are different k-parts)
    const uint32_t ua = (uint32_t)__builtin_amdgcn_readlane((int)qba, 0) | (uint32_t)__builtin_amdgcn_readlane((int)qba, 32);
    const uint32_t ub = (uint32_t)__builtin_amdgcn_readlane((int)qbb, 0) | (uint32_t)__builtin_amdgcn_readlane((int)qbb, 32);
    if (tid < Tv) L.nz[tid] = A.sigmoid_noise * nzv;        // (in front of the poll too: behind it, a kernel-argument reload sat on the critical path)
    // (LSA: this step's location features from the state the previous step left -- here, where the workgroup waits for the projection's
    // hand-off anyway; in front of the score pass they were 0.7 us of the critical path.  The barriers below lie in front of their use.)
    if (LSA) pd_lsa_features(A, L, Tv);
    PD_PIN();
    PD_STAMP(0, 13);
    // ---- S1: this utterance's row of prenet-0 pre-activations (granules tagged with the step they are for)
    if (tid < PD_P) {
        // (the keep decisions and everything else that does not need the granule: in FRONT of the poll, while the projection of the
        // previous step is still on its way -- behind it they were ~100 instructions on the step's critical path)
        if (R.hashed) {
            k0 = gt_drop_keep(R.seed, (uint32_t)t, 0u, (uint32_t)b, (uint32_t)tid, (uint32_t)PD_P, A.drop_rate);
            k1 = gt_drop_keep(R.seed, (uint32_t)t, 1u, (uint32_t)b, (uint32_t)tid, (uint32_t)PD_P, A.drop_rate);
        }
        if (R.drop) { k0 *= A.drop_scale; k1 *= A.drop_scale; }
        else { k0 = 1.f; k1 = 1.f; }
        L.ks1[tid] = k1;
        asm volatile("" : "+v"(k0) : : "memory");
        uint2 g = pd_ld2_sc1(A.z0g + (size_t)b * PD_P + tid);
        uint32_t spins = 0;
        while (__builtin_amdgcn_readfirstlane(__popcll(__ballot(g.y == (uint32_t)t))) != 64) {
            if (++spins > PD_SPIN_MAX) { if (lane == 0) pd_give_up(A, sh, true); break; }
            if ((spins & 63u) == 0u && __builtin_amdgcn_readfirstlane(pd_ld_sc1(A.err)) != 0u) { if (lane == 0) pd_give_up(A, sh, false); break; }
            g = pd_ld2_sc1(A.z0g + (size_t)b * PD_P + tid);
        }
        L.y0[tid] = fmaxf(__builtin_bit_cast(float, g.x), 0.f) * k0;
    }
    __syncthreads();
    PD_PHASE_ABORT(sh);
    PD_STAMP(0, 14);
    // ---- prenet layer 1 (Taco2.py:262-283): two launch-path threads' 16-row sums each.  The query weights (Steps.py:122) are
    // requested into the first half's registers: row i's request right behind the multiply-add that consumed register i, so the 16
    // requests of a wave -- a CU's address pipe takes ~16 cycles for each -- go out under the first half's arithmetic instead of
    // as a burst between the halves; rows whose input (mask 1) is dropped are not requested.
    float4 (&qa)[16] = ra;
    {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        float x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = L.y0[16 * wave + i];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            gt_fma4(acc, x[i], ra[i]);
            PD_PIN4(acc);
            qa[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < 8) {
                if ((ua >> i) & 1u) qa[i] = gt_bload4(rsWq, ((qba >> i) & 1u) ? (uint32_t)cgq * 16u : GT_OOB, (uint32_t)((8 * kpa + i) * PD_A * 4));
            } else {
                if ((ub >> (i - 8)) & 1u) qa[i] = gt_bload4(rsWq, ((qbb >> (i - 8)) & 1u) ? (uint32_t)cgq * 16u : GT_OOB, (uint32_t)((8 * kpb + i - 8) * PD_A * 4));
            }
            PD_PIN();
        }
        *reinterpret_cast<float4*>(L.partial + wave * PD_P + 4 * lane) = acc;
        PD_PIN();
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = L.y0[16 * (wave + 8) + i];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            gt_fma4(acc, x[i], rb[i]);
        }
        PD_PIN4(acc);
        *reinterpret_cast<float4*>(L.partial + (wave + 8) * PD_P + 4 * lane) = acc;
    }
    PD_STAMP(0, 23);
    __syncthreads();
    PD_STAMP(0, 15);
    if (tid < PD_P) {
        const float v = fmaxf(reduce_partial(L.partial, 16, PD_P, tid) + R.bias1, 0.f) * L.ks1[tid];
        L.y1[tid] = v;
    }
    __syncthreads();
    // ---- S2p: the prenet output leaves now (the LSTM-1 workgroups multiply it while the attention below runs)
    float* xa = A.xa[par];
    if (tid < 64) {
        if (MIRROR) {
            const float4 y = *reinterpret_cast<const float4*>(L.y1 + 4 * tid);
            uint2 pk;
            pk.x = (uint32_t)gt_bf16_bits(y.x) | ((uint32_t)gt_bf16_bits(y.y) << 16);
            pk.y = (uint32_t)gt_bf16_bits(y.z) | ((uint32_t)gt_bf16_bits(y.w) << 16);
            pd_st2_sc1(reinterpret_cast<uint2*>(A.xah[par] + gt_blk_off_h(b, 4 * tid, MT)), pk);
        } else
        pd_st4_sc1(xa + gt_blk_off(b, 4 * tid, MT), *reinterpret_cast<const float4*>(L.y1 + 4 * tid));
        // (drained and flagged BEHIND the query projection's barrier below: waiting for the write acknowledgement here held the other
        // seven waves at that barrier for ~0.7 us of every step, and nobody is short of the prenet part -- the LSTM-1 workgroups that
        // multiply it early wait for the context anyway)
    }
    if (HELPED && t > 0 && tid >= PD_NT - 64) {        // the last wave, meanwhile: this tile's layer-1 recurrent half from its helper must show step t
        const int l2 = tid & 63;
        uint32_t spins = 0;
        for (;;) {
            const uint32_t v = l2 < 1 ? pd_ld_sc1(A.ctl + PD_F_H + b * 32) : (uint32_t)t;
            if (__builtin_amdgcn_readfirstlane(__popcll(__ballot(v >= (uint32_t)t))) == 64) break;
            if (++spins > PD_SPIN_MAX) { if (l2 == 0) pd_give_up(A, sh, true); break; }
            if ((spins & 63u) == 0u && __builtin_amdgcn_readfirstlane(pd_ld_sc1(A.err)) != 0u) { if (l2 == 0) pd_give_up(A, sh, false); break; }
        }
    }
    PD_STAMP(0, 16);
    // ---- query projection
    {
        float x[8];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = L.y1[8 * kpa + i];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            gt_fma4(acc, x[i], qa[i]);
        }
        *reinterpret_cast<float4*>(L.partial + kpa * PD_A + 4 * cgq) = acc;
        acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = L.y1[8 * kpb + i];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            gt_fma4(acc, x[i], qa[8 + i]);
        }
        *reinterpret_cast<float4*>(L.partial + kpb * PD_A + 4 * cgq) = acc;
    }
    __syncthreads();
    if (tid < 64) {                                 // S2p's flag: the prenet stores were acknowledged under the projection above
        pd_drain();
        if (tid == 0) pd_st1_sc1(A.ctl + PD_F_P + b * PD_FS, (uint32_t)t + 1u);
    }
    PD_PHASE_ABORT(sh);
    if (HELPED) {       // (behind the barrier the flag poll joined: the helper's sums, consumed by cell 1.  Requested without a branch
                        // -- out of range at step 0 -- and handed out raw: a use in here is a full memory wait inside the chain)
        const auto rh = gt_rsrc(A.hpart, 2u * 32u * 512u * 4u);
        hraw = gt_bload4_sc1(rh, t > 0 ? (uint32_t)((b * 512 + (tid & ~3)) * 4) : GT_OOB, 0u);
    }
    if (tid < PD_A) L.qs[tid] = reduce_partial(L.partial, 32, PD_A, tid) + R.biasq;
    __syncthreads();
    PD_STAMP(0, 17);
    // ---- scores (Steps.py:126-152): 8 lanes per memory row, 4 x 16-byte pieces each; rows tid / 8, 64 + tid / 8, ... (tvp / 64 passes)
    // (the thread's four pieces of the query and of v are the same for every row it plays: read once -- re-read per row they were
    // two thirds of the pass's LDS bytes, 128 KB of 192, on a pass that is LDS- and transcendental-bound)
    const int npass = TV128 ? 2 : A.tvp >> 6;
    if (LSA) {
        // (dec_front_lsa.hip / front_body.h, whose comments say why: location features = a Toeplitz product of the state, location term
        // = a second product, both on v_mfma_f32_16x16x4_f32 -- a k-ordered fmaf chain, so a tile's sums do not depend on which wave
        // computes it -- then score = sum_a tanh(q + key + loc + bias))
        const LsaPack lp = gt_lsa_pack(PD_A, A.loc_f, A.loc_k);
        const int LFS = lp.LFS, LDWS = lp.LDWS;
        const float* ldw = L.lpack;
        const float* labias = L.lpack + lp.off_ab;
        const int l15 = lane & 15, lq = lane >> 4;
        // (the location features were computed at the END of the previous step -- pd_lsa_features -- from the state that step left)
        {   // wave = 16-row tile; its two halves of the channels are the 16-wave kernel's two waves of that tile: summed separately, then added
            const int m = wave;
            float z[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
            for (int wn = 0; wn < 2; ++wn) {
                f32x4 acc[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) { const float lb = labias[16 * (wn * 4 + g) + l15]; acc[g] = f32x4{lb, lb, lb, lb}; }
                const float* lfr = L.lfeat + (16 * m + l15) * LFS + lq;
                const float* lwr = ldw + lq * LDWS + 16 * wn * 4 + l15;
#pragma unroll 2
                for (int ks = 0; ks < (lp.LFp >> 2); ++ks) {
                    const float fa = lfr[4 * ks];
#pragma unroll
                    for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, lwr[4 * ks * LDWS + 16 * g], acc[g], 0, 0, 0);
                }
                // (two rows at a time on the packed fp32 ops: per element the same operations in the same order as gt_tanh)
                f32x2 e01 = {0.f, 0.f}, e23 = {0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int a = 16 * (wn * 4 + g) + l15;
                    const float qa = L.qs[a];
                    const float* tr = L.tile + (16 * m + 4 * lq) * PD_LDV + a;
                    e01 += gt_tanh2((f32x2{qa, qa} + f32x2{tr[0], tr[PD_LDV]}) + f32x2{acc[g][0], acc[g][1]});
                    e23 += gt_tanh2((f32x2{qa, qa} + f32x2{tr[2 * PD_LDV], tr[3 * PD_LDV]}) + f32x2{acc[g][2], acc[g][3]});
                }
                z[0] += gt_row_sum<16>(e01.x); z[1] += gt_row_sum<16>(e01.y); z[2] += gt_row_sum<16>(e23.x); z[3] += gt_row_sum<16>(e23.y);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (l15 == 0 && 16 * m + 4 * lq + i < Tv) L.sc[16 * m + 4 * lq + i] = z[i];
        }
    }
    float4 qr[4], wr[4];
#pragma unroll
    for (int j = 0; j < 4 && !LSA; ++j) {
        qr[j] = *reinterpret_cast<const float4*>(L.qs + 4 * ((tid & 7) + 8 * j));
        wr[j] = *reinterpret_cast<const float4*>(L.vs + 4 * ((tid & 7) + 8 * j));
    }
#pragma unroll 1
    for (int hh = 0; hh < (LSA ? 0 : npass); ++hh) {
        const int row = hh * 64 + (tid >> 3), li = tid & 7;
        f32x2 s2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a0 = 4 * (li + 8 * j);
            const float4 m4 = *reinterpret_cast<const float4*>(L.tile + row * PD_LDV + a0);
            const float4 q4 = qr[j], w4 = wr[j];
            s2 = __builtin_elementwise_fma(f32x2{w4.x, w4.y}, gt_tanh2(f32x2{q4.x, q4.y} + f32x2{m4.x, m4.y}), s2);
            s2 = __builtin_elementwise_fma(f32x2{w4.z, w4.w}, gt_tanh2(f32x2{q4.z, q4.w} + f32x2{m4.z, m4.w}), s2);
        }
        float s = s2.x + s2.y;
        s = gt_row_sum<8>(s);
        if (li == 0 && row < Tv) L.sc[row] = s + R.sbias;
    }
    __syncthreads();
    PD_STAMP(0, 18);
    // ---- noise + sigmoid + alignment: SMA (Steps.py:215-229) or BMA (Steps.py:168-199)
    // alp (TV128): the alignment once more in the context pass's order -- row group cp's sixteen rows cp, cp + 8, .. side by side -- so that
    // pass reads them as four 16-byte words instead of sixteen broadcast reads (a third of its LDS instructions); rows >= Tv: zeros.
    // In the slabs behind the context's row-group sums: free between the query's reduction and the cells' first spill.
    float* alp = L.red + 8 * PD_A;
    if (LSA) {
        // softmax (or the smoothing normalisation, Layers.py:426-444) over the Tv positions: one wave, a serial run per lane (front_body.h)
        if (tid < 64) {
            const int per = (Tv + 63) / 64;
            const int t0 = lane * per, t1 = min(Tv, t0 + per);
            float mx = -INFINITY;
            for (int tt = t0; tt < t1; ++tt) mx = fmaxf(mx, L.sc[tt]);
            mx = gt_wave_max(mx);
            float sum = 0.f;
            for (int tt = t0; tt < t1; ++tt) {
                const float e = A.lsa_smoothing ? 1.f / (1.f + expf(-L.sc[tt])) : expf(L.sc[tt] - mx);
                L.al[tt] = e;
                sum += e;
            }
            sum = gt_wave_sum(sum);
            const float inv = 1.f / sum;
            for (int tt = t0; tt < t1; ++tt) {
                L.al[tt] *= inv;
                if (TV128) alp[(tt & 7) * 16 + (tt >> 3)] = L.al[tt];
            }
        }
        if (TV128 && tid >= Tv && tid < 128) alp[(tid & 7) * 16 + (tid >> 3)] = 0.f;
    } else if (A.att_type == GSTTACO_ATT_SMA) {
        if (tid < Tv) {
            const int tt = tid;
            float v = L.pv[tt] * gt_sigmoid(L.sc[tt] + (R.noisy ? L.nz[tt] : 0.f));
            if (tt > 0) v = __builtin_fmaf(L.pv[tt - 1], 1.f - gt_sigmoid(L.sc[tt - 1] + (R.noisy ? L.nz[tt - 1] : 0.f)), v);
            L.al[tt] = v;
            if (TV128) alp[(tt & 7) * 16 + (tt >> 3)] = v;
        } else if (TV128 && tid < 128) alp[(tid & 7) * 16 + (tid >> 3)] = 0.f;
    } else {
        if (tid < Tv) {
            float s = L.sc[tid];
            if (R.noisy) s += L.nz[tid];
            L.sc[tid] = gt_sigmoid(s);
        }
        __syncthreads();
        if (tid < 64) {
            const int per = (Tv + 63) / 64;
            const int t0 = lane * per, t1 = min(Tv, t0 + per);
            float run = 0.f;
            for (int tt = t0; tt < t1; ++tt) run += logf(fminf(fmaxf(1.f - L.sc[tt], 1.17549435e-38f), 1.f));
            float base = front_wave_incl_scan(run, lane) - run;
            for (int tt = t0; tt < t1; ++tt) {
                const float lg = logf(fminf(fmaxf(1.f - L.sc[tt], 1.17549435e-38f), 1.f));
                L.al[tt] = expf(base);
                base += lg;
            }
            run = 0.f;
            for (int tt = t0; tt < t1; ++tt) run += L.pv[tt] / fminf(fmaxf(L.al[tt], 1e-10f), 1.f);
            base = front_wave_incl_scan(run, lane) - run;
            for (int tt = t0; tt < t1; ++tt) {
                base += L.pv[tt] / fminf(fmaxf(L.al[tt], 1e-10f), 1.f);
                L.al[tt] = L.sc[tt] * L.al[tt] * base;
                if (TV128) alp[(tt & 7) * 16 + (tt >> 3)] = L.al[tt];
            }
        }
        if (TV128 && tid >= Tv && tid < 128) alp[(tid & 7) * 16 + (tid >> 3)] = 0.f;
    }
    __syncthreads();
    PD_STAMP(0, 19);
    if (tid < TvFull) A.align[(size_t)b * A.ld_align + (size_t)t * TvFull + tid] = tid < Tv ? L.al[tid] : 0.f;
    // ---- context (Steps.py:160-166): lane = channel, 8 row groups (this thread plays groups tid / 128 and 4 + tid / 128)
    {
        // (beyond 128 tokens the launch path streams the memory through its 128-row tile and sums the row chunks last to first,
        // each chunk into its own pair of partial sums: the same order here, where every row is resident)
        const int ca = tid & (PD_A - 1);
        const int nchunks = TV128 ? 1 : (Tv + 127) >> 7;
#pragma unroll 1
        for (int hh = 0; hh < 2; ++hh) {
            const int cp = hh * 4 + (tid >> 7);
            float cacc = 0.f;
            if (TV128) {
                // the same multiply-adds in the same order (p0: rows cp, cp + 16, ..; p1: rows cp + 8, cp + 24, ..); rows >= Tv add an
                // exact zero (their alignment and their memory row are zero) where the general form skips them
                const float4* ap = reinterpret_cast<const float4*>(alp + cp * 16);
                const float* tl = L.tile + cp * PD_LDV + ca;
                float p0 = 0.f, p1 = 0.f;
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const float4 a4 = ap[k4];
                    p0 = __builtin_fmaf(a4.x, tl[(32 * k4) * PD_LDV], p0);
                    p1 = __builtin_fmaf(a4.y, tl[(32 * k4 + 8) * PD_LDV], p1);
                    p0 = __builtin_fmaf(a4.z, tl[(32 * k4 + 16) * PD_LDV], p0);
                    p1 = __builtin_fmaf(a4.w, tl[(32 * k4 + 24) * PD_LDV], p1);
                }
                cacc += p0 + p1;
            } else
            for (int c = nchunks - 1; c >= 0; --c) {
                const int nr = min(128, Tv - 128 * c);
                const float* alc = L.al + 128 * c;
                const float* tl = L.tile + 128 * c * PD_LDV;
                float p0 = 0.f, p1 = 0.f;
                int tt = cp;
                for (; tt + 8 < nr; tt += 16) {
                    p0 = __builtin_fmaf(alc[tt], tl[tt * PD_LDV + ca], p0);
                    p1 = __builtin_fmaf(alc[tt + 8], tl[(tt + 8) * PD_LDV + ca], p1);
                }
                if (tt < nr) p0 = __builtin_fmaf(alc[tt], tl[tt * PD_LDV + ca], p0);
                cacc += p0 + p1;
            }
            L.red[cp * PD_A + ca] = cacc;
        }
    }
    __syncthreads();
    // (the next step's "previous alignment" -- LSA: the running sum of the alignments, or the last one; every read of pv is behind barriers)
    if (tid < Tv) L.pv[tid] = (LSA && A.lsa_cumulate) ? L.pv[tid] + L.al[tid] : L.al[tid];
    if (tid < 64) {
        if (tid < PD_A / 4) {
            float c[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float z = 0.f;
#pragma unroll
                for (int w = 0; w < 8; ++w) z += L.red[w * PD_A + 4 * tid + e];
                c[e] = z;
            }
            if (MIRROR) {
                uint2 pk;
                pk.x = (uint32_t)gt_bf16_bits(c[0]) | ((uint32_t)gt_bf16_bits(c[1]) << 16);
                pk.y = (uint32_t)gt_bf16_bits(c[2]) | ((uint32_t)gt_bf16_bits(c[3]) << 16);
                pd_st2_sc1(reinterpret_cast<uint2*>(A.xah[par] + gt_blk_off_h(b, PD_P + 4 * tid, MT)), pk);
            } else
            pd_st4_sc1(xa + gt_blk_off(b, PD_P + 4 * tid, MT), make_float4(c[0], c[1], c[2], c[3]));
        }
        pd_drain();
        if (tid == 0) pd_st1_sc1(A.ctl + PD_F_C + b * PD_FS, (uint32_t)t + 1u);
    }
    PD_STAMP(0, 20);
}

// HELPED: the workgroup's layer-1 recurrent half comes from a helper workgroup (the one-group kernel) and is fetched here
// zt: 0, or a per-step opaque zero (the group kernels: nothing of the chain's address arithmetic may be hoisted out of the step loop)
// MIRROR (mixed precision): the prenet output and the context leave as the bf16 mirror (kernels.h gt_blk_off_h) only
template <bool HELPED, bool MIRROR = false, bool TV128 = false, bool LSA = false>
// HELPED: `hraw` returns the four sums around this thread's element of the helper's layer-1 half (step > 0; element tid % 4 is its own)
__device__ __forceinline__ void pd_chain(const PersistDecodeArgs& A, const PdChainLds& L, const PdChainRegs& R, int t, int b, PdShared* sh, float4& hraw, int zt = 0) {
    float4 ra[16], rb[16];
    pd_chain_issue(A, R, t, b, zt, ra, rb);
    pd_chain_rest<HELPED, MIRROR, TV128, LSA>(A, L, R, t, b, sh, hraw, zt, ra, rb);
}

// ====================================================================================================================== roles
// Register discipline: the roles are separate loops, each loading what IT keeps resident, so the allocation is the maximum over
// the roles, not their sum.
__device__ __forceinline__ void pd_run_chain(const PersistDecodeArgs& A, float* smem, PdShared* sh) {
    float* lds = smem;
    PdChainLds L = pd_carve(smem, A.tvp);
    const bool lsa = A.att_type == GSTTACO_ATT_LSA;         // (the launcher: then tvp = 128 and the LDS holds the LSA operands behind the tile)
    if (lsa) L.lpack = L.lfeat + 128 * gt_lsa_pack(PD_A, A.loc_f, A.loc_k).LFS;
    const int tile = blockIdx.x, b = blockIdx.x, tid = threadIdx.x, col = tid & 15;
    const bool live = b < A.B;                                  // (batches below 32: the spare chain workgroups only run their LSTM tile)
    // nothing of this tile's LSTM weights stays resident here: the chain's own operands (all of prenet 1's weights, prefetched)
    // need the registers.  W1x / W2x (24 + 64 KB) are streamed per step into registers the chain has freed -- they arrive while the
    // workgroup waits for the other chains / for h1 anyway -- and the tile's two recurrent halves come from helper workgroups.
    PdW W;
    float c1v = 0.f, c2v = 0.f, p1v = A.b1h[tile * 16 + col], p2v = A.b2h[tile * 16 + col];
    PdChainRegs R{};
    if (live) {
        R.Tv = A.tok_len ? max(1, min(A.Tv, (int)A.tok_len[b])) : A.Tv;         // masked mode (A12): only the first tok_len[b] positions exist
        R.drop = A.drop_rate > 0.f;
        R.hashed = R.drop && A.keep_hash != 0;
        R.noisy = A.sigmoid_noise > 0.f;
        R.seed = R.hashed ? *A.seed_ptr : 0ull;
        R.bias1 = tid < PD_P ? A.b1[tid] : 0.f;
        R.biasq = tid < PD_A ? A.bq[tid] : 0.f;
        R.sbias = lsa ? 0.f : A.score_bias[0];
        // processed memory of utterance b -> LDS, once (rows >= T_v: zeros, as the launch path's bounded descriptor reads them)
        const float4* src = reinterpret_cast<const float4*>(A.pm + (size_t)b * A.Tv * PD_A);
        for (int e = tid; e < A.tvp * PD_A / 4; e += PD_NT) {
            const int row = e / (PD_A / 4), c4 = e % (PD_A / 4);
            *reinterpret_cast<float4*>(L.tile + row * PD_LDV + 4 * c4) = row < R.Tv ? src[(size_t)row * (PD_A / 4) + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < PD_A) L.vs[tid] = lsa ? 0.f : A.av[tid];      // LSA has no attention_v (Layers.py:407)
        if (tid < A.tvp) L.pv[tid] = (tid == 0 && !lsa) ? 1.f : 0.f;     // one-hot(0) initial alignment (Steps.py:201-206); LSA: the zero state (Layers.py:356)
        if (lsa) {
            const int n4 = gt_lsa_pack(PD_A, A.loc_f, A.loc_k).total >> 2;
            for (int i = tid; i < n4; i += PD_NT) reinterpret_cast<float4*>(L.lpack)[i] = reinterpret_cast<const float4*>(A.loc_pack)[i];
        }
    }
    __syncthreads();
    for (int t = 0; t < A.steps; ++t) {
        if (sh->abort) return;          // (the one abort check of the step: PD_PHASE_ABORT)
        PD_STAMP(0, 0);
        // (diagnostic: when this workgroup's steps 0, 1, 2, 8, 64, half, 3/4 and last begin -- slots 24..31 of the chain role)
        if (A.dbg && threadIdx.x == 0 && blockIdx.x == 0) {
            const int ts[8] = {0, 1, 2, 8, 64, A.steps >> 1, (3 * A.steps) >> 2, A.steps - 1};
#pragma unroll
            for (int k = 0; k < 8; ++k) if (t == ts[k]) A.dbg[24 + k] = __builtin_amdgcn_s_memrealtime();
        }
        if (live) {
            PD_ZT(zt);
            float4 hraw;
            if (lsa) pd_chain<true, false, true, true>(A, L, R, t, b, sh, hraw, zt);
            else if (A.tvp == 128) pd_chain<true, false, true>(A, L, R, t, b, sh, hraw, zt);
            else pd_chain<true>(A, L, R, t, b, sh, hraw, zt);
            PD_PHASE_ABORT(sh);
            if (t > 0) { const int e = tid & 3; p1v = e == 0 ? hraw.x : e == 1 ? hraw.y : e == 2 ? hraw.z : hraw.w; }
        } else if (t > 0) {                                     // no chain to hide it behind: fetch the layer-1 half directly
            pd_wait_flags(A, A.ctl + PD_F_H + tile * 32, 1, (uint32_t)t, sh);
            PD_PHASE_ABORT(sh);
            const auto rh = gt_rsrc(A.hpart, 2u * 32u * 512u * 4u);
            const float4 d1 = gt_bload4_sc1(rh, (uint32_t)((tile * 512 + (tid & ~3)) * 4), 0u);
            const int e = tid & 3;
            p1v = e == 0 ? d1.x : e == 1 ? d1.y : e == 2 ? d1.z : d1.w;
        }
        pd_cell1(A, W, t, tile, lds, c1v, p1v, sh, false, true, 0, PD_X2_EARLY != 0);
        PD_PHASE_ABORT(sh);
        float unused = 0.f;
        // (the layer-2 half from its helper: flag and sums ride with the h1 arrivals' poll and fragments)
        pd_cell2(A, W, t, tile, lds, c2v, p2v, unused, false, sh, -1, false, PD_X2_EARLY == 0, 0, t > 0 ? 32 + tile : -1, true);
        PD_PHASE_ABORT(sh);
    }
}

__device__ __forceinline__ void pd_run_proj(const PersistDecodeArgs& A, float* lds, PdShared* sh) {
    const int tile = blockIdx.x, col = threadIdx.x & 15;
    const int pi = tile - PD_UTT, ptile = pi % A.pj_tiles, pmt = pi / A.pj_tiles;
    // resident: the input halves of both cells and the projection tile (80 registers); the recurrent-half tiles run behind the
    // projection, off the critical path, and are streamed
    PdW W;
    pd_load_tile<3>(A.w1x, tile, W.x1); pd_load_tile<8>(A.w2x, tile, W.x2);
    float4 wpj[9];
    pd_load_tile<9>(A.wp, ptile, wpj);
    float c1v = 0.f, c2v = 0.f, p1v = A.b1h[tile * 16 + col], p2v = A.b2h[tile * 16 + col];
    for (int t = 0; t < A.steps; ++t) {
        if (sh->abort) return;          // (the one abort check of the step: PD_PHASE_ABORT)
        const int par = t & 1;
        PD_STAMP(1, 0);
        pd_cell1(A, W, t, tile, lds, c1v, p1v, sh, true, false, 1);
        PD_PHASE_ABORT(sh);
        float unused = 0.f;
        pd_cell2(A, W, t, tile, lds, c2v, p2v, unused, false, sh, -1, false, false, 1);
        PD_PHASE_ABORT(sh);
        pd_proj(A, wpj, t, ptile, pmt, lds, sh);
        PD_PHASE_ABORT(sh);
        if (t + 1 == A.steps) break;
        pd_load_tile<8>(A.w1h, tile, W.h1);
        PD_PIN();
        p1v = pd_rec1_mem(A, A.h1[par], W.h1, tile, lds);       // for step t + 1 (h1_t re-read: off the critical path)
        pd_load_tile<8>(A.w2h, tile, W.h2);
        PD_PIN();
        p2v = pd_rec2(A, t, A.h2[par], W.h2, tile, -1, lds);
    }
}

// HELP: 0 = plain; 1 / 2 = also the layer-1 / layer-2 recurrent half of chain workgroup `help_tile`.  A helper keeps three of its own
// four weight tiles resident and streams the fourth (used off the critical path), so that the extra tile's fragments fit.
template <int HELP>
__device__ __forceinline__ void pd_run_plain(const PersistDecodeArgs& A, float* lds, PdShared* sh, int help_tile) {
    const int tile = blockIdx.x, col = threadIdx.x & 15;
    PdW W;
    pd_load_tile<3>(A.w1x, tile, W.x1); pd_load_tile<8>(A.w2x, tile, W.x2);
    if (HELP != 2) pd_load_tile<8>(A.w1h, tile, W.h1);
    if (HELP != 1) pd_load_tile<8>(A.w2h, tile, W.h2);
    float c1v = 0.f, c2v = 0.f, p1v = A.b1h[tile * 16 + col], p2v = A.b2h[tile * 16 + col];
    for (int t = 0; t < A.steps; ++t) {
        if (sh->abort) return;          // (the one abort check of the step: PD_PHASE_ABORT)
        const int par = t & 1;
        PD_STAMP(2, 0);
        pd_cell1(A, W, t, tile, lds, c1v, p1v, sh, true);
        PD_PHASE_ABORT(sh);
        pd_cell2(A, W, t, tile, lds, c2v, p2v, p1v, true, sh, HELP == 1 ? help_tile : -1, HELP == 2, false);
        PD_STAMP(2, 12);
        PD_PHASE_ABORT(sh);
        if (t + 1 == A.steps) break;
        if (HELP == 1) pd_load_tile<8>(A.w2h, tile, W.h2);      // (streamed: arrives during the wait)
        pd_wait_count(A, A.ctl + PD_CNT4, PD_WANT(A, t), sh);
        PD_PHASE_ABORT(sh);
        PD_STAMP(2, 13);
        p2v = pd_rec2(A, t, A.h2[par], W.h2, tile, HELP == 2 ? help_tile : -1, lds);        // for step t + 1
        PD_STAMP(2, 14);
    }
}

__global__ __launch_bounds__(PD_NT) void gt_persist_decode_kernel(PersistDecodeArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ PdShared sh;
    if (threadIdx.x == 0) sh.abort = 0;
    // (diagnostic: when workgroups 0 and 255 enter the kernel -- slots 9 / 10 of the chain role; against the first step's start in slot 24)
    if (A.dbg && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == PD_NWG - 1)) A.dbg[blockIdx.x == 0 ? 9 : 10] = __builtin_amdgcn_s_memrealtime();
    __syncthreads();
    const int tile = blockIdx.x;
    const int n_pj = A.pj_tiles * A.MT;
#ifndef PD_ONLY
#define PD_ONLY -1          // (register-budget diagnosis: compile one role alone)
#endif
    if (tile < PD_UTT) { if (PD_ONLY < 0 || PD_ONLY == 0) pd_run_chain(A, smem, &sh); }
    else if (tile < PD_UTT + n_pj) { if (PD_ONLY < 0 || PD_ONLY == 1) pd_run_proj(A, smem, &sh); }
    else {
        const int hidx = tile - (PD_UTT + n_pj);            // 64 helpers: even = layer 1, odd = layer 2 of chain tile hidx / 2
        if (hidx >= PD_HELP) { if (PD_ONLY < 0 || PD_ONLY == 2) pd_run_plain<0>(A, smem, &sh, -1); }
        else if ((hidx & 1) == 0) { if (PD_ONLY < 0 || PD_ONLY == 3) pd_run_plain<1>(A, smem, &sh, hidx >> 1); }
        else if (PD_ONLY < 0 || PD_ONLY == 4) pd_run_plain<2>(A, smem, &sh, hidx >> 1);
    }
}


#include "persist_groups.h"      // the group kernels (33..128 rows in fp32)
#include "persist_bf16.h"        // the bf16 kernel (mixed precision, <= 64 rows)

// z0 granules of step 0: the first frame is zero (Taco2.py:162-165), so prenet 0's pre-activations are its bias; + the control words
__global__ void gt_persist_decode_init_kernel(uint2* z0g, const float* b0, uint32_t* ctl, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < PD_CTL_WORDS) ctl[i] = 0u;
    if (i < B * PD_P) {
        uint2 g;
        g.x = __builtin_bit_cast(uint32_t, b0[i % PD_P]); g.y = 0u;
        z0g[i] = g;
    }
}

}  // namespace

size_t gt_persist_decode_ctl_words() { return PD_CTL_WORDS; }

int gt_persist_decode_max_batch() { return PD_BMAX; }

bool gt_persist_decode_lsa_fits(int B, int Tv, int loc_f, int loc_k) {
    return B <= 32 && Tv <= 128 && loc_f >= 1 && loc_k >= 1 && pd_lds_floats(128) + pd_lsa_floats(loc_f, loc_k) <= pd_lds_floats(PD_TVMAX, true);   // (the opted-in LDS size)
}

// `split16`: a batch of 17..32 rows as two groups of 16 (experiment; the default is the one-group kernel with its helper workgroups)
// the bf16 kernel's LDS: slabs of 64 rows; the chain workgroups' 16 (or, beside more than 128 tokens, 8) + their chain state
__host__ __device__ constexpr int pdh_chain_floats(int tvp, int nslab) { return nslab * PDH_SLAB + 3 * PD_P + 2 * PD_A + 4 * tvp + tvp * PD_LDV; }
__host__ __device__ constexpr int pdh_chain_slabs(int) { return 8; }      // (the chain workgroups' recurrent halves are the helpers': cell reductions only)
__host__ __device__ constexpr int pdh_lds_floats(int tvp) {
    return pdh_chain_floats(tvp, pdh_chain_slabs(tvp)) > 16 * PDH_SLAB ? pdh_chain_floats(tvp, pdh_chain_slabs(tvp)) : 16 * PDH_SLAB;
}
constexpr int PDH_TVMAX = 192;
static_assert(pdh_lds_floats(PDH_TVMAX) * 4 + 64 <= 160 * 1024, "the bf16 kernel's chain state must fit a CU's LDS");

bool gt_persist_decode_supported(int mel, int r, int P0, int P1, int A, int H1, int H2, int B, int Tv, int pj_tiles, int pj_nkb, int slots, int split16, int bf16) {
    (void)mel; (void)r;
    if (!(P0 == PD_P && P1 == PD_P && A == PD_A && H1 == PD_H && H2 == PD_H && B >= 1 && Tv >= 1 && Tv <= PD_TVMAX && pj_nkb == PD_KBPJ && pj_tiles >= 1 &&
          slots >= PD_NWG))
        return false;
    if (bf16) return B <= PDH_BMAX && Tv <= PDH_TVMAX && 2 * B + pj_tiles * ((B + 15) / 16) <= PD_NWG;     // (a chain and a helper per utterance)
    if (B <= 32 && !(split16 && B > 16)) return PD_UTT + pj_tiles * ((B + 15) / 16) + PD_HELP <= PD_NWG;
    return B <= PD_BMAX && B + pj_tiles * (B <= 32 ? 1 : 2) <= PD_NWG;
}

namespace {
typedef void (*PdKernel)(PersistDecodeArgs);
// which kernel a batch runs on: the one-group kernel up to 32 rows, groups of 32 rows above (2 up to 64 rows, else 4)
PdKernel pd_kernel_for(int B, int split16, int* G, int* mtg) {
    if (B <= 32 && !(split16 && B > 16)) { *G = 1; *mtg = 2; return gt_persist_decode_kernel; }
    if (B <= 32) { *G = 2; *mtg = 1; return gt_persist_decode_g_kernel<2, 1>; }
    *mtg = 2; *G = (B + 31) / 32;
    return B <= 64 ? gt_persist_decode_g_kernel<2, 2> : gt_persist_decode_g_kernel<4, 2>;
}
const PdKernel kPdKernels[] = {gt_persist_decode_kernel, gt_persist_decode_g_kernel<2, 1>, gt_persist_decode_g_kernel<2, 2>, gt_persist_decode_g_kernel<4, 2>,
                               gt_persist_decode_h_kernel};
}  // namespace

hipError_t gt_persist_decode_init() {
    for (PdKernel k : kPdKernels) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, pd_lds_floats(PD_TVMAX, true) * 4);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// the smallest occupancy over the kernels at the largest LDS size: every one of them needs its 256 workgroups resident
int gt_persist_decode_blocks_per_cu() {
    int worst = 1 << 30;
    for (PdKernel k : kPdKernels) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(k), PD_NT, (size_t)pd_lds_floats(PD_TVMAX, true) * 4) != hipSuccess) return 0;
        worst = n < worst ? n : worst;
    }
    return worst;
}

hipError_t gt_launch_persist_decode(const PersistDecodeArgs& a_in, const float* b0, int split16, hipStream_t stream) {
    PersistDecodeArgs a = a_in;
    if (a.bf16) {           // mixed precision: one group of up to 64 rows
        a.G = 1;
        a.tvp = (a.Tv + 63) / 64 * 64;
        a.n_chain = a.B;
        a.twopass = 0;          // (the workgroups that sum recurrent halves have 16 slabs: no chain state beside them)
        const int n = PD_CTL_WORDS > a.B * PD_P ? PD_CTL_WORDS : a.B * PD_P;
        hipLaunchKernelGGL(gt_persist_decode_init_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a.z0g, b0, a.ctl, a.B);
        hipLaunchKernelGGL(gt_persist_decode_h_kernel, dim3(PD_NWG), dim3(PD_NT), (size_t)pdh_lds_floats(a.tvp) * 4, stream, a);
        return hipGetLastError();
    }
    int mtg = 2;
    const bool lsa = a.att_type == GSTTACO_ATT_LSA;
    if (lsa) split16 = 0;
    const PdKernel k = pd_kernel_for(a.B, split16, &a.G, &mtg);
    a.tvp = lsa ? 128 : (a.Tv + 63) / 64 * 64;       // (LSA: the 128-token chain only; gt_persist_decode_lsa_fits)
    const bool gk = k != gt_persist_decode_kernel;
    if (lsa && (gk || !gt_persist_decode_lsa_fits(a.B, a.Tv, a.loc_f, a.loc_k))) return hipErrorInvalidValue;
    a.n_chain = gk ? a.B : PD_UTT;
    a.twopass = pd_chain_slabs(a.tvp, gk) == 8 ? 1 : 0;
    const int n = PD_CTL_WORDS > a.B * PD_P ? PD_CTL_WORDS : a.B * PD_P;
    hipLaunchKernelGGL(gt_persist_decode_init_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, a.z0g, b0, a.ctl, a.B);
    hipLaunchKernelGGL(k, dim3(PD_NWG), dim3(PD_NT), (size_t)(pd_lds_floats(a.tvp, gk) + (lsa ? pd_lsa_floats(a.loc_f, a.loc_k) : 0)) * 4, stream, a);
    return hipGetLastError();
}
