// Device code of the fused decoder-step front end (see dec_front.hip for the design notes): the GEMV helpers, the worker
// workgroups, the lean utterance path (front_lean.h) and the general utterance kernel.  Included by dec_front.hip (BMA / SMA)
// and dec_front_lsa.hip (the step-wise location-sensitive extension: the same kernel with LSA = true).
#pragma once
#include "skinny_body.h"
#include "lean_body.h"
#include "chain_common.h"
#include "../../include/gsttaco.h"
#include <stdlib.h>

#define FT 1024            // threads per workgroup
#define WT 2               // recurrent-half tiles per worker job
#define FMAXR 16           // max weight rows (float4 loads) per lane per GEMV phase (two register blocks of 8)

struct GemvPlan {
    int ncg, kparts, rows, cg, kp;
};

__device__ __forceinline__ GemvPlan make_plan(int K, int N, int tid) {
    GemvPlan p;
    p.ncg = N >> 2;
    p.kparts = FT / p.ncg;
    if (p.kparts > K) p.kparts = K;
    p.rows = (K + p.kparts - 1) / p.kparts;
    p.cg = tid % p.ncg;
    p.kp = tid / p.ncg;
    return p;
}

// loads rows [i0, i0+MAXR) of this lane's k range.  EXACT: the plan divides evenly (every lane has exactly i0+MAXR or more
// rows, all inside K) -- true for the prenet-1 and query layers at the reference's dimensions -- so the loads carry no
// predicates: with them each load costs ~30 instructions of exec-mask bookkeeping, ~1 us over the prologue's 24 loads.
template <int MAXR, bool EXACT = false>
__device__ __forceinline__ void gemv_load(const float* __restrict__ W, int K, int N, const GemvPlan& p, int i0, float4 (&r)[MAXR],
                                          uint32_t bits = 0xFFFFFFFFu) {
    if (EXACT) {
        // uniform (SGPR) base per row + ONE per-lane 32-bit byte offset: no per-load address registers.
        // `bits`: bit i clear <=> row i0 + i multiplies an exact zero (dropped by the prenet's dropout) and is not requested
        const uint32_t off = (uint32_t)(((p.kp * p.rows + i0) * N + p.cg * 4) * 4);
#pragma unroll
        for (int i = 0; i < MAXR; ++i) {
            r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((bits >> i) & 1u) r[i] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(W) + (size_t)i * N * 4 + off);
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int k = p.kp * p.rows + i0 + i;
        r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i0 + i < p.rows && p.kp < p.kparts && k < K)
            r[i] = *reinterpret_cast<const float4*>(W + (size_t)k * N + p.cg * 4);
    }
}

template <int MAXR, bool EXACT = false>
__device__ __forceinline__ void gemv_acc(const float* xs, int K, const GemvPlan& p, int i0, const float4 (&r)[MAXR], float4& acc) {
#pragma unroll
    for (int i = 0; i < MAXR; ++i) {
        const int k = p.kp * p.rows + i0 + i;
        if (EXACT || (i0 + i < p.rows && p.kp < p.kparts && k < K)) {
            const float x = xs[k];
            gt_fma4(acc, x, r[i]);
        }
    }
}

// partial[kp][N] <- this lane's partial sums; caller syncs and reduces
__device__ __forceinline__ void gemv_store(const GemvPlan& p, int N, const float4& acc, float* partial) {
    if (p.kp < p.kparts) *reinterpret_cast<float4*>(partial + (size_t)p.kp * N + p.cg * 4) = acc;
}

// ---- worker workgroup (blockIdx.x >= B): recurrent half of an LSTM gate GEMM (see DecFrontArgs::rec).
// Jobs are groups of up to WT adjacent tiles of one layer sharing one pass over that layer's state (skinny_body.h).
template <int LEAN>
__device__ __forceinline__ void front_worker(const DecFrontArgs& P, float* smem, const int slot) {
    const int wt = P.worker_tiles == 1 ? 1 : WT;
    const int j0 = (P.rec_end[0] - P.rec_begin[0] + wt - 1) / wt;
    const int total = j0 + (P.rec_end[1] - P.rec_begin[1] + wt - 1) / wt;
    const int mchunks = (P.B + 31) / 32;
    const int wslot = slot == 0 ? 8 : ((int)blockIdx.x == (int)gridDim.x - 1 ? 10 : -1);   // diagnostics
    if (P.dbg && wslot >= 0 && threadIdx.x == 0) P.dbg[wslot] = __builtin_amdgcn_s_memrealtime();
    if (LEAN != 0 && mchunks > 1) {
        // Batches above 32 rows: a job's weights stay in registers over its chunks (lean_body.h gt_lean_partial_mc).  Schedule
        // (DecFrontArgs::sched_*): the pure workers take whole jobs [0, pf) round-robin, then pieces of `e` chunks of jobs
        // [pf, pf + ne); utterance workgroups that have finished their chain take pieces of `y` chunks of jobs [pf + ne, total).
        auto run = [&](const int job, const int c0, const int c1) {
            const int layer = job < j0 ? 0 : 1;
            const int tile = P.rec_begin[layer] + (layer == 0 ? job : job - j0) * wt;
            const int ntile = min(wt, P.rec_end[layer] - tile);
            if (LEAN == 2) gt_lean_partial_mc<FT / 64, 2, WT, true>(P.lrec[layer], tile, ntile, c0, c1, smem);
            else gt_lean_partial_mc<FT / 64, 4, WT>(P.lrec[layer], tile, ntile, c0, c1, smem);
            __syncthreads();
        };
        auto pieces = [&](const int first, const int stride, const int base, const int njobs, const int z) {
            if (z <= 0 || njobs <= 0) return;
            const int npp = (mchunks + z - 1) / z;
            for (int q = first; q < njobs * npp; q += stride) {
                const int c0 = (q % npp) * z;
                run(base + q / npp, c0, min(mchunks, c0 + z));
            }
        };
        if (slot < P.n_workers) {
            for (int job = slot; job < P.sched_pf; job += P.n_workers) run(job, 0, mchunks);
            pieces(slot, P.n_workers, P.sched_pf, P.sched_ne, P.sched_e);
        } else {
            pieces(slot - P.n_workers, P.utt_jobs, P.sched_pf + P.sched_ne, total - P.sched_pf - P.sched_ne, P.sched_y);
        }
    } else {
        for (int job = slot; job < total; job += P.n_workers) {
            const int layer = job < j0 ? 0 : 1;
            const int tile = P.rec_begin[layer] + (layer == 0 ? job : job - j0) * wt;
            const int ntile = min(wt, P.rec_end[layer] - tile);
            for (int mc = 0; mc < mchunks; ++mc) {
                if (LEAN == 2) gt_lean_partial<FT / 64, 2, WT, true>(P.lrec[layer], tile, ntile, mc, smem);
                else if (LEAN == 1) gt_lean_partial<FT / 64, 4, WT>(P.lrec[layer], tile, ntile, mc, smem);
                else gt_skinny_partial_multi<FT / 64, WT, true>(P.rec[layer], tile, ntile, mc, smem);
                __syncthreads();
            }
        }
    }
    if (P.dbg && wslot >= 0 && threadIdx.x == 0) P.dbg[wslot + 1] = __builtin_amdgcn_s_memrealtime();
}

#include "front_lean.h"

// Lean utterance path (front_lean.h) + the same workers: the decode loop's launch from step 1 on at the reference's sizes.
template <int L, int NP, int LEAN>
__global__ __launch_bounds__(FT) void gt_dec_front_lean_kernel(DecFrontArgs P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x >= P.B) {
        front_worker<LEAN>(P, smem, (int)blockIdx.x - P.B);
        return;
    }
    gt_front_lean<L, NP>(P, smem, (int)blockIdx.x, blockIdx.x == 0);
    if ((int)blockIdx.x < P.utt_jobs) {     // large batches: this workgroup's CU takes recurrent-half jobs once its utterance is done
        __syncthreads();
        front_worker<LEAN>(P, smem, P.n_workers + (int)blockIdx.x);
    }
}

// Z0: the projection launch of the previous step already produced this step's prenet-0 pre-activations (the projection and
// the first prenet Dense are both linear: frame.W0 + b0 = [h2|ctx].(Wp_last.W0) + (bp_last.W0 + b0), DecFrontArgs::z0), so the
// chain starts at prenet 1 and the query weights are requested at kernel start in place of prenet 0's.
// sum helper of the LSA score epilogue: the value of another lane of the same 16-lane row (DPP control word CTRL), no LDS round trip
template <int CTRL>
__device__ __forceinline__ float front_dpp(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}

//
// LSA (dec_front_lsa.hip): the step-wise location-sensitive extension (reference Modules/Attention/Layers.py:345-424) in
// the same kernel.  Per utterance and step it is two small GEMMs -- location features  lfeat[t][f] = cb[f] + sum_j state[t+j-pad].cw[j][f]
// (a Toeplitz product, K = kernel size) and  loc[t][a] = db[a] + bias[a] + sum_f lfeat[t][f].dw[f][a] -- 0.65 M multiply-adds at
// 128 positions x 32 filters x 31 taps x 128 channels, which as scalar loops over LDS operands took 100+ us per step
// (attention.hip, tools/lsa_time.py).  Here both run on v_mfma_f32_16x16x4_f32 (exact fp32, a k-ordered fmaf chain: the same sums
// in the same order as the scalar loops), 16 x 16 output tiles dealt round-robin to the 16 waves, operands in LDS (weights staged
// with the small operands at kernel start), and the tanh / row-sum epilogue works on the accumulator tile in place.
template <int L, int NP, bool Z0, int LEAN, bool EXACT, bool LSA = false>
__global__ __launch_bounds__(FT) void gt_dec_front_kernel(DecFrontArgs P) {
    constexpr int A = 4 * L * NP;
    constexpr int ROWS = FT / L;            // memory rows per pass (one LDS tile)
    constexpr int LD = A + 4;               // padded tile row: conflict-free 16-byte row writes and 4-byte column reads
    constexpr int CPARTS = FT / A;          // row groups of the context pass
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x >= P.B) {
        front_worker<LEAN>(P, smem, (int)blockIdx.x - P.B);
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int b = blockIdx.x;
    GT_STAMP(P.dbg, 12);
    const int mel = P.mel, P0 = P.P0, P1 = P.P1, TvFull = P.Tv;
    // the seed is requested first and waited for only where the keep decisions are derived (below), so its latency
    // overlaps the address arithmetic and the small-operand requests
    // (read through the constant address space: a scalar load the compiler schedules itself -- the seed was written by an
    // earlier kernel of the stream, and a plain load of it compiles to a vector load + readfirstlane, waited for on the spot)
    const bool need_seed = P.drop_rate > 0.f && (!P.mask0 || !P.mask1);
    uint64_t kseed = 0;
    if (need_seed) kseed = *(const __attribute__((address_space(4))) uint64_t*)P.seed_ptr;
    // masked mode (A12): only the first tok_len[b] memory positions exist for this utterance
    const int Tv = P.tok_len ? max(1, min(TvFull, P.tok_len[b])) : TvFull;

    // LDS carve (floats): xs | y0 | y1 | q | v | score | prev | align | red[FT] | partial[4 FT] | tile[ROWS][LD]
    int mx = mel > P0 ? mel : P0; if (P1 > mx) mx = P1;
    float* xs = smem;
    float* y0 = xs + ((mx + 3) & ~3);
    float* y1 = y0 + P0;
    float* qs = y1 + P1;
    float* vs = qs + A;
    float* sc = vs + A;
    float* pv = sc + ((TvFull + 3) & ~3);
    float* al = pv + ((TvFull + 3) & ~3);
    float* red = al + ((TvFull + 3) & ~3);
    // staged at kernel start so that no phase of the dependent chain issues a global load of its own: loads return in
    // issue order, so waiting for a late small load also waits for every weight prefetch issued before it (~2 us)
    float* sb0 = red + FT;                     // prenet0 bias [P0]
    float* sb1 = sb0 + P0;                     // prenet1 bias [P1]
    float* sbq = sb1 + P1;                     // query bias [A]
    float* sk0 = sbq + A;                      // prenet0 keep * scale [P0] (injected mask or Philox)
    float* sk1 = sk0 + P0;                     // prenet1 keep * scale [P1]
    float* snz = sk1 + P1;                     // sigmoid noise [Tv] (injected or Philox)
    float* partial = snz + ((TvFull + 3) & ~3);
    float* tile = partial + 4 * FT;
    // LSA operands behind the tile (front_lds_bytes): location features of the tile's rows, the two weight matrices (zero-padded to
    // the MFMA k / n granules, row strides chosen so that fragment reads spread over the banks) and the two bias vectors
    const int LF = LSA ? P.loc_f : 0, LK = LSA ? P.loc_k : 0;
    const LsaPack lp = gt_lsa_pack(A, LF, LK);
    const int LFc = lp.LFc, LFp = lp.LFp, LKp = lp.LKp, LFS = lp.LFS, LDWS = lp.LDWS, LCS = lp.LCS;
    float* lfeat = tile + ROWS * LD;            // [ROWS][LFS]
    float* ldw = lfeat + ROWS * LFS;            // the weight image (kernels.h LsaPack): dw | cw | cb | ab
    float* lcw = ldw + lp.off_cw;
    float* lcbs = ldw + lp.off_cb;
    float* labias = ldw + lp.off_ab;

    // ---- issue every independent global load first
    const int row = tid / L, li = tid % L;
    const float* pm = P.pm + (size_t)b * TvFull * A;
    auto load_rows = [&](float4 (&v)[NP], int c) {
        const int t = c * ROWS + row;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t < Tv) v[j] = *reinterpret_cast<const float4*>(pm + (size_t)t * A + 4 * (li + L * j));
        }
    };
    auto store_rows = [&](const float4 (&v)[NP]) {
#pragma unroll
        for (int j = 0; j < NP; ++j) *reinterpret_cast<float4*>(tile + row * LD + 4 * (li + L * j)) = v[j];
    };
    // small operands first: loads return in issue order, so their LDS writes below wait only for themselves.  They are
    // BRANCH-FREE (clamped index + select): inside an exec-masked block the compiler sinks the first use (the ReLU of the
    // z0 row) next to the load and waits for it there -- a full memory latency before any other load is even requested.
    // The Philox seed is read only by the branches that draw random numbers, for the same reason.
    const float x_raw = Z0 ? P.z0[(size_t)b * P0 + min(tid, P0 - 1)] : P.frame[(size_t)b * P.ldframe + min(tid, mel - 1)];
    const float in_x = tid < (Z0 ? P0 : mel) ? x_raw : 0.f;                         // mel <= FT (checked on the host)
    const float v_raw = LSA ? 0.f : P.v[min(tid, A - 1)];          // LSA has no attention_v (Layers.py:407)
    const float in_v = tid < A ? v_raw : 0.f;
    float in_p = tid == 0 ? 1.f : 0.f;
    if (P.prev) in_p = P.prev[(size_t)b * P.ldprev + min(tid, Tv - 1)];
    const float sbias = LSA ? 0.f : P.score_bias[0];
    // small vectors staged into LDS (see the carve): requested before the big loads, written after they are issued
    float t_b0 = 0.f, t_k0 = 1.f, t_k1 = 1.f, t_nz = 0.f;
    if (!Z0) t_b0 = P.b0[min(tid, P0 - 1)];
    const float t_b1 = P.b1[min(tid, P1 - 1)];
    const float t_bq = P.bq[min(tid, A - 1)];
    if (P.drop_rate > 0.f && P.mask0) t_k0 = P.mask0[(size_t)b * P0 + min(tid, P0 - 1)];
    if (P.drop_rate > 0.f && P.mask1) t_k1 = P.mask1[(size_t)b * P1 + min(tid, P1 - 1)];
    if (P.sigmoid_noise > 0.f && P.noise) t_nz = P.noise[(size_t)b * P.ldnoise + min(tid, Tv - 1)];
    const GemvPlan g0 = make_plan(mel, P0, tid);
    const GemvPlan g1 = make_plan(P0, P1, tid);
    const GemvPlan g2 = make_plan(P1, A, tid);
    // Request order = need order (loads return in issue order): with the prenet-0 pre-activations already there (Z0) the
    // chain starts at prenet 1, so ALL of W1 goes first, then the query weights; the processed-memory rows, needed three
    // phases later, are requested once prenet 1 has consumed W1 and take its registers (128 VGPRs at 16 waves/CU: three
    // 8-row weight blocks is what fits).  Without Z0 (step 0) prenet 0 comes first and the rows ride along early.
    float4 v0[NP];
    float4 r0a[8], r1a[8], r1b[8], r2a[8];
    // Throughput mode at rate 0.5 (keep_hash): the keep decisions are words of a counter hash of the seed (gt_keep_word), so
    // each wave knows after one scalar load and ~25 scalar instructions which of ITS weight rows meet an exact zero: wave w
    // owns prenet-1 rows 16w..16w+15 (mask 0) and query rows 16w..16w+15 (mask 1, 8 per lane half).
    GT_STAMP(P.dbg, 13);
    uint32_t rb1 = 0xFFFFu, rbq = 0xFFu;
    if (need_seed) {
        if (EXACT && Z0 && P.keep_hash) {
            const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
            rb1 = (gt_keep_word(kseed, P.rng_step, 0u, (uint32_t)b, wave >> 1) >> ((wave & 1u) * 16u)) & 0xFFFFu;
            const uint32_t q16 = (gt_keep_word(kseed, P.rng_step, 1u, (uint32_t)b, wave >> 1) >> ((wave & 1u) * 16u)) & 0xFFFFu;
            rbq = (q16 >> ((lane >> 5) * 8)) & 0xFFu;
        }
    }
    GT_STAMP(P.dbg, 14);
    if (Z0) {
        gemv_load<8, EXACT>(P.w1, P0, P1, g1, 0, r1a, rb1);
        gemv_load<8, EXACT>(P.w1, P0, P1, g1, 8, r1b, rb1 >> 8);
    } else {
        load_rows(v0, 0);                               // processed-memory rows of chunk 0 (64 KiB at 128x128)
        gemv_load<8>(P.w0, mel, P0, g0, 0, r0a);
        gemv_load<8, EXACT>(P.w1, P0, P1, g1, 0, r1a);
    }

    GT_STAMP(P.dbg, 15);
    // dropout keep-scales and sigmoid noise (Philox, ~100 VALU ops each) while the first loads are in flight
    if (P.drop_rate > 0.f) {
        if (!P.mask0 && tid < P0) t_k0 = gt_drop_keep(kseed, P.rng_step, 0u, (uint32_t)b, (uint32_t)tid, (uint32_t)P0, P.drop_rate);
        if (!P.mask1 && tid < P1) t_k1 = gt_drop_keep(kseed, P.rng_step, 1u, (uint32_t)b, (uint32_t)tid, (uint32_t)P1, P.drop_rate);
        t_k0 *= P.drop_scale; t_k1 *= P.drop_scale;
    }
    if (P.sigmoid_noise > 0.f && !P.noise && tid < Tv) {
        const Philox4 r = gt_philox(*P.seed_ptr, (uint32_t)(b * TvFull + tid), P.rng_step, 0u, GT_RNG_NOISE);
        t_nz = gt_normal(r.x, r.y);
    }
    if (tid < P0) sk0[tid] = t_k0;
    if (tid < P1) sk1[tid] = t_k1;
    if (tid < Tv) snz[tid] = P.sigmoid_noise * t_nz;
    // (sizes beyond one pass of the workgroup: never at the reference's dimensions)
    for (int c = tid + FT; c < P0; c += FT) {
        if (!Z0) sb0[c] = P.b0[c];
        float keep = 1.f;
        if (P.drop_rate > 0.f) {
            keep = P.mask0 ? P.mask0[(size_t)b * P0 + c]
                           : gt_drop_keep(kseed, P.rng_step, 0u, (uint32_t)b, (uint32_t)c, (uint32_t)P0, P.drop_rate);
            keep *= P.drop_scale;
        }
        sk0[c] = keep;
    }
    for (int c = tid + FT; c < P1; c += FT) {
        sb1[c] = P.b1[c];
        float keep = 1.f;
        if (P.drop_rate > 0.f) {
            keep = P.mask1 ? P.mask1[(size_t)b * P1 + c]
                           : gt_drop_keep(kseed, P.rng_step, 1u, (uint32_t)b, (uint32_t)c, (uint32_t)P1, P.drop_rate);
            keep *= P.drop_scale;
        }
        sk1[c] = keep;
    }
    for (int c = tid + FT; c < A; c += FT) sbq[c] = P.bq[c];
    for (int t = tid + FT; t < Tv; t += FT) {
        float nz = 0.f;
        if (P.sigmoid_noise > 0.f) {
            if (P.noise) nz = P.noise[(size_t)b * P.ldnoise + t];
            else {
                const Philox4 r = gt_philox(*P.seed_ptr, (uint32_t)(b * TvFull + t), P.rng_step, 0u, GT_RNG_NOISE);
                nz = gt_normal(r.x, r.y);
            }
        }
        snz[t] = P.sigmoid_noise * nz;
    }
    if (!Z0 && tid < P0) sb0[tid] = t_b0;
    if (tid < P1) sb1[tid] = t_b1;
    if (tid < A) sbq[tid] = t_bq;
    if (!Z0 && tid < mel) xs[tid] = in_x;
    if (Z0) {
        if (tid < P0) y0[tid] = fmaxf(in_x, 0.f) * t_k0;
        for (int c = tid + FT; c < P0; c += FT) y0[c] = fmaxf(P.z0[(size_t)b * P0 + c], 0.f) * sk0[c];
    }
    if (tid < A) vs[tid] = in_v;
    if (tid < Tv) pv[tid] = in_p;
    for (int t = tid + FT; t < Tv; t += FT) pv[t] = P.prev ? P.prev[(size_t)b * P.ldprev + t] : 0.f;
    if (LSA)        // (requested behind the prenet-1 weights, which the first GEMV waits for anyway)
        for (int i = tid; i < (lp.total >> 2); i += FT) reinterpret_cast<float4*>(ldw)[i] = reinterpret_cast<const float4*>(P.loc_pack)[i];
    GT_STAMP(P.dbg, 0);
    if (!Z0) {
        store_rows(v0);                                 // the rows were requested first, so they are back first
        asm volatile("" ::: "memory");                    // keep the next request BEHIND the tile write (register budget)
        gemv_load<8, EXACT>(P.w1, P0, P1, g1, 8, r1b);         // second half of prenet1 takes the registers the rows freed
    }
    __syncthreads();
    GT_STAMP(P.dbg, 1);

    // ---- prenet layer 0
    if (!Z0) {
        {
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
            gemv_acc<8>(xs, mel, g0, 0, r0a, acc);
            gemv_store(g0, P0, acc, partial);
        }
        asm volatile("" ::: "memory");                    // r0a is dead from here: its registers take the query weights
        gemv_load<8, EXACT>(P.wq, P1, A, g2, 0, r2a);          // query weights: in flight while prenet0/1 compute
        __syncthreads();
        for (int c = tid; c < P0; c += FT) y0[c] = fmaxf(reduce_partial(partial, g0.kparts, P0, c) + sb0[c], 0.f) * sk0[c];
        __syncthreads();
    }
    GT_STAMP(P.dbg, 2);
    // ---- prenet layer 1
    {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        gemv_acc<8, EXACT>(y0, P0, g1, 0, r1a, acc);
        gemv_acc<8, EXACT>(y0, P0, g1, 8, r1b, acc);
        gemv_store(g1, P1, acc, partial);
    }
    if (Z0) {
        asm volatile("" ::: "memory");                    // W1 is consumed: its registers take the query weights and the rows
        gemv_load<8, EXACT>(P.wq, P1, A, g2, 0, r2a, rbq);
        load_rows(v0, 0);
    }
    __syncthreads();
    for (int c = tid; c < P1; c += FT) {
        const float v = fmaxf(reduce_partial(partial, g1.kparts, P1, c) + sb1[c], 0.f) * sk1[c];
        y1[c] = v;
        P.xa[gt_blk_off(b, c, P.MT)] = v;           // LSTM-1 input (blocked), k in [0, P1)
        if (P.xah) P.xah[gt_blk_off_h(b, c, P.MT)] = gt_bf16_bits(v);
    }
    __syncthreads();
    GT_STAMP(P.dbg, 3);
    // ---- query projection
    {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        gemv_acc<8, EXACT>(y1, P1, g2, 0, r2a, acc);
        gemv_store(g2, A, acc, partial);
    }
    __syncthreads();
    for (int c = tid; c < A; c += FT) qs[c] = reduce_partial(partial, g2.kparts, A, c) + sbq[c];
    if (Z0) store_rows(v0);
    __syncthreads();
    GT_STAMP(P.dbg, 4);

    // ---- scores: L lanes per memory row read their NP 16-byte pieces of the LDS tile
    const int nchunks = (Tv + ROWS - 1) / ROWS;
    for (int c = 0; c < nchunks; ++c) {
        if (c > 0) {                                   // Tv > ROWS: stream further chunks through the one tile
            float4 v[NP];
            load_rows(v, c);
            __syncthreads();
            store_rows(v);
            __syncthreads();
        }
        if (LSA) {
            const int wave = tid >> 6, l15 = lane & 15, lq = lane >> 4;
            const int r0 = c * ROWS, lpad = (LK - 1) / 2;              // TF 'same', stride 1: (k - 1) // 2 zeros in front
            // location features of this chunk's rows: Conv1D(state) + bias (Layers.py:362-363) as a Toeplitz product
            const int NF = LFc >> 4;
            for (int it = wave; it < (ROWS / 16) * NF; it += FT / 64) {
                const int m = it / NF, n = it - m * NF;
                const float cb = lcbs[16 * n + l15];
                f32x4 acc = {cb, cb, cb, cb};
#pragma unroll 2
                for (int ks = 0; ks < (LKp >> 2); ++ks) {
                    const int j = 4 * ks + lq, ts = r0 + 16 * m + l15 + j - lpad;
                    const float xa = (j < LK && ts >= 0 && ts < Tv) ? pv[ts] : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, lcw[j * LCS + 16 * n + l15], acc, 0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) lfeat[(16 * m + 4 * lq + i) * LFS + 16 * n + l15] = acc[i];
            }
            __syncthreads();
            // score = sum_a tanh(q + key + Dense(lfeat) + bias)   (Layers.py:364, 407; no v vector, scale 1)
            // a wave = one 16-row tile x NPW channel tiles: one location-feature fragment per k step feeds NPW independent MFMAs,
            // the channel tiles are summed in registers and the 16 lanes of a row group with four DPP moves
            constexpr int NA = A / 16;
            constexpr int NPW = NA * ROWS / 256 > 0 ? NA * ROWS / 256 : 1;      // channel tiles per wave
            constexpr int WPM = NA / NPW;                                          // waves per row tile
            static_assert(NA * (ROWS / 16) == NPW * (FT / 64) && WPM * NPW == NA, "16 waves cover the ROWS x A score tile exactly");
            {
                const int m = wave / WPM, wn = wave - m * WPM;
                f32x4 acc[NPW];
#pragma unroll
                for (int g = 0; g < NPW; ++g) { const float lb = labias[16 * (wn * NPW + g) + l15]; acc[g] = f32x4{lb, lb, lb, lb}; }
                const float* lfr = lfeat + (16 * m + l15) * LFS + lq;
                const float* lwr = ldw + lq * LDWS + 16 * wn * NPW + l15;
#pragma unroll 2
                for (int ks = 0; ks < (LFp >> 2); ++ks) {
                    const float fa = lfr[4 * ks];
#pragma unroll
                    for (int g = 0; g < NPW; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, lwr[4 * ks * LDWS + 16 * g], acc[g], 0, 0, 0);
                }
                float e[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < NPW; ++g) {
                    const int a = 16 * (wn * NPW + g) + l15;
                    const float qa = qs[a];
#pragma unroll
                    for (int i = 0; i < 4; ++i) e[i] += gt_tanh(qa + tile[(16 * m + 4 * lq + i) * LD + a] + acc[g][i]);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    e[i] += front_dpp<0xB1>(e[i]);      // quad_perm [1,0,3,2]
                    e[i] += front_dpp<0x4E>(e[i]);      // quad_perm [2,3,0,1]
                    e[i] += front_dpp<0x141>(e[i]);     // row_half_mirror: the other quad of the half row
                    e[i] += front_dpp<0x140>(e[i]);     // row_mirror: the other half of the 16-lane row
                    if (l15 == 0) partial[wn * ROWS + 16 * m + 4 * lq + i] = e[i];
                }
            }
            __syncthreads();
            for (int r = tid; r < ROWS; r += FT) {
                float z = 0.f;
#pragma unroll
                for (int n = 0; n < WPM; ++n) z += partial[n * ROWS + r];
                if (c * ROWS + r < Tv) sc[c * ROWS + r] = z;
            }
            continue;
        }
        const int t = c * ROWS + row;
        f32x2 s2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int a0 = 4 * (li + L * j);
            const float4 m4 = *reinterpret_cast<const float4*>(tile + row * LD + a0);
            const float4 q4 = *reinterpret_cast<const float4*>(qs + a0);
            const float4 w4 = *reinterpret_cast<const float4*>(vs + a0);
            s2 = __builtin_elementwise_fma(f32x2{w4.x, w4.y}, gt_tanh2(f32x2{q4.x, q4.y} + f32x2{m4.x, m4.y}), s2);
            s2 = __builtin_elementwise_fma(f32x2{w4.z, w4.w}, gt_tanh2(f32x2{q4.z, q4.w} + f32x2{m4.z, m4.w}), s2);
        }
        float s = s2.x + s2.y;
        s = gt_row_sum<L>(s);
        if (li == 0 && t < Tv) sc[t] = s + sbias;
    }
    __syncthreads();
    GT_STAMP(P.dbg, 5);
    // ---- noise + sigmoid + alignment
    if (LSA) {
        // softmax (or the smoothing normalisation, Layers.py:426-444) over the Tv positions: one wave, a serial run per lane;
        // the state the next step's location features read is the running sum of the alignments (or the last one)
        if (tid < 64) {
            const int per = (Tv + 63) / 64;
            const int t0 = lane * per, t1 = min(Tv, t0 + per);
            float mx = -INFINITY;
            for (int t = t0; t < t1; ++t) mx = fmaxf(mx, sc[t]);
            mx = gt_wave_max(mx);
            float sum = 0.f;
            for (int t = t0; t < t1; ++t) {
                const float e = P.lsa_smoothing ? 1.f / (1.f + expf(-sc[t])) : expf(sc[t] - mx);
                al[t] = e;
                sum += e;
            }
            sum = gt_wave_sum(sum);
            const float inv = 1.f / sum;
            for (int t = t0; t < t1; ++t) {
                al[t] *= inv;
                P.lsa_state[(size_t)b * TvFull + t] = P.lsa_cumulate ? pv[t] + al[t] : al[t];
            }
        }
    } else if (P.type == GSTTACO_ATT_SMA) {
        // each position needs its own and its left neighbour's probability: both sigmoids are evaluated here (same
        // arithmetic as a separate sigmoid pass, one barrier and one LDS round trip fewer)
        for (int t = tid; t < Tv; t += FT) {
            const bool nz = P.sigmoid_noise > 0.f;
            float v = pv[t] * gt_sigmoid(sc[t] + (nz ? snz[t] : 0.f));
            if (t > 0) v = __builtin_fmaf(pv[t - 1], 1.f - gt_sigmoid(sc[t - 1] + (nz ? snz[t - 1] : 0.f)), v);
            al[t] = v;
        }
    } else {
        for (int t = tid; t < Tv; t += FT) {
            float s = sc[t];
            if (P.sigmoid_noise > 0.f) s += snz[t];
            sc[t] = gt_sigmoid(s);
        }
        __syncthreads();
    }
    if (!LSA && P.type != GSTTACO_ATT_SMA && tid < 64) {
        const int per = (Tv + 63) / 64;
        const int t0 = lane * per, t1 = min(Tv, t0 + per);
        float run = 0.f;
        for (int t = t0; t < t1; ++t) run += logf(fminf(fmaxf(1.f - sc[t], 1.17549435e-38f), 1.f));
        float base = front_wave_incl_scan(run, lane) - run;
        for (int t = t0; t < t1; ++t) {
            const float lg = logf(fminf(fmaxf(1.f - sc[t], 1.17549435e-38f), 1.f));
            al[t] = expf(base);
            base += lg;
        }
        run = 0.f;
        for (int t = t0; t < t1; ++t) run += pv[t] / fminf(fmaxf(al[t], 1e-10f), 1.f);
        base = front_wave_incl_scan(run, lane) - run;
        for (int t = t0; t < t1; ++t) {
            base += pv[t] / fminf(fmaxf(al[t], 1e-10f), 1.f);
            al[t] = sc[t] * al[t] * base;
        }
    }
    __syncthreads();
    GT_STAMP(P.dbg, 6);
    for (int t = tid; t < TvFull; t += FT) P.align[(size_t)b * P.ldalign + t] = t < Tv ? al[t] : 0.f;

    // ---- context: ctx[a] = sum_t al[t] * pm[t][a]; lane = channel a (conflict-free column reads of the tile),
    //      CPARTS row groups reduced through LDS
    const int ca = tid % A, cp = tid / A;
    float cacc = 0.f;
    for (int c = nchunks - 1; c >= 0; --c) {           // the tile still holds the LAST chunk of the score pass
        if (c != nchunks - 1) {
            float4 v[NP];
            load_rows(v, c);
            __syncthreads();
            store_rows(v);
            __syncthreads();
        }
        const int nr = min(ROWS, Tv - c * ROWS);
        const float* alc = al + c * ROWS;
        float p0 = 0.f, p1 = 0.f;
        int t = cp;
        for (; t + CPARTS < nr; t += 2 * CPARTS) {
            p0 = __builtin_fmaf(alc[t], tile[t * LD + ca], p0);
            p1 = __builtin_fmaf(alc[t + CPARTS], tile[(t + CPARTS) * LD + ca], p1);
        }
        if (t < nr) p0 = __builtin_fmaf(alc[t], tile[t * LD + ca], p0);
        cacc += p0 + p1;
    }
    red[cp * A + ca] = cacc;
    __syncthreads();
    for (int a = tid; a < A; a += FT) {
        float v[CPARTS];
#pragma unroll
        for (int w = 0; w < CPARTS; ++w) v[w] = red[w * A + a];
        float z = 0.f;
#pragma unroll
        for (int w = 0; w < CPARTS; ++w) z += v[w];
        P.xa[gt_blk_off(b, P1 + a, P.MT)] = z;      // context, k in [P1, P1+A)
        if (P.xah) P.xah[gt_blk_off_h(b, P1 + a, P.MT)] = gt_bf16_bits(z);
    }
    GT_STAMP(P.dbg, 7);
    if (b < P.utt_jobs) {       // (see gt_dec_front_lean_kernel)
        __syncthreads();
        front_worker<LEAN>(P, smem, P.n_workers + b);
    }
}
