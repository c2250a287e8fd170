// Group kernels of the persistent decode launch (fp32, 33..128 rows; 17..32 rows as two groups of 16 as an experiment): included by
// persist_decode.hip inside its anonymous namespace, behind the one-group kernel whose primitives (waits, fragment loads, MFMA pieces,
// reductions, the per-utterance chain) they share.  DESIGN.md 3.1c, EXPERIMENTS.md round 5 item 1.
// INVARIANT (give-up safety, PD_PHASE_ABORT in persist_decode.hip): after a bounded wait has given up, the workgroup runs the REST of the
// step on whatever the wait left and leaves at the top of its next step.  That is only safe because NO ADDRESS AND NO LOOP BOUND behind a
// wait depends on data that came through a hand-off: every index below is a function of blockIdx / threadIdx / the step counter / launch
// arguments (token lengths are read from the caller's tensor before the first wait).  Keep it so when editing this file.
#pragma once

// ====================================================================================================================== groups
// Batches above 32 rows (and, as an experiment, 17..32 rows as two groups of 16): G groups of 16 MTG rows go through ONE set of
// resident weights.  Every workgroup still owns gate tile `blockIdx.x` of both cells and runs each GEMM phase group by group:
//   * ONE wait per phase for all groups (context flags, h1 arrivals, h2 arrivals) instead of one per group -- a satisfied wait still
//     costs a poll's round trip;
//   * the NEXT group's activation fragments are requested while the current group is multiplied (cell 1: a second fragment buffer;
//     the K = 1024 phases: each fragment re-requested right behind the MFMAs that consumed it, lean_body.h's multi-chunk order), so a
//     group's hand-off and fragment latencies hide behind its neighbours' arithmetic -- v1 of this kernel ran them back to back and
//     spent 30 of 59 us per step at 128 rows waiting for fragments (profiles/r05_group_stamps.txt).
// Per group the arithmetic is the one-group kernel's (= the launch path's single-chunk bodies', which its multi-chunk bodies
// reproduce per 32-row chunk): bitwise the launches at any batch.  Roles: workgroup b < B runs utterance b's chain, then its tile for
// every group with the tile's weights STREAMED (the chain's operands own the registers meanwhile; nobody helps: with one chain per
// CU on half of the chip there is no idle half to help from); the next pj_tiles x MTG own a projection (tile, M-tile of the group);
// the rest are plain.  Control: per-group arrival counters, per-utterance flags as before.
template <int GM> struct PdG { float c1[GM], c2[GM], p1[GM], p2[GM]; };


// ---- one wait for ALL groups: the context flags of every utterance (two per lane: B <= 128) ...
__device__ __forceinline__ void pd_wait_flags_all(const PersistDecodeArgs& A, const uint32_t* f, uint32_t want, PdShared* sh) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const bool h0 = lane < A.B, h1 = lane + 64 < A.B;
        const uint32_t* p0 = f + (h0 ? lane : 0) * PD_FS;
        const uint32_t* p1 = f + (h1 ? lane + 64 : 0) * PD_FS;
        uint32_t spins = 0;
        for (;;) {
            uint32_t v0, v1;
            asm volatile("global_load_dword %0, %2, off sc1\n\tglobal_load_dword %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(p0), "v"(p1) : "memory");
            const bool ok = (!h0 || v0 >= want) && (!h1 || v1 >= want);
            if (__builtin_amdgcn_readfirstlane(__popcll(__ballot(ok))) == 64) break;
            if (++spins > PD_SPIN_MAX) { if (lane == 0) pd_give_up(A, sh, true); break; }
            if ((spins & 63u) == 0u && __builtin_amdgcn_readfirstlane(pd_ld_sc1(A.err)) != 0u) { if (lane == 0) pd_give_up(A, sh, false); break; }
        }
    }
    __syncthreads();
}
// ... and the arrival counters of groups [0, G) (PD_NSH shards each, one per lane; all requested, then one wait)
template <int GM>
__device__ __forceinline__ void pd_wait_count_all(const PersistDecodeArgs& A, const uint32_t* c, uint32_t want, PdShared* sh) {
    static_assert(PD_NSH == 64, "one counter shard per lane");
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        uint32_t spins = 0;
        for (;;) {
            uint32_t u[GM];
#pragma unroll
            for (int g = 0; g < GM; ++g) {
                u[g] = want;
                if (g < A.G) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(u[g]) : "v"(c + g * (PD_NSH * 32) + lane * 32) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            bool ok = true;                 // (per shard: pd_wait_count)
#pragma unroll
            for (int g = 0; g < GM; ++g) { asm volatile("" : "+v"(u[g])); ok = ok && u[g] >= want; }
            if (__builtin_amdgcn_readfirstlane(__popcll(__ballot(ok))) == 64) break;
            if (++spins > PD_SPIN_MAX) { if (lane == 0) pd_give_up(A, sh, true); break; }
            if ((spins & 63u) == 0u && __builtin_amdgcn_readfirstlane(pd_ld_sc1(A.err)) != 0u) { if (lane == 0) pd_give_up(A, sh, false); break; }
        }
    }
    __syncthreads();
}

// the wave's eight K = 1024 fragments times one weight tile.  ORDER16: two accumulator pairs, fragments 0, 2, 4, 6 -> a, 1, 3, 5, 7 -> b
// (the launch path's 16-wave order, pd_rec_tile); else all eight -> a in ascending order.  RELOAD: fragment i of group `gn` is requested
// right behind the MFMAs that consumed fragment i (gn = the group itself when there is no next one: re-read, never multiplied).
template <int MTG, bool ORDER16, bool RELOAD>
__device__ __forceinline__ void pd_g_mma8(float4 (&x0)[8], float4 (&x1)[8], const float4 (&w)[8], f32x4& a0, f32x4& a1, f32x4& b0, f32x4& b1,
                                          const float* base, int MT, int gn) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const auto rs = gt_rsrc(base, 0x7FFFF000u);
    const uint32_t m0 = (uint32_t)(MTG * gn) * 1024u, m1 = (uint32_t)min(MTG * gn + 1, MT - 1) * 1024u;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        f32x4& c0 = (ORDER16 && (i & 1)) ? b0 : a0;
        f32x4& c1 = (ORDER16 && (i & 1)) ? b1 : a1;
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].x, w[i].x, c0, 0, 0, 0); if (MTG == 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].x, w[i].x, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].y, w[i].y, c0, 0, 0, 0); if (MTG == 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].y, w[i].y, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].z, w[i].z, c0, 0, 0, 0); if (MTG == 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].z, w[i].z, c1, 0, 0, 0);
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].w, w[i].w, c0, 0, 0, 0); if (MTG == 2) c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].w, w[i].w, c1, 0, 0, 0);
        if (RELOAD) {
            PD_PIN();
            const uint32_t so = (uint32_t)((wave + i * PD_NW) * MT) * 1024u;
            x0[i] = gt_bload4_sc1(rs, (uint32_t)lane * 16u, so + m0);
            if (MTG == 2) x1[i] = gt_bload4_sc1(rs, (uint32_t)lane * 16u, so + m1);
            PD_PIN();
        }
    }
}
// sums of a recurrent-half tile from its accumulators (pd_rec_tile's orders).  TWOPASS: the 16-slab sum through EIGHT slabs -- virtual
// waves 0..7, then 8..15 onto the running sum: the same sequence of additions (chain workgroups: their LDS holds the processed memory).
// `cnt`: a deferred arrival (pd_reduce_arrive) or NULL
template <bool ORDER16, bool TWOPASS>
__device__ __forceinline__ float pd_g_rec_sum(float* lds, float bias, const f32x4& a0, const f32x4& a1, const f32x4& b0, const f32x4& b1, uint32_t* cnt = nullptr,
                                              bool two = true) {
    const int wave = threadIdx.x >> 6;
    if (ORDER16) {
        if (TWOPASS && two) {
            pd_spill(lds, wave, a0, a1);
            const float z = pd_reduce_arrive<8>(lds, bias, cnt);
            pd_spill(lds, wave, b0, b1);
            return pd_reduce<8>(lds, z);
        }
        pd_spill(lds, wave, a0, a1);
        pd_spill(lds, wave + 8, b0, b1);
        return pd_reduce_arrive<16>(lds, bias, cnt);
    }
    pd_spill(lds, wave, a0, a1);
    return pd_reduce_arrive<8>(lds, bias, cnt);
}

// ---- the phases, for all groups.  State access: `c(g)` / `p(g)` return references (registers, or LDS in the projection role).
// LSTM cell 1 of every group: z = [p | ctx] . W1x + p1[g]; the next group's fragments in a second buffer
template <int GM, int MTG, class C1, class P1>
__device__ __forceinline__ void pd_g_cell1_all(const PersistDecodeArgs& A, const float4 (&wx1)[3], int t, int tile, float* lds, C1 c1, P1 p1, PdShared* sh, int role, int zt) {
    constexpr int RG = 16 * MTG;
    const int par = t & 1, MT = A.MT;
    pd_wait_flags_all(A, A.ctl + zt + PD_F_C, (uint32_t)t + 1u, sh);       // (a chain's context flag follows its prenet flag)
    PD_PHASE_ABORT(sh);
    PD_STAMP(role, 2);
    float4 xa0[3], xa1[3], xb0[3], xb1[3];
    pd_g_xload<MTG, 0, 3, 3>(A.xa[par], MT, 0, xa0, xa1);
#pragma unroll
    for (int g = 0; g < GM; ++g) {
        if (g < A.G) {
            float4 (&x0)[3] = (g & 1) ? xb0 : xa0;
            float4 (&x1)[3] = (g & 1) ? xb1 : xa1;
            PD_PIN();
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0};
            pd_g_mma<MTG, 3, 0, 1, 3>(x0, x1, wx1, a0, a1);
            // (group g - 1's h1 stores were left in flight: their acknowledgement arrived under this group's MFMAs; drained here, its
            // arrival is signalled behind the reduction's first barrier -- the last group's at once, everybody waits for it.  The next
            // group's fragments are requested BEHIND the drain -- in front of it the drain waited for them -- and land under the epilogue)
            if (g > 0) pd_drain();
            PD_PIN();
            if (g + 1 < GM && g + 1 < A.G) {
                if (g & 1) pd_g_xload<MTG, 0, 3, 3>(A.xa[par], MT, g + 1, xa0, xa1);
                else pd_g_xload<MTG, 0, 3, 3>(A.xa[par], MT, g + 1, xb0, xb1);
            }
            PD_PIN();
            pd_spill(lds, threadIdx.x >> 6, a0, a1);
            const float z = pd_reduce_arrive<8>(lds, p1(g), g > 0 ? A.ctl + zt + PD_CNT3 + (g - 1) * (PD_NSH * 32) : nullptr);
            pd_gates_store(z, c1(g), A.h1[par], tile + zt, A.B, MT, RG * g, RG);
            if (g == A.G - 1) pd_arrive(A.ctl + zt + PD_CNT3 + g * (PD_NSH * 32));
            PD_STAMP(role, 3 + 7 * g);
        }
    }
}

// LSTM cell 2 of every group: z = h1_t . W2x + p2[g]; REC1: then, from the same fragments, the layer-1 recurrent half for the next step,
// the next group's fragments requested behind its MFMAs (else behind cell 2's own)
// STREAM_H2 (chain workgroups): the layer-2 recurrent tile for the next phase is requested once the last group's cell-2 MFMAs have
// released W2x's registers
template <int GM, int MTG, bool REC1, bool TWOPASS, bool STREAM_H2, class C2, class P2, class P1>
__device__ __forceinline__ void pd_g_cell2_all(const PersistDecodeArgs& A, const float4 (&wx2)[8], const float4 (&wh1)[8], float4 (&wh2)[8], int t, int tile, float* lds,
                                               C2 c2, P2 p2, P1 p1, PdShared* sh, int role, int zt) {
    constexpr int RG = 16 * MTG;
    const int par = t & 1, MT = A.MT;
    pd_wait_count_all<GM>(A, A.ctl + zt + PD_CNT3, PD_WANT(A, t), sh);
    PD_PHASE_ABORT(sh);
    PD_STAMP(role, 4);
    float4 x0[8], x1[8];
    pd_g_xload<MTG, 0, 8, 8>(A.h1[par], MT, 0, x0, x1);
    PD_PIN();
    const float bias1 = A.b1h[tile * 16 + (threadIdx.x & 15)];
#pragma unroll
    for (int g = 0; g < GM; ++g) {
        if (g < A.G) {
            const int gn = min(g + 1, A.G - 1);
            f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
            pd_g_mma8<MTG, false, !REC1>(x0, x1, wx2, a0, a1, b0, b1, A.h1[par], MT, gn);
            // REC1: group g - 1's h2 stores stayed in flight under its recurrent half and this group's MFMAs -- which waited for every
            // fragment re-requested in between, so nothing is left to wait for here -- and its arrival is signalled behind this
            // reduction's first barrier; the last group's at once (everybody waits for it)
            const bool deferred = REC1 && g > 0;
            if (deferred) pd_drain();
            pd_spill(lds, threadIdx.x >> 6, a0, a1);
            const float z = pd_reduce_arrive<8>(lds, p2(g), deferred ? A.ctl + zt + PD_CNT4 + (g - 1) * (PD_NSH * 32) : nullptr);
            pd_gates_store(z, c2(g), A.h2[par], tile + zt, A.B, MT, RG * g, RG);
            if (!REC1 || g == A.G - 1) pd_arrive(A.ctl + zt + PD_CNT4 + g * (PD_NSH * 32));
            PD_STAMP(role, 5 + 7 * g);
            if (STREAM_H2 && g == A.G - 1 && t + 1 < A.steps) pd_load_tile<8>(A.w2h, tile, wh2);
            if (REC1) {
                a0 = f32x4{0, 0, 0, 0}; a1 = f32x4{0, 0, 0, 0};
                pd_g_mma8<MTG, true, true>(x0, x1, wh1, a0, a1, b0, b1, A.h1[par], MT, gn);
                p1(g) = pd_g_rec_sum<true, TWOPASS>(lds, bias1, a0, a1, b0, b1, nullptr, A.twopass != 0);
                PD_STAMP(role, 6 + 7 * g);
            }
        }
    }
}

// a recurrent half of every group from the state in memory: LAYER 1 -> p(g) = h1_t . W1h + b1 (projection role), 2 -> h2_t . W2h + b2
// (tiles below co_tiles sum in the projection launch's co-workers' 8-wave order)
template <int GM, int MTG, int LAYER, bool TWOPASS, bool WAIT, class P>
__device__ __forceinline__ void pd_g_rec_all(const PersistDecodeArgs& A, const float4 (&wh)[8], int t, int tile, float* lds, P p, PdShared* sh, int role, int zt) {
    const int MT = A.MT;
    if (WAIT) {
        pd_wait_count_all<GM>(A, A.ctl + zt + PD_CNT4, PD_WANT(A, t), sh);
        PD_PHASE_ABORT(sh);
    }
    PD_STAMP(role, 7);
    const float* hb = LAYER == 1 ? A.h1[t & 1] : A.h2[t & 1];
    const float bias = (LAYER == 1 ? A.b1h : A.b2h)[tile * 16 + (threadIdx.x & 15)];
    // (the summation order is a property of the tile: the branch encloses the whole phase -- first fragment loads included -- so that
    // the fragments carried from group to group belong to ONE arm: shared between the arms, the allocator gave every re-requested
    // fragment a register of its own and spilled)
    if (LAYER == 1 || tile >= A.co_tiles) {
        float4 x0[8], x1[8];
        pd_g_xload<MTG, 0, 8, 8>(hb, MT, 0, x0, x1);
        PD_PIN();
#pragma unroll
        for (int g = 0; g < GM; ++g) {
            if (g < A.G) {
                f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
                pd_g_mma8<MTG, true, true>(x0, x1, wh, a0, a1, b0, b1, hb, MT, min(g + 1, A.G - 1));
                p(g) = pd_g_rec_sum<true, TWOPASS>(lds, bias, a0, a1, b0, b1, nullptr, A.twopass != 0);
                PD_STAMP(role, 8 + 7 * g);
            }
        }
    } else {
        float4 x0[8], x1[8];
        pd_g_xload<MTG, 0, 8, 8>(hb, MT, 0, x0, x1);
        PD_PIN();
#pragma unroll
        for (int g = 0; g < GM; ++g) {
            if (g < A.G) {
                f32x4 a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, b0 = {0, 0, 0, 0}, b1 = {0, 0, 0, 0};
                pd_g_mma8<MTG, false, true>(x0, x1, wh, a0, a1, b0, b1, hb, MT, min(g + 1, A.G - 1));
                p(g) = pd_g_rec_sum<false, TWOPASS>(lds, bias, a0, a1, b0, b1);
                PD_STAMP(role, 8 + 7 * g);
            }
        }
    }
}

// a tile's four GEMM halves for every group; CHAIN: utterance blockIdx.x's chain first, the tile's weights streamed -- each requested a
// phase ahead of its use
template <int GM, int MTG, bool CHAIN>
__device__ __forceinline__ void pd_g_run_tile(const PersistDecodeArgs& A, float* smem, PdShared* sh) {
    float* lds = smem;
    const int tile = blockIdx.x, b = blockIdx.x, tid = threadIdx.x, col = tid & 15;
    constexpr int role = CHAIN ? 0 : 2;
    PdW W;
    PdG<GM> S;
#pragma unroll
    for (int g = 0; g < GM; ++g) { S.c1[g] = 0.f; S.c2[g] = 0.f; S.p1[g] = A.b1h[tile * 16 + col]; S.p2[g] = A.b2h[tile * 16 + col]; }
    auto c1 = [&](int g) -> float& { return S.c1[g]; };
    auto c2 = [&](int g) -> float& { return S.c2[g]; };
    auto p1 = [&](int g) -> float& { return S.p1[g]; };
    auto p2 = [&](int g) -> float& { return S.p2[g]; };
    PdChainLds L{};
    PdChainRegs R{};
    if (CHAIN) {
        L = pd_carve(smem, A.tvp, A.twopass ? 8 : 16);
        R.Tv = A.tok_len ? max(1, min(A.Tv, (int)A.tok_len[b])) : A.Tv;
        R.drop = A.drop_rate > 0.f;
        R.hashed = R.drop && A.keep_hash != 0;
        R.noisy = A.sigmoid_noise > 0.f;
        R.seed = R.hashed ? *A.seed_ptr : 0ull;
        R.bias1 = tid < PD_P ? A.b1[tid] : 0.f;
        R.biasq = tid < PD_A ? A.bq[tid] : 0.f;
        R.sbias = A.score_bias[0];
        const float4* src = reinterpret_cast<const float4*>(A.pm + (size_t)b * A.Tv * PD_A);
        for (int e = tid; e < A.tvp * PD_A / 4; e += PD_NT) {
            const int row = e / (PD_A / 4), c4 = e % (PD_A / 4);
            *reinterpret_cast<float4*>(L.tile + row * PD_LDV + 4 * c4) = row < R.Tv ? src[(size_t)row * (PD_A / 4) + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (tid < PD_A) L.vs[tid] = A.av[tid];
        if (tid < A.tvp) L.pv[tid] = tid == 0 ? 1.f : 0.f;
    } else {
        pd_load_tile<3>(A.w1x, tile, W.x1); pd_load_tile<8>(A.w2x, tile, W.x2);
        pd_load_tile<8>(A.w1h, tile, W.h1); pd_load_tile<8>(A.w2h, tile, W.h2);
    }
    __syncthreads();
    // The chain needs the whole register file (all of prenet 1's weights in flight): the tile's per-group state -- cell states,
    // recurrent halves -- waits in memory meanwhile (its own rows of `stash`, written and re-read by the same thread: L2 hits,
    // requested back before the first flag wait).
    for (int t = 0; t < A.steps; ++t) {
        if (sh->abort) return;          // (the one abort check of the step: PD_PHASE_ABORT)
        PD_STAMP(role, 0);
        PD_ZT(zt);
        float* st = A.stash + (size_t)blockIdx.x * (4 * GM * PD_NT) + tid + zt;
        if (CHAIN) {
#pragma unroll
            for (int g = 0; g < GM; ++g) {
                st[(4 * g + 0) * PD_NT] = S.c1[g]; st[(4 * g + 1) * PD_NT] = S.c2[g];
                st[(4 * g + 2) * PD_NT] = S.p1[g]; st[(4 * g + 3) * PD_NT] = S.p2[g];
            }
            PD_PIN();
            float4 unused4;
            int bs = b;
            asm volatile("" : "+s"(bs));
            pd_chain<false>(A, L, R, t, bs, sh, unused4, zt);
            PD_PHASE_ABORT(sh);
            PD_STAMP(role, 1);
            PD_PIN();
            pd_load_tile<3>(A.w1x, tile, W.x1);
#pragma unroll
            for (int g = 0; g < GM; ++g) {
                S.c1[g] = st[(4 * g + 0) * PD_NT]; S.c2[g] = st[(4 * g + 1) * PD_NT];
                S.p1[g] = st[(4 * g + 2) * PD_NT]; S.p2[g] = st[(4 * g + 3) * PD_NT];
            }
            pd_load_tile<8>(A.w2x, tile, W.x2); pd_load_tile<8>(A.w1h, tile, W.h1);        // (for the NEXT phase: they arrive during cell 1)
        }
        pd_g_cell1_all<GM, MTG>(A, W.x1, t, tile, lds, c1, p1, sh, role, zt);
        PD_PHASE_ABORT(sh);
        pd_g_cell2_all<GM, MTG, true, CHAIN, CHAIN>(A, W.x2, W.h1, W.h2, t, tile, lds, c2, p2, p1, sh, role, zt);
        PD_PHASE_ABORT(sh);
        if (t + 1 == A.steps) break;
        pd_g_rec_all<GM, MTG, 2, CHAIN, true>(A, W.h2, t, tile, lds, p2, sh, role, zt);
        PD_PHASE_ABORT(sh);
    }
}

// a projection (tile, M-tile of the group) + the LSTM tile.  Resident: the layer-2 input half, the layer-1 recurrent half (fused into
// cell 2 as in the tile roles: as a pass of its own over h1 behind the projections it made this role the last to finish every step,
// and every other workgroup waited for its cell-1 arrivals: v1 of this kernel, 60 us per step at 128 rows) ;
// W1x (24 KB) is streamed at the start of every step (it arrives while the workgroup waits for the chains), the projection tile in
// front of the wait for the h2 arrivals, the layer-2 recurrent tile behind the projections; the per-group state (cell states, recurrent halves) lives in LDS behind the slabs -- this
// role has no chain and the LDS to spare, and not the registers.
template <int GM, int MTG>
__device__ __forceinline__ void pd_g_run_proj(const PersistDecodeArgs& A, float* lds, PdShared* sh) {
    const int tile = blockIdx.x, col = threadIdx.x & 15;
    const int pi = tile - A.n_chain, ptile = pi % A.pj_tiles, pm = pi / A.pj_tiles;
    PdW W;
    pd_load_tile<8>(A.w2x, tile, W.x2); pd_load_tile<8>(A.w1h, tile, W.h1);
    float* sl = lds + 16 * PD_SLAB + threadIdx.x;           // [4 GM][512]: c1, c2, p1, p2 of group g at rows 4 g ..
#pragma unroll
    for (int g = 0; g < GM; ++g) { sl[(4 * g + 0) * PD_NT] = 0.f; sl[(4 * g + 1) * PD_NT] = 0.f; sl[(4 * g + 2) * PD_NT] = A.b1h[tile * 16 + col]; sl[(4 * g + 3) * PD_NT] = A.b2h[tile * 16 + col]; }
    auto c1 = [&](int g) -> float& { return sl[(4 * g + 0) * PD_NT]; };
    auto c2 = [&](int g) -> float& { return sl[(4 * g + 1) * PD_NT]; };
    auto p1 = [&](int g) -> float& { return sl[(4 * g + 2) * PD_NT]; };
    auto p2 = [&](int g) -> float& { return sl[(4 * g + 3) * PD_NT]; };
    for (int t = 0; t < A.steps; ++t) {
        if (sh->abort) return;          // (the one abort check of the step: PD_PHASE_ABORT)
        PD_STAMP(1, 0);
        PD_ZT(zt);
        pd_load_tile<3>(A.w1x, tile, W.x1);
        pd_g_cell1_all<GM, MTG>(A, W.x1, t, tile, lds, c1, p1, sh, 1, zt);
        PD_PHASE_ABORT(sh);
        pd_g_cell2_all<GM, MTG, true, false, false>(A, W.x2, W.h1, W.h2, t, tile, lds, c2, p2, p1, sh, 1, zt);
        PD_PHASE_ABORT(sh);
        // every group's projection behind ONE wait for the h2 arrivals (the chains that need them start ~10 us later: their
        // workgroups still have this step's recurrent halves to multiply); the projection tile (72 KB) arrives during that wait
        float4 wpj[9];
        pd_load_tile<9>(A.wp, ptile, wpj);
        pd_wait_count_all<GM>(A, A.ctl + zt + PD_CNT4, PD_WANT(A, t), sh);
        PD_PHASE_ABORT(sh);
#pragma unroll
        for (int g = 0; g < GM; ++g)
            // (an M-tile past the batch's last -- 33..48 rows: group 1's second -- does not exist: its fragment would be read one block
            // past the END of the state buffers, harmless (its rows are never stored) until the buffer is the last of a mapping: round 6
            // met that layout as a memory fault at 40 rows.  The workgroup skips the tile; the condition is uniform.)
            if (g < A.G && MTG * g + pm < A.MT) pd_proj<true>(A, wpj, t, ptile, MTG * g + pm, lds, sh, g);
        if (t + 1 == A.steps) break;
        pd_load_tile<8>(A.w2h, tile, W.h2);
        PD_PIN();
        pd_g_rec_all<GM, MTG, 2, false, false>(A, W.h2, t, tile, lds, p2, sh, 1, zt);
    }
}

template <int GM, int MTG>
__global__ __launch_bounds__(PD_NT) void gt_persist_decode_g_kernel(PersistDecodeArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ PdShared sh;
    if (threadIdx.x == 0) sh.abort = 0;
    __syncthreads();
    const int tile = blockIdx.x;
#ifndef PD_GONLY
#define PD_GONLY -1         // (register-budget diagnosis: compile one role alone)
#endif
    if (tile < A.n_chain) { if (PD_GONLY < 0 || PD_GONLY == 0) pd_g_run_tile<GM, MTG, true>(A, smem, &sh); }
    else if (tile < A.n_chain + A.pj_tiles * MTG) { if (PD_GONLY < 0 || PD_GONLY == 1) pd_g_run_proj<GM, MTG>(A, smem, &sh); }
    else if (PD_GONLY < 0 || PD_GONLY == 2) pd_g_run_tile<GM, MTG, false>(A, smem, &sh);
}
