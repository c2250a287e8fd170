// The bf16 kernel of the persistent decode launch (Use_Mixed_Precision, <= 64 rows as ONE group, helper workgroups for the chain tiles'
// recurrent halves): included by persist_decode.hip inside its anonymous namespace, behind the group kernels (it shares their waits and
// the per-utterance chain).  DESIGN.md 3.1d, EXPERIMENTS.md round 5 item 2.
// INVARIANT (give-up safety, PD_PHASE_ABORT in persist_decode.hip): after a bounded wait has given up, the workgroup runs the REST of the
// step on whatever the wait left and leaves at the top of its next step.  That is only safe because NO ADDRESS AND NO LOOP BOUND behind a
// wait depends on data that came through a hand-off: every index below is a function of blockIdx / threadIdx / the step counter / launch
// arguments (token lengths are read from the caller's tensor before the first wait).  Keep it so when editing this file.
#pragma once

// ====================================================================================================================== bf16
// Mixed precision (Use_Mixed_Precision: bf16 GEMM operands, fp32 accumulation / state / epilogues; BASELINE configs[4]) on the
// persistent launch: ONE group of up to 64 rows (four M-tiles).  On bf16 MFMA (v_mfma_f32_16x16x32_bf16) the step's GEMMs are a few
// hundred nanoseconds; what a step costs is its dependent phases and their epilogues, so the batch is NOT cut into groups of 32 rows
// (each with its own epilogues): a wave's fragments of four M-tiles are as many registers as two in fp32.  Roles as in the group
// kernels: workgroup b < B runs utterance b's chain (fp32, as in the launch path) and its gate tile with streamed weights, the next
// pj_tiles x MT own a projection (tile, M-tile), the rest are plain.  Activations travel ONLY as the bf16 mirrors the launch path's
// multi-chunk bodies read (kernels.h gt_blk_off_h: the MFMA's A operand as is), rounded once (RNE) by their producer -- the same
// values the launch path's consumers get, in the same k-block -> wave assignment and summation orders (lean_body.h
// gt_lean_core_bf16 / gt_lean_mc on eight waves; the recurrent halves in the front launch's 16-wave or the projection launch's
// 8-wave order): bitwise the launches.
constexpr int PDH_MT = 4, PDH_BMAX = 16 * PDH_MT;
struct PdWh { u32x4 x1[2], h1[4], x2[4], h2[4]; };
constexpr int PDH_SLAB = PDH_MT * 16 * 17;

// (zt: the per-step opaque zero for tiles streamed inside the step loop -- their addresses are loop invariants the compiler otherwise
// hoists and, in the helper role, spills)
template <int KPW32>
__device__ __forceinline__ void pdh_load_tile(const float* wp, int tile, int nkb32, u32x4 (&dst)[KPW32], int zt = 0) {
    const int lane = (threadIdx.x & 63) + zt;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const u32x4* wl = reinterpret_cast<const u32x4*>(wp) + ((size_t)tile * nkb32 + wave) * 64 + lane;
#pragma unroll
    for (int i = 0; i < KPW32; ++i) dst[i] = wl[(size_t)((wave + i * PD_NW < nkb32) ? i : 0) * PD_NW * 64];      // (past the end: re-read, never multiplied)
}
// the wave's fragments of 32-k blocks kb_off + wave + 8 i (i < KPW32, those below nkb32) of a mirror, all PDH_MT M-tiles
template <int KPW32>
__device__ __forceinline__ void pdh_xload(const uint16_t* base, int MT, int nkb32, int kb_off, u32x4 (&x)[PDH_MT][KPW32]) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const auto rs = gt_rsrc(base, 0x7FFFF000u);
#pragma unroll
    for (int i = 0; i < KPW32; ++i) {
        const int kb = kb_off + ((wave + i * PD_NW < nkb32) ? wave + i * PD_NW : wave);
#pragma unroll
        for (int mt = 0; mt < PDH_MT; ++mt) {
            const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (kb * MT + min(mt, MT - 1)) * 1024, 16);
            __builtin_memcpy(&x[mt][i], &t, 16);
        }
    }
}
template <int KPW32, int OFF, int STRIDE>
__device__ __forceinline__ void pdh_mma(const u32x4 (&x)[PDH_MT][KPW32], const u32x4 (&w)[KPW32], int nkb32, f32x4 (&acc)[PDH_MT]) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#pragma unroll
    for (int i = OFF; i < KPW32; i += STRIDE) {
        if (wave + i * PD_NW < nkb32) {
            bf16x8 bw;
            __builtin_memcpy(&bw, &w[i], 16);
#pragma unroll
            for (int mt = 0; mt < PDH_MT; ++mt) {
                bf16x8 a;
                __builtin_memcpy(&a, &x[mt][i], 16);
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bw, acc[mt], 0, 0, 0);
            }
        }
    }
}
__device__ __forceinline__ void pdh_spill(float* lds, int slab, const f32x4 (&acc)[PDH_MT]) {
    float (*part)[PDH_MT * 16][17] = reinterpret_cast<float (*)[PDH_MT * 16][17]>(lds);
    const int lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < PDH_MT; ++mt)
#pragma unroll
        for (int v = 0; v < 4; ++v) part[slab][mt * 16 + q * 4 + v][r] = acc[mt][v];
}
// elements (row = tid >> 4, col) and (row + 32, col): base + the first NSLAB slabs in ascending order
template <int NSLAB>
__device__ __forceinline__ void pdh_reduce(float* lds, const float (&base)[2], float (&z)[2]) {
    __syncthreads();
    const float (*part)[PDH_MT * 16][17] = reinterpret_cast<const float (*)[PDH_MT * 16][17]>(lds);
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        float v = base[e];
#pragma unroll
        for (int w = 0; w < NSLAB; ++w) v += part[w][row + 32 * e][col];
        z[e] = v;
        asm volatile("" : "+v"(z[e]));      // (one element's reads at a time: with both elements' 32 reads in flight the allocator spilled)
    }
    __syncthreads();
}
// gates (pd_gates_store's arithmetic) of both elements; h leaves as the bf16 mirror only: 4 units = one 8-byte write-through store
__device__ __forceinline__ void pdh_gates_store(const float (&z)[2], float (&c)[2], uint16_t* hh, int tile, int M, int MT) {
    const int col = threadIdx.x & 15;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int row = (threadIdx.x >> 4) + 32 * e;
        const float zf = gt_row_down<4>(z[e]), zg = gt_row_down<8>(z[e]), zo = gt_row_down<12>(z[e]);
        float hv = 0.f;
        if (col < 4 && row < M) {
            const float gi = gt_sigmoid(z[e]), gf = gt_sigmoid(zf), gg = gt_tanh(zg), go = gt_sigmoid(zo);
            c[e] = __builtin_fmaf(gf, c[e], gi * gg);
            hv = go * gt_tanh(c[e]);
        }
        const float h1v = gt_row_down<1>(hv), h2v = gt_row_down<2>(hv), h3v = gt_row_down<3>(hv);
        if (col == 0 && row < M) {
            uint2 pk;
            pk.x = (uint32_t)gt_bf16_bits(hv) | ((uint32_t)gt_bf16_bits(h1v) << 16);
            pk.y = (uint32_t)gt_bf16_bits(h2v) | ((uint32_t)gt_bf16_bits(h3v) << 16);
            pd_st2_sc1(reinterpret_cast<uint2*>(hh + gt_blk_off_h(row, tile * 4, MT)), pk);
        }
    }
}
// sums of a recurrent-half tile: 16-wave order (fragments 0, 2 -> a; 1, 3 -> b; 16 slabs, or two passes over 8) or 8-wave order
template <bool ORDER16>
__device__ __forceinline__ void pdh_rec(const u32x4 (&x)[PDH_MT][4], const u32x4 (&w)[4], float bias, float* lds, bool two, float (&p)[2]) {
    const int wave = threadIdx.x >> 6;
    const float b2[2] = {bias, bias};
    if (ORDER16) {
        f32x4 a[PDH_MT], b[PDH_MT];
#pragma unroll
        for (int mt = 0; mt < PDH_MT; ++mt) { a[mt] = f32x4{0, 0, 0, 0}; b[mt] = f32x4{0, 0, 0, 0}; }
        pdh_mma<4, 0, 2>(x, w, 32, a);
        pdh_mma<4, 1, 2>(x, w, 32, b);
        if (two) {
            float z1[2];
            pdh_spill(lds, wave, a);
            pdh_reduce<8>(lds, b2, z1);
            pdh_spill(lds, wave, b);
            pdh_reduce<8>(lds, z1, p);
        } else {
            pdh_spill(lds, wave, a);
            pdh_spill(lds, wave + 8, b);
            pdh_reduce<16>(lds, b2, p);
        }
    } else {
        f32x4 a[PDH_MT];
#pragma unroll
        for (int mt = 0; mt < PDH_MT; ++mt) a[mt] = f32x4{0, 0, 0, 0};
        pdh_mma<4, 0, 1>(x, w, 32, a);
        pdh_spill(lds, wave, a);
        pdh_reduce<8>(lds, b2, p);
    }
}
struct PdHS { float c1[2], c2[2], p1[2], p2[2]; };

// The chain workgroups' recurrent halves are computed by HELPER workgroups -- plain workgroup i helps chain tile i; at <= 64 rows there
// are more plain workgroups than chains -- from the state fragments the helper holds anyway, and handed back through memory
// (hpart [layer][tile][64 rows][16 columns], one tagged flag per layer and tile).  A chain workgroup's step is then chain -> cell 1 ->
// cell 2 and straight on to the next chain, whose prenet-0 hand-off is what it waits for: the step's critical path no longer carries
// its two recurrent halves (1.4 us each + the wait for the h2 arrivals between them), which the helpers multiply while they would
// otherwise wait for the chains.
constexpr int PDH_HP = PDH_MT * 16 * 16;     // floats of one handed-back tile
__device__ __forceinline__ void pdh_publish(const float (&v)[2], float* dst, uint32_t* flag, uint32_t tag) {
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float v1 = gt_row_down<1>(v[e]), v2 = gt_row_down<2>(v[e]), v3 = gt_row_down<3>(v[e]);
        if ((col & 3) == 0) pd_st4_sc1(dst + (row + 32 * e) * 16 + col, make_float4(v[e], v1, v2, v3));
    }
    pd_drain();
    __syncthreads();
    if (threadIdx.x == 0) pd_st1_sc1(flag, tag);
}
// a chain workgroup's wait for every context flag (<= 64 utterances: one per lane) AND, in the same polls, for its two helpers' flags
__device__ __forceinline__ void pdh_wait_flags_helped(const PersistDecodeArgs& A, const uint32_t* f, uint32_t want, const uint32_t* fh1, const uint32_t* fh2, uint32_t want_h,
                                                      PdShared* sh) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const bool h0 = lane < A.B;
        const uint32_t* p0 = f + (h0 ? lane : 0) * PD_FS;
        const uint32_t* p1 = lane == 0 ? fh1 : fh2;
        uint32_t spins = 0;
        for (;;) {
            uint32_t v0, v1;
            asm volatile("global_load_dword %0, %2, off sc1\n\tglobal_load_dword %1, %3, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(p0), "v"(p1) : "memory");
            const bool ok = (!h0 || v0 >= want) && (lane > 1 || v1 >= want_h);
            if (__builtin_amdgcn_readfirstlane(__popcll(__ballot(ok))) == 64) break;
            if (++spins > PD_SPIN_MAX) { if (lane == 0) pd_give_up(A, sh, true); break; }
            if ((spins & 63u) == 0u && __builtin_amdgcn_readfirstlane(pd_ld_sc1(A.err)) != 0u) { if (lane == 0) pd_give_up(A, sh, false); break; }
        }
    }
    __syncthreads();
}

// HELPED (chain workgroups): the recurrent halves of this step arrive from the helpers; they are requested behind the flag wait and
// land under the fragments and MFMAs
template <bool HELPED>
__device__ __forceinline__ void pdh_cell1(const PersistDecodeArgs& A, const u32x4 (&wx1)[2], int t, int tile, float* lds, PdHS& S, PdShared* sh, int role, int zt) {
    const int par = t & 1, MT = A.MT;
    const uint16_t* xahp = A.xah[par];
    uint16_t* h1hd = A.h1h[par];
    PD_HOLD(xahp); PD_HOLD(h1hd);           // (persist_decode.hip PD_HOLD: no kernel-argument reload behind the wait)
    if (HELPED) pdh_wait_flags_helped(A, A.ctl + zt + PD_F_C, (uint32_t)t + 1u, A.ctl + zt + PD_F_H + tile * 32, A.ctl + zt + PD_F_H + (PDH_BMAX + tile) * 32, (uint32_t)t, sh);
    else pd_wait_flags_all(A, A.ctl + zt + PD_F_C, (uint32_t)t + 1u, sh);
    PD_PHASE_ABORT(sh);
    PD_STAMP(role, 2);
    // (the helpers' sums: requested without a branch -- out of range at step 0 -- and assigned BEHIND the MFMAs: a use inside a branch
    // right here is a wait for them, and for every fragment request that follows in program order, before the first MFMA)
    float hp1[2] = {0.f, 0.f}, hp2[2] = {0.f, 0.f};
    if (HELPED) {
        const auto rh = gt_rsrc(A.hpart, 2u * PDH_BMAX * PDH_HP * 4u);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const uint32_t off = t > 0 ? (uint32_t)((tile * PDH_HP + (threadIdx.x + zt) + 512 * e) * 4) : GT_OOB;
            hp1[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rh, (int)off, 0, 16));
            hp2[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rh, (int)off, PDH_BMAX * PDH_HP * 4, 16));
        }
    }
    u32x4 x[PDH_MT][2];
    pdh_xload<2>(xahp, MT, PD_KBP / 2 + PD_KBC / 2, 0, x);
    PD_PIN();
    f32x4 a[PDH_MT];
#pragma unroll
    for (int mt = 0; mt < PDH_MT; ++mt) a[mt] = f32x4{0, 0, 0, 0};
    pdh_mma<2, 0, 1>(x, wx1, PD_KBP / 2 + PD_KBC / 2, a);
    if (HELPED && t > 0) {
#pragma unroll
        for (int e = 0; e < 2; ++e) { S.p1[e] = hp1[e]; S.p2[e] = hp2[e]; }
    }
    pdh_spill(lds, threadIdx.x >> 6, a);
    float z[2];
    pdh_reduce<8>(lds, S.p1, z);
    pdh_gates_store(z, S.c1, h1hd, tile + zt, A.B, MT);
    pd_arrive(A.ctl + zt + PD_CNT3);
    PD_STAMP(role, 3);
}
// cell 2; the h1 fragments stay in `x` for the recurrent halves that follow (pdh_rec1)
__device__ __forceinline__ void pdh_cell2(const PersistDecodeArgs& A, const u32x4 (&wx2)[4], u32x4 (&x)[PDH_MT][4], int t, int tile, float* lds, PdHS& S, PdShared* sh, int role,
                                          int zt) {
    const int par = t & 1, MT = A.MT;
    const uint16_t* h1hp = A.h1h[par];
    uint16_t* h2hd = A.h2h[par];
    PD_HOLD(h1hp); PD_HOLD(h2hd);
    pd_wait_count(A, A.ctl + zt + PD_CNT3, PD_WANT(A, t), sh);
    PD_PHASE_ABORT(sh);
    PD_STAMP(role, 4);
    pdh_xload<4>(h1hp, MT, PD_KBH / 2, 0, x);
    PD_PIN();
    f32x4 a[PDH_MT];
#pragma unroll
    for (int mt = 0; mt < PDH_MT; ++mt) a[mt] = f32x4{0, 0, 0, 0};
    pdh_mma<4, 0, 1>(x, wx2, PD_KBH / 2, a);
    pdh_spill(lds, threadIdx.x >> 6, a);
    float z[2];
    pdh_reduce<8>(lds, S.p2, z);
    pdh_gates_store(z, S.c2, h2hd, tile + zt, A.B, MT);
    pd_arrive(A.ctl + zt + PD_CNT4);
    PD_STAMP(role, 5);
}
// layer-1 recurrent half of tile `tl` for the next step from the fragments cell 2 left in `x` (own tile: into `p`; a chain tile's: published)
__device__ __forceinline__ void pdh_rec1(const PersistDecodeArgs& A, const u32x4 (&x)[PDH_MT][4], const u32x4 (&wh1)[4], int tl, float* lds, bool two, float (&p)[2]) {
    pdh_rec<true>(x, wh1, A.b1h[tl * 16 + (threadIdx.x & 15)], lds, two, p);
}
// layer-2 recurrent half for the next step: this workgroup's tile and, HELP, chain tile `help` (published)
template <bool WAIT, bool HELP>
__device__ __forceinline__ void pdh_rec2(const PersistDecodeArgs& A, u32x4 (&wh2)[4], int t, int tile, float* lds, PdHS& S, PdShared* sh, int role, int zt, int help) {
    if (HELP) pdh_load_tile<4>(A.w2h, tile, PD_KBH / 2, wh2, zt);  // (a helper streams its own W2h: arrives during the wait)
    if (WAIT) {
        pd_wait_count(A, A.ctl + zt + PD_CNT4, PD_WANT(A, t), sh);
        PD_PHASE_ABORT(sh);
    }
    PD_STAMP(role, 7);
    const int col = threadIdx.x & 15;
    // (the summation order is a property of the tile.  Each arm loads its own fragments: shared between the arms of a branch, the
    // allocator gave up on packing them and spilled -- the group kernels' pd_g_rec_all met the same)
    if (tile < A.co_tiles) {
        u32x4 x[PDH_MT][4];
        pdh_xload<4>(A.h2h[t & 1], A.MT, PD_KBH / 2, 0, x);
        PD_PIN();
        pdh_rec<false>(x, wh2, A.b2h[tile * 16 + col], lds, false, S.p2);
    } else {
        u32x4 x[PDH_MT][4];
        pdh_xload<4>(A.h2h[t & 1], A.MT, PD_KBH / 2, 0, x);
        PD_PIN();
        pdh_rec<true>(x, wh2, A.b2h[tile * 16 + col], lds, false, S.p2);
    }
    PD_STAMP(role, 8);
    if (HELP) {         // (behind the own half: the chain workgroup needs it at its next cell 2, a whole chain away)
        pdh_load_tile<4>(A.w2h, help, PD_KBH / 2, wh2, zt);
        float v[2];
        if (help < A.co_tiles) {
            u32x4 x[PDH_MT][4];
            pdh_xload<4>(A.h2h[t & 1], A.MT, PD_KBH / 2, 0, x);
            PD_PIN();
            pdh_rec<false>(x, wh2, A.b2h[help * 16 + col], lds, false, v);
        } else {
            u32x4 x[PDH_MT][4];
            pdh_xload<4>(A.h2h[t & 1], A.MT, PD_KBH / 2, 0, x);
            PD_PIN();
            pdh_rec<true>(x, wh2, A.b2h[help * 16 + col], lds, false, v);
        }
        pdh_publish(v, A.hpart + (size_t)(PDH_BMAX + help) * PDH_HP + zt, A.ctl + zt + PD_F_H + (PDH_BMAX + help) * 32, (uint32_t)t + 1u);
    }
}
// projection tile `ptile`, M-tile `pmt` from the mirrors of h2 (32-k blocks 0..31) and of the context (the last 4 of xa's 12)
// (h2hp / xahp / z0g: this step's mirrors and the granule buffer, held in scalar registers by the caller in front of its wait)
__device__ __forceinline__ void pdh_proj(const PersistDecodeArgs& A, const u32x4 (&wp)[5], int t, int ptile, int pmt, float* lds, const uint16_t* h2hp, const uint16_t* xahp,
                                         uint2* z0g) {
    const int MT = A.MT;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const auto rh = gt_rsrc(h2hp, 0x7FFFF000u);
    const auto rx = gt_rsrc(xahp, 0x7FFFF000u);
    constexpr int NKB = PD_KBPJ / 2;
    u32x4 x[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int kb = (wave + i * PD_NW < NKB) ? wave + i * PD_NW : wave;          // wave-uniform
        const auto tv = kb < PD_KBH / 2 ? __builtin_amdgcn_raw_buffer_load_b128(rh, lane * 16, (kb * MT + pmt) * 1024, 16)
                                        : __builtin_amdgcn_raw_buffer_load_b128(rx, lane * 16, ((kb - PD_KBH / 2 + PD_KBP / 2) * MT + pmt) * 1024, 16);
        __builtin_memcpy(&x[i], &tv, 16);
    }
    PD_PIN();
    f32x4 a0 = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        if (wave + i * PD_NW < NKB) {
            bf16x8 av, bw;
            __builtin_memcpy(&av, &x[i], 16);
            __builtin_memcpy(&bw, &wp[i], 16);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bw, a0, 0, 0, 0);
        }
    }
    const f32x4 zero = {0, 0, 0, 0};
    pd_spill(lds, threadIdx.x >> 6, a0, zero);
    const float v = pd_reduce<8>(lds, A.bp[ptile * 16 + (threadIdx.x & 15)]);
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
}

// CHAIN: utterance blockIdx.x's chain, then cell 1 and cell 2 of its tile with W1x / W2x streamed (its recurrent halves come from a
// helper); else a plain workgroup: its tile's four halves with resident weights and, HELP, the recurrent halves of chain tile `help`
template <bool CHAIN, bool HELP>
__device__ __forceinline__ void pdh_run_tile(const PersistDecodeArgs& A, float* smem, PdShared* sh, int help) {
    float* lds = smem;
    const int tile = blockIdx.x, b = blockIdx.x, tid = threadIdx.x, col = tid & 15;
    constexpr int role = CHAIN ? 0 : 2;
    PdWh W;
    PdHS S;
#pragma unroll
    for (int e = 0; e < 2; ++e) { S.c1[e] = 0.f; S.c2[e] = 0.f; S.p1[e] = A.b1h[tile * 16 + col]; S.p2[e] = A.b2h[tile * 16 + col]; }
    PdChainLds L{};
    PdChainRegs R{};
    if (CHAIN) {
        L = pd_carve(smem, A.tvp, 8, PDH_SLAB);
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
        // (a helper keeps the input halves resident and streams the recurrent tiles -- its own, then the chain tile's, through the same
        // registers: four resident tiles and a streamed fifth did not fit beside the 64-row fragments)
        pdh_load_tile<2>(A.w1x, tile, PD_KBP / 2 + PD_KBC / 2, W.x1); pdh_load_tile<4>(A.w2x, tile, PD_KBH / 2, W.x2);
        if (!HELP) { pdh_load_tile<4>(A.w1h, tile, PD_KBH / 2, W.h1); pdh_load_tile<4>(A.w2h, tile, PD_KBH / 2, W.h2); }
    }
    __syncthreads();
    for (int t = 0; t < A.steps; ++t) {
        if (sh->abort) return;          // (the one abort check of the step: PD_PHASE_ABORT)
        PD_STAMP(role, 0);
        PD_ZT(zt);
        if (CHAIN) {
            float4 unused4;
            int bs = b;
            asm volatile("" : "+s"(bs));
            pd_chain<false, true>(A, L, R, t, bs, sh, unused4, zt);
            PD_PHASE_ABORT(sh);
            PD_STAMP(role, 1);
            PD_PIN();
            pdh_load_tile<2>(A.w1x, tile, PD_KBP / 2 + PD_KBC / 2, W.x1, zt);
            pdh_load_tile<4>(A.w2x, tile, PD_KBH / 2, W.x2, zt);    // (arrives during cell 1)
        }
        pdh_cell1<CHAIN>(A, W.x1, t, tile, lds, S, sh, role, zt);
        PD_PHASE_ABORT(sh);
        if (HELP && t + 1 < A.steps) pdh_load_tile<4>(A.w1h, tile, PD_KBH / 2, W.h1, zt);  // (streamed: arrives during the wait for h1)
        u32x4 x[PDH_MT][4];
        pdh_cell2(A, W.x2, x, t, tile, lds, S, sh, role, zt);
        PD_PHASE_ABORT(sh);
        if (CHAIN) continue;                          // (a chain workgroup goes straight on to the next chain)
        if (t + 1 == A.steps) break;
        pdh_rec1(A, x, W.h1, tile, lds, false, S.p1);
        PD_STAMP(role, 6);
        if (HELP) {         // (the chain workgroup needs it at its NEXT cell 1, a whole chain away: own half first)
            pdh_load_tile<4>(A.w1h, help, PD_KBH / 2, W.h1, zt);
            float v[2];
            pdh_rec1(A, x, W.h1, help, lds, false, v);
            pdh_publish(v, A.hpart + (size_t)help * PDH_HP + zt, A.ctl + zt + PD_F_H + help * 32, (uint32_t)t + 1u);
        }
        pdh_rec2<true, HELP>(A, W.h2, t, tile, lds, S, sh, role, zt, help);
        PD_PHASE_ABORT(sh);
    }
}

// a projection (tile, M-tile) + the LSTM tile, everything resident; the projection comes FIRST behind the h2 arrivals (the chains wait
// for it), the recurrent halves of layer 1 (h1 re-read) and of layer 2 behind it
__device__ __forceinline__ void pdh_run_proj(const PersistDecodeArgs& A, float* lds, PdShared* sh) {
    const int tile = blockIdx.x, col = threadIdx.x & 15;
    const int pi = tile - A.n_chain, ptile = pi % A.pj_tiles, pmt = pi / A.pj_tiles;
    PdWh W;
    pdh_load_tile<2>(A.w1x, tile, PD_KBP / 2 + PD_KBC / 2, W.x1); pdh_load_tile<4>(A.w2x, tile, PD_KBH / 2, W.x2);
    pdh_load_tile<4>(A.w1h, tile, PD_KBH / 2, W.h1); pdh_load_tile<4>(A.w2h, tile, PD_KBH / 2, W.h2);
    u32x4 wpj[5];
    pdh_load_tile<5>(A.wp, ptile, PD_KBPJ / 2, wpj);
    PdHS S;
#pragma unroll
    for (int e = 0; e < 2; ++e) { S.c1[e] = 0.f; S.c2[e] = 0.f; S.p1[e] = A.b1h[tile * 16 + col]; S.p2[e] = A.b2h[tile * 16 + col]; }
    for (int t = 0; t < A.steps; ++t) {
        if (sh->abort) return;          // (the one abort check of the step: PD_PHASE_ABORT)
        PD_STAMP(1, 0);
        PD_ZT(zt);
        pdh_cell1<false>(A, W.x1, t, tile, lds, S, sh, 1, zt);
        PD_PHASE_ABORT(sh);
        {
            u32x4 x[PDH_MT][4];
            pdh_cell2(A, W.x2, x, t, tile, lds, S, sh, 1, zt);
            PD_PHASE_ABORT(sh);
        }
        const uint16_t* h2hp = A.h2h[t & 1];
        const uint16_t* xahp = A.xah[t & 1];
        uint2* z0g = A.z0g;
        PD_HOLD(h2hp); PD_HOLD(xahp); PD_HOLD(z0g);
        pd_wait_count(A, A.ctl + zt + PD_CNT4, PD_WANT(A, t), sh);
        PD_PHASE_ABORT(sh);
        pdh_proj(A, wpj, t, ptile, pmt, lds, h2hp, xahp, z0g);
        PD_STAMP(1, 6);
        if (t + 1 == A.steps) break;
        {   // (the h1 fragments again: kept in registers across the projection they cost this role spills; nobody waits for this half)
            u32x4 x[PDH_MT][4];
            pdh_xload<4>(A.h1h[t & 1], A.MT, PD_KBH / 2, 0, x);
            PD_PIN();
            pdh_rec1(A, x, W.h1, tile, lds, false, S.p1);
        }
        pdh_rec2<false, false>(A, W.h2, t, tile, lds, S, sh, 1, zt, -1);
    }
}

__global__ __launch_bounds__(PD_NT) void gt_persist_decode_h_kernel(PersistDecodeArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ PdShared sh;
    if (threadIdx.x == 0) sh.abort = 0;
    __syncthreads();
    const int tile = blockIdx.x;
    const int n_pj = A.pj_tiles * A.MT;
#ifndef PD_HONLY
#define PD_HONLY -1         // (register-budget diagnosis: compile one role alone)
#endif
    if (tile < A.n_chain) { if (PD_HONLY < 0 || PD_HONLY == 0) pdh_run_tile<true, false>(A, smem, &sh, -1); }
    else if (tile < A.n_chain + n_pj) { if (PD_HONLY < 0 || PD_HONLY == 1) pdh_run_proj(A, smem, &sh); }
    else if (tile - (A.n_chain + n_pj) < A.n_chain) {       // (helps chain tile `index among the plain`)
        if (PD_HONLY < 0 || PD_HONLY == 2) pdh_run_tile<false, true>(A, smem, &sh, tile - (A.n_chain + n_pj));
    } else if (PD_HONLY < 0 || PD_HONLY == 3) pdh_run_tile<false, false>(A, smem, &sh, -1);
}
