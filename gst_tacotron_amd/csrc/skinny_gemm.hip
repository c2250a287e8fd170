// Skinny (batch-rows) fp32 GEMM on MFMA with fused epilogues -- the decode-loop workhorse.
//
//   Z[M, 16*ntiles] = concat(seg0, seg1, seg2)[M, K] . W[K, 16*ntiles] + bias      (M = batch rows)
//
// Replaces, per decoder step (reference Modules/Taco2.py:96-120):
//   Prenet Dense+ReLU+Dropout (Taco2.py:262-283)            -> EPI_RELU_DROP
//   attention Query Dense (Steps.py:122)                    -> EPI_LINEAR
//   LSTMCell x2 via StackedRNNCells (Taco2.py:77-85,111)    -> EPI_LSTM (gates fused, Appendix A.6)
//   Projection Dense + split mel/stop (Taco2.py:112-118)    -> EPI_LINEAR with a split column
// and per encoder time step the two directions of the BiLSTM (Taco2.py:39-43) -> EPI_LSTM, grid.z = 2.
//
// Design (gfx950): one workgroup owns 16 output columns (for an LSTM: 4 hidden units x 4 gates, so the
// gate non-linearity and the cell update never leave the CU) and ALL batch rows (<=32 per blockIdx.y);
// K is split over the workgroup's waves and reduced through LDS.  Weights are repacked at finalize into
// MFMA-fragment order [tile][k-block of 16][lane][4] so a wave streams them with one 1-KiB coalesced
// global_load_dwordx4 per k-block and never touches LDS for them (weights are used once per step --
// cdna_hip_programming.md 'GEMV / M<=16 decode weights: load straight to VGPRs').  The 4 values a lane
// loads are the B operands of 4 consecutive v_mfma_f32_16x16x4_f32; the matching A operands (activations)
// are 4 consecutive k of one batch row = one 16-byte load.  fp32 in / fp32 accumulate (exact fp32 FMA
// chain), since the parity bar is 1e-3 through a 1000-frame recurrence.
#include "skinny_body.h"
#include "lean_body.h"

// TAG only separates kernel symbols per call site (decode LSTM layer 1 / 2, encoder BiLSTM) so that
// rocprofv3 --stats reports them on separate lines.
template <int EPI, int NW, int TAG>
__global__ __launch_bounds__(NW * 64) void gt_skinny_kernel(SkinnyArgs a0, SkinnyArgs a1) {
    const SkinnyArgs& A = (blockIdx.z == 0) ? a0 : a1;
    __shared__ __attribute__((aligned(16))) float lds[SkinnyLds<NW>::kFloats];
    // weights of the big decode LSTM GEMMs are read by exactly one CU once per step: stream them non-temporally
    constexpr bool NT = (TAG == TAG_DEC_LSTM1 || TAG == TAG_DEC_LSTM2);
    gt_skinny_body<EPI, NW, NT>(A, blockIdx.x, blockIdx.y, lds);
}

// Projection (EPI_LINEAR, 11 workgroups at 161 columns) co-scheduled with recurrent-half partial GEMM tiles, CT tiles
// per worker workgroup (CT = 2: one pass over the activations for both, gt_skinny_partial_multi).
template <int NW, int CT>
__global__ __launch_bounds__(NW * 64) void gt_skinny_co_kernel(SkinnyArgs main_args, SkinnyArgs co, int n_main, int co_begin, int co_end) {
    constexpr int kLds = SkinnyLds<NW>::kFloats > SkinnyMultiLds<NW, CT>::kFloats ? SkinnyLds<NW>::kFloats : SkinnyMultiLds<NW, CT>::kFloats;
    __shared__ __attribute__((aligned(16))) float lds[kLds];
    const int mchunks = (main_args.M + 31) / 32;
    if ((int)blockIdx.x < n_main) {
        for (int mc = 0; mc < mchunks; ++mc) {
            gt_skinny_body<EPI_LINEAR, NW, false>(main_args, blockIdx.x, mc, lds);
            __syncthreads();
        }
    } else {
        const int tile = co_begin + ((int)blockIdx.x - n_main) * CT;
        for (int mc = 0; mc < mchunks; ++mc) {
            if (CT == 1) gt_skinny_body<EPI_PARTIAL, NW, true>(co, tile, mc, lds);
            else gt_skinny_partial_multi<NW, CT, true>(co, tile, min(CT, co_end - tile), mc, lds);
            __syncthreads();
        }
    }
}

hipError_t gt_launch_skinny_co(const SkinnyArgs& main_args, int ntiles, const SkinnyArgs& co, int co_begin, int co_end,
                               int tiles_per_worker, hipStream_t stream) {
    const int nco = co_end > co_begin ? co_end - co_begin : 0;
    if (tiles_per_worker == 2)
        hipLaunchKernelGGL((gt_skinny_co_kernel<8, 2>), dim3(ntiles + (nco + 1) / 2), dim3(512), 0, stream, main_args, co, ntiles, co_begin, co_end);
    else
        hipLaunchKernelGGL((gt_skinny_co_kernel<8, 1>), dim3(ntiles + nco), dim3(512), 0, stream, main_args, co, ntiles, co_begin, co_end);
    return hipGetLastError();
}

template <int EPI, int TAG>
static hipError_t launch_epi(const SkinnyArgs& a0, const SkinnyArgs* a1, int ntiles, hipStream_t stream) {
    dim3 grid(ntiles, (a0.M + 31) / 32, a1 ? 2 : 1);
    const SkinnyArgs& b = a1 ? *a1 : a0;
    // K-split width: 8 waves when the K loop is long enough to feed them (24: the decode LSTM-1 input half, same split as
    // its lean kernel so the two stay bitwise equal)
    if (a0.nkb >= 24) {
        hipLaunchKernelGGL((gt_skinny_kernel<EPI, 8, TAG>), grid, dim3(512), 0, stream, a0, b);
    } else {
        hipLaunchKernelGGL((gt_skinny_kernel<EPI, 4, TAG>), grid, dim3(256), 0, stream, a0, b);
    }
    return hipGetLastError();
}

hipError_t gt_launch_skinny(int epi, const SkinnyArgs& a0, const SkinnyArgs* a1, int ntiles, hipStream_t stream, int tag) {
    switch (epi) {
        case EPI_LINEAR: return launch_epi<EPI_LINEAR, 0>(a0, a1, ntiles, stream);
        case EPI_RELU_DROP: return launch_epi<EPI_RELU_DROP, 0>(a0, a1, ntiles, stream);
        case EPI_LSTM:
            switch (tag) {
                case TAG_DEC_LSTM1: return launch_epi<EPI_LSTM, TAG_DEC_LSTM1>(a0, a1, ntiles, stream);
                case TAG_DEC_LSTM2: return launch_epi<EPI_LSTM, TAG_DEC_LSTM2>(a0, a1, ntiles, stream);
                default: return launch_epi<EPI_LSTM, TAG_ENC_BILSTM>(a0, a1, ntiles, stream);
            }
    }
    return hipErrorInvalidValue;
}

// ======================================================================================================================
// Lean decode-step kernels (see lean_body.h for why they exist)
// ======================================================================================================================

// z[32 rows x 16 gate columns] of tile `blockIdx.x` = x . W_x + partial_in; tile-local column g*4+u is gate g (i,f,c~,o)
// of hidden unit tile*4+u (Appendix A.6; reference Taco2.py:79-85 via StackedRNNCells).  The 16 columns of a row sit in
// 16 adjacent lanes, so the lanes with column < 4 collect their unit's four gates with three lane shifts -- no second LDS
// round trip or barrier.
template <int NW, int KPW, int TAG, bool BF16>
__global__ __launch_bounds__(NW * 64) void gt_lstm_x_kernel(LstmXArgs A) {
    __shared__ __attribute__((aligned(16))) float lds[LeanLds<NW, 1>::kFloats];
    constexpr int NE = 512 / (NW * 64);
    const int tile = blockIdx.x, mchunk = blockIdx.y;
    const int m0 = mchunk * 32, MT = A.MT;
    GT_STAMP(A.dbg, 4);
    float pin[NE], c_prev[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = threadIdx.x + i * NW * 64;
        const int row = e >> 4, col = e & 15;
        const int grow = m0 + row, unit = tile * 4 + col;
        pin[i] = (grow < MT * 16) ? A.partial_in[((size_t)tile * MT * 16 + grow) * 16 + col] : 0.f;
        c_prev[i] = (col < 4 && grow < A.M && unit < A.H) ? A.c[(size_t)grow * A.H + unit] : 0.f;
    }
    GT_STAMP(A.dbg, 0);
    f32x4 acc0[1] = {f32x4{0.f, 0.f, 0.f, 0.f}}, acc1[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    if constexpr (BF16) {
        gt_lean_core_bf16<NW, KPW, 1, false>(A.wp, tile, 1, LeanX{A.x, A.x, A.nkb}, MT, mchunk, (A.nkb + 1) >> 1, acc0, acc1);
    } else {
        gt_lean_core<NW, KPW, 1, false>(A.wp, tile, 1, LeanX{A.x, A.x, NW * KPW}, MT, mchunk, acc0, acc1);
    }
    GT_STAMP(A.dbg, 1);
    gt_lean_spill<NW, 1>(lds, acc0, acc1);
    __syncthreads();
    GT_STAMP(A.dbg, 2);
    const float (*part)[32][17] = reinterpret_cast<const float (*)[32][17]>(lds);     // [NW][32][17]
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = threadIdx.x + i * NW * 64;
        const int row = e >> 4, col = e & 15;
        float z = pin[i];
#pragma unroll
        for (int w = 0; w < NW; ++w) z += part[w][row][col];
        const float zf = gt_row_down<4>(z), zg = gt_row_down<8>(z), zo = gt_row_down<12>(z);
        const int grow = m0 + row, unit = tile * 4 + col;
        if (col < 4 && grow < A.M && unit < A.H) {
            if (A.row_len && A.t_index >= A.row_len[grow]) {
                A.h[gt_blk_off(grow, unit, MT)] = 0.f;          // masked mode: this step does not exist for this utterance
            } else {
                const float gi = gt_sigmoid(z), gf = gt_sigmoid(zf), gg = gt_tanh(zg), go = gt_sigmoid(zo);
                const float c2 = __builtin_fmaf(gf, c_prev[i], gi * gg);
                A.c[(size_t)grow * A.H + unit] = c2;
                A.h[gt_blk_off(grow, unit, MT)] = go * gt_tanh(c2);
            }
        }
    }
    GT_STAMP(A.dbg, 3);
}

__device__ __forceinline__ uint32_t gt_ldu_sc1(const uint32_t* p) {
    uint32_t v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// ======================================================================================================================
// Both decode LSTM cells in ONE launch (fp32, batch <= 32): workgroup = tile `blockIdx.x` of layer 1, then of layer 2.
//
// What it buys (tools/persist_phase.hip, measured on MI355X): an LSTM-2-shaped phase costs 6.3-6.9 us as a launch of its own
// -- kernel boundary, then 64 KB of weights per CU from the Infinity Cache, then the 128 KB state -- and 5.0 us when the
// weights are already in registers and the boundary is an in-kernel hand-off.  Here the layer-2 tile's weights, bias-side
// partial sums and cell state are REQUESTED BEFORE the workgroup starts waiting for layer 1 (they depend on nothing this
// launch computes), so the wait for the other 255 tiles' h1 hides their latency; h1 itself is stored write-through and read
// with sc1 loads (gt_xload), the arrival counter is sharded over 8 cache lines and polled with sc1 loads.
// Arithmetic = gt_lstm_x_kernel<8,3> followed by gt_lstm_x_kernel<8,8>, same orders: bitwise equal states.
//
// The hand-off needs all of the launch's workgroups resident together: the host checks the grid against occupancy x CUs at
// finalize (gt_lstm12_blocks_per_cu) and only takes this path while the process has ONE live context and no other context's
// fused launches are in flight (several decode loops could each hold part of the chip and wait for the rest).  The wait is bounded all the same: a give-up raises the
// host-mapped error word (gsttaco_synchronize reports it, the next call falls back to two launches).
// ======================================================================================================================
__device__ __forceinline__ void gt_sth_sc1(uint16_t* p, const uint16_t v) {
    asm volatile("global_store_short %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"((uint32_t)v) : "memory");
}
__device__ __forceinline__ void gt_st1_sc1(float* p, float v) {
    asm volatile("global_store_dword %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

__global__ __launch_bounds__(512) void gt_lstm12_kernel(Lstm12Args P) {
    constexpr int NW = 8;
    __shared__ __attribute__((aligned(16))) float lds[LeanLds<NW, 1>::kFloats];
    __shared__ int s_abort;
    const int tile = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    const float (*part)[32][17] = reinterpret_cast<const float (*)[32][17]>(lds);
    GT_STAMP(P.l1.dbg, 4);
    if (threadIdx.x == 0) s_abort = 0;
    // ---------------------------------------------------------------- layer 1 (gt_lstm_x_kernel<8, 3>)
    {
        const LstmXArgs& A = P.l1;
        const int MT = A.MT, unit = tile * 4 + col;
        const float pin = (row < MT * 16) ? A.partial_in[((size_t)tile * MT * 16 + row) * 16 + col] : 0.f;
        const float c_prev = (col < 4 && row < A.M && unit < A.H) ? A.c[(size_t)row * A.H + unit] : 0.f;
        GT_STAMP(A.dbg, 0);
        f32x4 acc0[1] = {f32x4{0.f, 0.f, 0.f, 0.f}}, acc1[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
        gt_lean_core<NW, 3, 1, false>(A.wp, tile, 1, LeanX{A.x, A.x, NW * 3}, MT, 0, acc0, acc1);
        GT_STAMP(A.dbg, 1);
        gt_lean_spill<NW, 1>(lds, acc0, acc1);
        __syncthreads();
        GT_STAMP(A.dbg, 2);
        float z = pin;
#pragma unroll
        for (int w = 0; w < NW; ++w) z += part[w][row][col];
        const float zf = gt_row_down<4>(z), zg = gt_row_down<8>(z), zo = gt_row_down<12>(z);
        if (col < 4 && row < A.M && unit < A.H) {
            const float gi = gt_sigmoid(z), gf = gt_sigmoid(zf), gg = gt_tanh(zg), go = gt_sigmoid(zo);
            const float c2 = __builtin_fmaf(gf, c_prev, gi * gg);
            A.c[(size_t)row * A.H + unit] = c2;
            gt_st1_sc1(A.h + gt_blk_off(row, unit, MT), go * gt_tanh(c2));     // write-through: read by every other workgroup below
        }
        GT_STAMP(A.dbg, 3);
    }
    // ---------------------------------------------------------------- layer 2 (gt_lstm_x_kernel<8, 8>)
    const LstmXArgs& A = P.l2;
    const int MT = A.MT, unit = tile * 4 + col;
    // arrive first: this workgroup's part of h1 is out once its stores are acknowledged (nothing else is in flight yet)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&P.arrive[(blockIdx.x & (GT_L12_NSH - 1)) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // ... then everything of layer 2 that does not depend on layer 1 is requested and arrives during the wait
    const float pin = (row < MT * 16) ? A.partial_in[((size_t)tile * MT * 16 + row) * 16 + col] : 0.f;
    const float c_prev = (col < 4 && row < A.M && unit < A.H) ? A.c[(size_t)row * A.H + unit] : 0.f;
    constexpr int KPW = 8, NKB = NW * KPW;
    float4 b[KPW];
    {
        const float4* wl = reinterpret_cast<const float4*>(A.wp) + ((size_t)tile * NKB + wave) * 64 + lane;
#pragma unroll
        for (int i = 0; i < KPW; ++i) b[i] = wl[(size_t)i * NW * 64];
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (threadIdx.x < 64) {
        uint32_t spins = 0;
        for (;;) {
            uint32_t v = lane < GT_L12_NSH ? gt_ldu_sc1(P.arrive + lane * 32) : 0u;
            v = gt_row_sum_u32<GT_L12_NSH>(v);
            if (__builtin_amdgcn_readfirstlane(v) >= P.expect) break;
            ++spins;
            if (spins > (1u << 18)) { if (lane == 0) { atomicOr(P.err, 1u); s_abort = 1; } break; }
            if ((spins & 63u) == 0u && __builtin_amdgcn_readfirstlane(gt_ldu_sc1(P.err)) != 0u) { if (lane == 0) s_abort = 1; break; }
        }
    }
    __syncthreads();
    if (s_abort) return;
    GT_STAMP(A.dbg, 4);
    const LeanX X{A.x, A.x, NKB};
    const LeanXR XR = gt_x_rsrc(X);
    float4 x0[KPW], x1[KPW];
    const int mt1 = min(1, MT - 1);
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        x0[i] = gt_xload(XR, X, wave + i * NW, MT, 0);
        x1[i] = gt_xload(XR, X, wave + i * NW, MT, mt1);
    }
    GT_STAMP(A.dbg, 0);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].x, b[i].x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].x, b[i].x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].y, b[i].y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].y, b[i].y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].z, b[i].z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].z, b[i].z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].w, b[i].w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].w, b[i].w, acc1, 0, 0, 0);
    }
    GT_STAMP(A.dbg, 1);
    {
        f32x4 a0[1] = {acc0}, a1[1] = {acc1};
        gt_lean_spill<NW, 1>(lds, a0, a1);          // (layer 1's partial sums were read before the arrival barrier)
    }
    __syncthreads();
    GT_STAMP(A.dbg, 2);
    float z = pin;
#pragma unroll
    for (int w = 0; w < NW; ++w) z += part[w][row][col];
    const float zf = gt_row_down<4>(z), zg = gt_row_down<8>(z), zo = gt_row_down<12>(z);
    if (col < 4 && row < A.M && unit < A.H) {
        const float gi = gt_sigmoid(z), gf = gt_sigmoid(zf), gg = gt_tanh(zg), go = gt_sigmoid(zo);
        const float c2 = __builtin_fmaf(gf, c_prev, gi * gg);
        A.c[(size_t)row * A.H + unit] = c2;
        A.h[gt_blk_off(row, unit, MT)] = go * gt_tanh(c2);
    }
    GT_STAMP(A.dbg, 3);
}

// `slots`: workgroups of the kernel the device holds at once = occupancy (gt_lstm12_blocks_per_cu) x compute units.  The in-kernel
// hand-off needs the whole grid resident.
bool gt_lstm12_supported(int nkb1, int nkb2, int H1, int H2, int M, int slots) {
    return nkb1 == 24 && nkb2 == 64 && H1 == H2 && H1 % 4 == 0 && M <= 32 && H1 / 4 <= slots;
}

hipError_t gt_launch_lstm12(const Lstm12Args& a, hipStream_t stream) {
    hipLaunchKernelGGL(gt_lstm12_kernel, dim3((a.l1.H + 3) / 4), dim3(512), 0, stream, a);
    return hipGetLastError();
}

// (pair of tiles, chunks [c0, c1)) of one decode LSTM cell.  WT: h is stored write-through (read by other workgroups of the same
// launch); PRE: the pair's weights were requested by the caller (LeanW W).
template <int NW, int KPW, bool BF16, bool WT, bool PRE>
__device__ __forceinline__ void gt_lstm_x_mc_body(const LstmXArgs& A, const int tile0, const int ntile, const int c0, const int c1, float* lds,
                                                  LeanW<KPW, 2, BF16>& W) {
    const int MT = A.MT;
    const float (*part)[NW][32][17] = reinterpret_cast<const float (*)[NW][32][17]>(lds);
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    float pin[2], c_prev[2];
    auto pre = [&](const int mc) {
        const int grow = mc * 32 + row;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int tile = tile0 + (j < ntile ? j : 0), unit = tile * 4 + col;
            // (rows / units beyond the matrix read an element that exists and are never used: no load under a branch)
            pin[j] = A.partial_in[((size_t)tile * MT * 16 + min(grow, MT * 16 - 1)) * 16 + col];
            c_prev[j] = A.c[(size_t)min(grow, A.M - 1) * A.H + min(unit, A.H - 1)];
        }
    };
    auto epi = [&](const int mc) {
        const int grow = mc * 32 + row;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float z = pin[j];
#pragma unroll
            for (int w = 0; w < NW; ++w) z += part[j][w][row][col];
            const float zf = gt_row_down<4>(z), zg = gt_row_down<8>(z), zo = gt_row_down<12>(z);
            const int unit = (tile0 + j) * 4 + col;
            if (j < ntile && col < 4 && grow < A.M && unit < A.H) {
                float hv = 0.f;                                     // masked mode: a step that does not exist for this utterance writes 0
                if (!(A.row_len && A.t_index >= A.row_len[grow])) {
                    const float gi = gt_sigmoid(z), gf = gt_sigmoid(zf), gg = gt_tanh(zg), go = gt_sigmoid(zo);
                    const float c2 = __builtin_fmaf(gf, c_prev[j], gi * gg);
                    A.c[(size_t)grow * A.H + unit] = c2;
                    hv = go * gt_tanh(c2);
                }
                if (WT) gt_st1_sc1(A.h + gt_blk_off(grow, unit, MT), hv);
                else A.h[gt_blk_off(grow, unit, MT)] = hv;
                if (A.hh) {                                          // bf16 mirror for the next GEMMs' bf16 bodies
                    if (WT) gt_sth_sc1(A.hh + gt_blk_off_h(grow, unit, MT), gt_bf16_bits(hv));
                    else A.hh[gt_blk_off_h(grow, unit, MT)] = gt_bf16_bits(hv);
                }
            }
        }
    };
    const LeanX X{A.x, A.x, BF16 ? A.nkb : NW * KPW, A.xh, A.xh};
    if (BF16 && A.xh) gt_lean_mc_impl<NW, KPW, 2, BF16, false, PRE, BF16>(A.wp, tile0, ntile, X, (A.nkb + 1) >> 1, MT, c0, c1, lds, pre, epi, A.dbg, W);
    else gt_lean_mc_impl<NW, KPW, 2, BF16, false, PRE, false>(A.wp, tile0, ntile, X, (A.nkb + 1) >> 1, MT, c0, c1, lds, pre, epi, A.dbg, W);
}

// which (pair, chunk range) a workgroup of the 1-D grid owns: the two workgroups of a pair sit 8 block indices apart
__device__ __forceinline__ bool gt_lstm_mc_job(const LstmXArgs& A, int& tile0, int& ntile, int& c0, int& c1) {
    const int ntiles = (A.H + 3) / 4, npairs = (ntiles + 1) / 2;
    const int pair = ((int)blockIdx.x >> 4) * 8 + ((int)blockIdx.x & 7), rpart = ((int)blockIdx.x >> 3) & 1;
    tile0 = pair * 2; ntile = min(2, ntiles - tile0);
    const int mchunks = (A.M + 31) / 32, csplit = (mchunks + 1) / 2;
    c0 = rpart == 0 ? 0 : csplit; c1 = rpart == 0 ? csplit : mchunks;
    return pair < npairs && c0 < c1;
}

template <int NW, int KPW, int TAG, bool BF16>
__global__ __launch_bounds__(NW * 64) void gt_lstm_x_mc_kernel(LstmXArgs A) {
    static_assert(NW * 64 == 512, "one (row, col) element of each of the two tiles per thread");
    __shared__ __attribute__((aligned(16))) float lds[LeanLds<NW, 2>::kFloats];
    int tile0, ntile, c0, c1;
    if (!gt_lstm_mc_job(A, tile0, ntile, c0, c1)) return;
    GT_STAMP(A.dbg, 4);
    LeanW<KPW, 2, BF16> W;
    gt_lstm_x_mc_body<NW, KPW, BF16, false, false>(A, tile0, ntile, c0, c1, lds, W);
}

// Both cells in one launch at batches above 32 rows: gt_lstm12_kernel's protocol on the multi-chunk bodies.  Every workgroup
// runs its (pair, chunk half) of layer 1, arrives, requests its pair's layer-2 weights, waits for all arrivals and runs the
// same (pair, chunk half) of layer 2.  (Workgroups without a job -- a padded pair index -- arrive at once and leave.)
template <int KPW1, int KPW2, bool BF16>
__global__ __launch_bounds__(512) void gt_lstm12_mc_kernel(Lstm12Args P) {
    constexpr int NW = 8;
    __shared__ __attribute__((aligned(16))) float lds[LeanLds<NW, 2>::kFloats];
    __shared__ int s_abort;
    const int lane = threadIdx.x & 63;
    int tile0, ntile, c0, c1;
    const bool job = gt_lstm_mc_job(P.l1, tile0, ntile, c0, c1);
    if (threadIdx.x == 0) s_abort = 0;
    GT_STAMP(P.l1.dbg, 4);
    if (job) {
        LeanW<KPW1, 2, BF16> W1;
        gt_lstm_x_mc_body<NW, KPW1, BF16, true, false>(P.l1, tile0, ntile, c0, c1, lds, W1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(&P.arrive[(blockIdx.x & (GT_L12_NSH - 1)) * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!job) return;
    LeanW<KPW2, 2, BF16> W2;
    gt_lean_mc_load_w<NW, KPW2, 2, BF16, false>(P.l2.wp, tile0, ntile, (P.l2.nkb + 1) >> 1, W2);
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if (threadIdx.x < 64) {
        uint32_t spins = 0;
        for (;;) {
            uint32_t v = lane < GT_L12_NSH ? gt_ldu_sc1(P.arrive + lane * 32) : 0u;
            v = gt_row_sum_u32<GT_L12_NSH>(v);
            if (__builtin_amdgcn_readfirstlane(v) >= P.expect) break;
            ++spins;
            if (spins > (1u << 18)) { if (lane == 0) { atomicOr(P.err, 1u); s_abort = 1; } break; }
            if ((spins & 63u) == 0u && __builtin_amdgcn_readfirstlane(gt_ldu_sc1(P.err)) != 0u) { if (lane == 0) s_abort = 1; break; }
        }
    }
    __syncthreads();
    if (s_abort) return;
    GT_STAMP(P.l2.dbg, 4);
    gt_lstm_x_mc_body<NW, KPW2, BF16, false, true>(P.l2, tile0, ntile, c0, c1, lds, W2);
}

// batches above 32 rows (fp32 and bf16): grid = pairs of tiles (rounded up to 8) x 2 chunk halves, as gt_lstm_x_mc_kernel's
bool gt_lstm12_mc_supported(int nkb1, int nkb2, int H1, int H2, int M, int slots) {
    return nkb1 == 24 && nkb2 == 64 && H1 == H2 && H1 % 4 == 0 && M > 32 && (((H1 + 3) / 4 + 1) / 2 + 7) / 8 * 16 <= slots;
}
int gt_lstm12_mc_grid(int H) { return (((H + 3) / 4 + 1) / 2 + 7) / 8 * 16; }

// Workgroups of a fused launch one compute unit holds at once, from the occupancy API (0: gt_lstm12_kernel, 1 / 2: the multi-chunk
// kernel in fp32 / bf16 -- 179 / 158 registers: ONE 512-thread workgroup per CU, so its 256-workgroup grid needs every CU of a
// 256-CU device and does not fit a partition with fewer).
int gt_lstm12_blocks_per_cu(int which) {
    int n = 0;
    const void* f = which == 0 ? reinterpret_cast<const void*>(gt_lstm12_kernel)
                  : which == 1 ? reinterpret_cast<const void*>(gt_lstm12_mc_kernel<3, 8, false>)
                               : reinterpret_cast<const void*>(gt_lstm12_mc_kernel<2, 4, true>);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, f, 512, 0) != hipSuccess) return 0;
    return n;
}

hipError_t gt_launch_lstm12_mc(const Lstm12Args& a, bool bf16, hipStream_t stream) {
    const dim3 g(gt_lstm12_mc_grid(a.l1.H));
    if (bf16) hipLaunchKernelGGL((gt_lstm12_mc_kernel<2, 4, true>), g, dim3(512), 0, stream, a);
    else hipLaunchKernelGGL((gt_lstm12_mc_kernel<3, 8, false>), g, dim3(512), 0, stream, a);
    return hipGetLastError();
}

// Batches above 32 rows (lean_body.h, "Batches above 32 rows"): a workgroup owns a PAIR of tiles (8 hidden units) and half of
// the batch's 32-row chunks, keeps the pair's weights in registers over its chunks and multiplies every activation fragment
// with both tiles.  The two workgroups of a pair sit 8 block indices apart = on one XCD under the round-robin block -> XCD
// deal (speed only), so the pair's weights leave the Infinity Cache once.  Per chunk the arithmetic is gt_lstm_x_kernel's:
// the states are bitwise the same.
bool gt_lstm_x_supported(int nkb) { return nkb == 24 || nkb == 64; }

template <int TAG>
static void launch_lstm_x(const LstmXArgs& a, int nkb, bool bf16, hipStream_t stream) {
    if (a.M > 32) {         // weights read once per step at any batch
        const dim3 g1((((a.H + 3) / 4 + 1) / 2 + 7) / 8 * 16);      // pairs of tiles (rounded up to 8) x 2 halves of the chunks
        if (nkb == 24) {
            if (bf16) hipLaunchKernelGGL((gt_lstm_x_mc_kernel<8, 2, TAG, true>), g1, dim3(512), 0, stream, a);
            else hipLaunchKernelGGL((gt_lstm_x_mc_kernel<8, 3, TAG, false>), g1, dim3(512), 0, stream, a);
        } else {
            if (bf16) hipLaunchKernelGGL((gt_lstm_x_mc_kernel<8, 4, TAG, true>), g1, dim3(512), 0, stream, a);
            else hipLaunchKernelGGL((gt_lstm_x_mc_kernel<8, 8, TAG, false>), g1, dim3(512), 0, stream, a);
        }
        return;
    }
    const dim3 grid((a.H + 3) / 4, 1);
    if (nkb == 24) {
        // K = 384 on 8 waves x 3 k-blocks (4 x 6 left the reduce + gate epilogue to 256 threads: 2.4 -> 1.7 us in-kernel)
        if (bf16) hipLaunchKernelGGL((gt_lstm_x_kernel<8, 2, TAG, true>), grid, dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((gt_lstm_x_kernel<8, 3, TAG, false>), grid, dim3(512), 0, stream, a);
    } else {
        if (bf16) hipLaunchKernelGGL((gt_lstm_x_kernel<8, 4, TAG, true>), grid, dim3(512), 0, stream, a);
        else hipLaunchKernelGGL((gt_lstm_x_kernel<8, 8, TAG, false>), grid, dim3(512), 0, stream, a);
    }
}

hipError_t gt_launch_lstm_x(const LstmXArgs& a, int nkb, int tag, bool bf16, hipStream_t stream) {
    if (!gt_lstm_x_supported(nkb)) return hipErrorInvalidValue;
    if (tag == TAG_DEC_LSTM1) launch_lstm_x<TAG_DEC_LSTM1>(a, nkb, bf16, stream);
    else launch_lstm_x<TAG_DEC_LSTM2>(a, nkb, bf16, stream);
    return hipGetLastError();
}

// (tile, 16-row M-tile) of workgroup `i` of `n` (= tiles x MT) one-M-tile workgroups.  With two M-tiles both workgroups of a
// tile read the same 64-72 KB of weights: they are placed 8 block indices apart, i.e. on the SAME XCD under the round-robin
// block -> XCD deal (speed only, never correctness), so the second read is an L2 hit instead of a second trip to the
// Infinity Cache (round 1 PMC: 13.25 MB fetched per launch for 5.24 MB of algorithmic bytes).
__device__ __forceinline__ void gt_pair_map(const int i, const int n, const int MT, int& tile, int& mt) {
    if (MT != 2) { tile = i / MT; mt = i % MT; return; }
    const int g = i >> 4, r = i & 15;
    const int half = min(8, (n - g * 16) >> 1);        // tiles in this group of <= 16 workgroups
    tile = g * 8 + r % half;
    mt = r / half;
}

// Projection (K = 1152 = 8 waves x 9 k-blocks) + co-scheduled layer-2 recurrent tiles (K = 1024 = 8 x 8), CT per worker.
template <int CT, bool BF16>
__global__ __launch_bounds__(512) void gt_proj_lean_kernel(ProjArgs P, LeanPartialArgs co, int n_main, int co_begin, int co_end) {
    constexpr int NW = 8;
    __shared__ __attribute__((aligned(16))) float lds[LeanLds<NW, CT>::kFloats];
    const int mchunks = (P.M + 31) / 32;
    if ((int)blockIdx.x >= n_main) {
        const bool st = P.dbg && (int)blockIdx.x == n_main && threadIdx.x == 0;
        if (st) P.dbg[4] = __builtin_amdgcn_s_memrealtime();
        if (CT == 1) {          // one workgroup per (tile, 16-row M-tile): lighter jobs that end with the projection's own
            int ct, cm;
            gt_pair_map((int)blockIdx.x - n_main, (int)gridDim.x - n_main, P.MT, ct, cm);
            gt_lean_partial<NW, BF16 ? 4 : 8, 1, BF16, true>(co, co_begin + ct, 1, cm, lds);
        } else {
            const int tile = co_begin + ((int)blockIdx.x - n_main) * CT;
            for (int mc = 0; mc < mchunks; ++mc) {
                gt_lean_partial<NW, BF16 ? 4 : 8, CT, BF16>(co, tile, min(CT, co_end - tile), mc, lds);
                if (mc + 1 < mchunks) __syncthreads();
            }
        }
        if (st) P.dbg[5] = __builtin_amdgcn_s_memrealtime();
        return;
    }
    GT_STAMP(P.dbg, 0);
    // main tiles: one workgroup per (tile, 16-row M-tile) -- 27 tiles would leave most CUs idle, so the rows are split; or
    // (P.both_m) one workgroup per tile and both M-tiles: the same sums per output, the tile's weights requested once
    int tile, mt;
    if (P.both_m) { tile = (int)blockIdx.x; mt = 0; }
    else gt_pair_map((int)blockIdx.x, n_main, P.MT, tile, mt);
    const int col = threadIdx.x & 15;
    const int row = P.both_m ? (threadIdx.x >> 4) : ((threadIdx.x >> 4) & 15), half = P.both_m ? 0 : (threadIdx.x >> 8);   // one M-tile: waves 0-3 reduce
    const int gcol = tile * 16 + col;
    const float bias = P.bias[gcol];
    f32x4 acc0[1] = {f32x4{0.f, 0.f, 0.f, 0.f}}, acc1[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    if (P.both_m) {
        if (BF16) gt_lean_core_bf16<NW, 5, 1, false, false>(P.wp, tile, 1, LeanX{P.xa, P.xb, P.nkb_a}, P.MT, 0, 36, acc0, acc1);
        else gt_lean_core<NW, 9, 1, false, false>(P.wp, tile, 1, LeanX{P.xa, P.xb, P.nkb_a}, P.MT, 0, acc0, acc1);
    } else {
        if (BF16) gt_lean_core_bf16<NW, 5, 1, false, true>(P.wp, tile, 1, LeanX{P.xa, P.xb, P.nkb_a}, P.MT, mt, 36, acc0, acc1);
        else gt_lean_core<NW, 9, 1, false, true>(P.wp, tile, 1, LeanX{P.xa, P.xb, P.nkb_a}, P.MT, mt, acc0, acc1);
    }
    GT_STAMP(P.dbg, 1);
    gt_lean_spill<NW, 1>(lds, acc0, acc1);          // (one M-tile: rows 16..31 of the slab are unused zeros)
    __syncthreads();
    GT_STAMP(P.dbg, 2);
    const float (*part)[32][17] = reinterpret_cast<const float (*)[32][17]>(lds);
    if (half == 0) {
        float v = bias;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += part[w][row][col];
        const int grow = mt * 16 + row;
        if (grow < P.M && gcol < P.N) {
            if (P.out3 && gcol >= P.col3) P.out3[(size_t)grow * P.ldo3 + (gcol - P.col3)] = v;
            else if (gcol < P.n_split) P.out[(size_t)grow * P.ldo + gcol] = v;
            else if (!P.out3 || gcol < P.n_valid2) P.out2[(size_t)grow * P.ldo2 + (gcol - P.n_split)] = v;
        }
    }
    GT_STAMP(P.dbg, 3);
}

// Batches above 32 rows.  Main workgroups own (tile, 32-row chunk); the chunks of a tile sit 8 block indices apart, i.e. on
// one XCD under the round-robin block -> XCD deal, so a tile's 72 KB of weights leave the Infinity Cache once.  Workers: (pair
// of layer-2 recurrent tiles, chunk) units.
template <bool BF16>
__global__ __launch_bounds__(512) void gt_proj_mc_kernel(ProjArgs P, LeanPartialArgs co, int n_main, int co_begin, int co_end) {
    constexpr int NW = 8;
    __shared__ __attribute__((aligned(16))) float lds[LeanLds<NW, 2>::kFloats];
    const int mchunks = (P.M + 31) / 32;
    if ((int)blockIdx.x >= n_main) {
        // one workgroup per (pair, chunk): the launch has ~150 CUs to spare for a few microseconds, not 30 of them for the
        // 25 us a pair takes over four chunks; a pair's chunks sit 8 block indices apart (one XCD: the weights' second read is an L2 hit)
        const int wi = (int)blockIdx.x - n_main, npairs = (co_end - co_begin + 1) / 2;
        const int g = wi / (8 * mchunks), r = wi % (8 * mchunks);
        const int half = min(8, npairs - g * 8);
        const int tile = co_begin + (g * 8 + r % half) * 2, mc = r / half;
        gt_lean_partial_mc<NW, BF16 ? 4 : 8, 2, BF16>(co, tile, min(2, co_end - tile), mc, mc + 1, lds);
        return;
    }
    const int ntiles = n_main / mchunks;
    const int g = (int)blockIdx.x / (8 * mchunks), r = (int)blockIdx.x % (8 * mchunks);
    const int half = min(8, ntiles - g * 8);
    const int tile = g * 8 + r % half, mc = r / half;
    const int row = threadIdx.x >> 4, col = threadIdx.x & 15;
    const int gcol = tile * 16 + col;
    const float bias = P.bias[gcol];
    f32x4 acc0[1] = {f32x4{0.f, 0.f, 0.f, 0.f}}, acc1[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    if (BF16 && P.xah && P.xbh) gt_lean_core_bf16<NW, 5, 1, false, false, BF16>(P.wp, tile, 1, LeanX{P.xa, P.xb, P.nkb_a, P.xah, P.xbh}, P.MT, mc, 36, acc0, acc1);
    else if (BF16) gt_lean_core_bf16<NW, 5, 1, false>(P.wp, tile, 1, LeanX{P.xa, P.xb, P.nkb_a}, P.MT, mc, 36, acc0, acc1);
    else gt_lean_core<NW, 9, 1, false>(P.wp, tile, 1, LeanX{P.xa, P.xb, P.nkb_a}, P.MT, mc, acc0, acc1);
    gt_lean_spill<NW, 1>(lds, acc0, acc1);
    __syncthreads();
    const float (*part)[32][17] = reinterpret_cast<const float (*)[32][17]>(lds);
    float v = bias;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += part[w][row][col];
    const int grow = mc * 32 + row;
    if (grow < P.M && gcol < P.N) {
        if (P.out3 && gcol >= P.col3) P.out3[(size_t)grow * P.ldo3 + (gcol - P.col3)] = v;
        else if (gcol < P.n_split) P.out[(size_t)grow * P.ldo + gcol] = v;
        else if (!P.out3 || gcol < P.n_valid2) P.out2[(size_t)grow * P.ldo2 + (gcol - P.n_split)] = v;
    }
}

bool gt_proj_lean_supported(int nkb_main, int nkb_co) { return nkb_main == 72 && (nkb_co == 64 || nkb_co == 0); }

hipError_t gt_launch_proj_lean(const ProjArgs& m, int ntiles, const float* co_wp, const float* co_bias, const float* co_x,
                               float* co_out, int co_begin, int co_end, int tiles_per_worker, bool bf16, hipStream_t stream) {
    const int nco = co_end > co_begin ? co_end - co_begin : 0;
    LeanPartialArgs co{co_wp, co_bias, co_x, co_out, m.MT};
    if (co_x == m.xa) co.xh = m.xah;          // the workers multiply the same h2 the projection reads: its bf16 mirror, if there is one
    if (m.M > 32) {
        const int n_mc = ntiles * ((m.M + 31) / 32);
        const dim3 g(n_mc + (nco + 1) / 2 * ((m.M + 31) / 32));
        if (bf16) hipLaunchKernelGGL((gt_proj_mc_kernel<true>), g, dim3(512), 0, stream, m, co, n_mc, co_begin, co_end);
        else hipLaunchKernelGGL((gt_proj_mc_kernel<false>), g, dim3(512), 0, stream, m, co, n_mc, co_begin, co_end);
        return hipGetLastError();
    }
    const int n_main = m.both_m ? ntiles : ntiles * m.MT;
    const dim3 g2(n_main + (nco + 1) / 2), g1(n_main + nco * m.MT);
    if (tiles_per_worker == 2) {
        if (bf16) hipLaunchKernelGGL((gt_proj_lean_kernel<2, true>), g2, dim3(512), 0, stream, m, co, n_main, co_begin, co_end);
        else hipLaunchKernelGGL((gt_proj_lean_kernel<2, false>), g2, dim3(512), 0, stream, m, co, n_main, co_begin, co_end);
    } else {
        if (bf16) hipLaunchKernelGGL((gt_proj_lean_kernel<1, true>), g1, dim3(512), 0, stream, m, co, n_main, co_begin, co_end);
        else hipLaunchKernelGGL((gt_proj_lean_kernel<1, false>), g1, dim3(512), 0, stream, m, co, n_main, co_begin, co_end);
    }
    return hipGetLastError();
}

// Bidirectional LSTM time step on the hoisted input halves (see BiLstmArgs): K = H = 256 -> 8 waves x 2 k-blocks.
template <int NW, int KPW>
__global__ __launch_bounds__(NW * 64) void gt_bilstm_lean_kernel(BiLstmArgs A) {
    __shared__ __attribute__((aligned(16))) float lds[LeanLds<NW, 1>::kFloats];
    constexpr int NE = 512 / (NW * 64);
    const BiLstmDir& D = A.d[blockIdx.z];
    const int tile = blockIdx.x, mchunk = blockIdx.y;
    const int m0 = mchunk * 32, MT = A.MT;
    float pin[NE], c_prev[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = threadIdx.x + i * NW * 64;
        const int row = e >> 4, col = e & 15;
        const int grow = m0 + row, unit = tile * 4 + col;
        pin[i] = (grow < A.M) ? D.zx[(size_t)grow * A.ldz + tile * 16 + col] : 0.f;
        c_prev[i] = (col < 4 && grow < A.M && unit < A.H) ? D.c[(size_t)grow * A.H + unit] : 0.f;
    }
    f32x4 acc0[1] = {f32x4{0.f, 0.f, 0.f, 0.f}}, acc1[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
    gt_lean_core<NW, KPW, 1, false>(D.wp, tile, 1, LeanX{D.hprev, D.hprev, NW * KPW}, MT, mchunk, acc0, acc1);
    gt_lean_spill<NW, 1>(lds, acc0, acc1);
    __syncthreads();
    const float (*part)[32][17] = reinterpret_cast<const float (*)[32][17]>(lds);
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = threadIdx.x + i * NW * 64;
        const int row = e >> 4, col = e & 15;
        float z = pin[i];
#pragma unroll
        for (int w = 0; w < NW; ++w) z += part[w][row][col];
        const float zf = gt_row_down<4>(z), zg = gt_row_down<8>(z), zo = gt_row_down<12>(z);
        const int grow = m0 + row, unit = tile * 4 + col;
        if (col < 4 && grow < A.M && unit < A.H) {
            float hv = 0.f;
            if (!(A.row_len && D.t_index >= A.row_len[grow])) {
                const float gi = gt_sigmoid(z), gf = gt_sigmoid(zf), gg = gt_tanh(zg), go = gt_sigmoid(zo);
                const float c2 = __builtin_fmaf(gf, c_prev[i], gi * gg);
                D.c[(size_t)grow * A.H + unit] = c2;
                hv = go * gt_tanh(c2);
            }
            D.hnext[gt_blk_off(grow, unit, MT)] = hv;
            D.out[(size_t)grow * A.ldo + unit] = hv;
        }
    }
}


// ======================================================================================================================
// Persistent Bidirectional LSTM: ALL time steps of both directions in ONE launch (reference Taco2.py:39-43, 394-398).
//
// The per-step kernel above is a chain of 128 launches of ~2 us of kernel boundary + ~3 us of kernel each, whose work (16
// MFMAs per wave) is a small part of it, and which re-reads its weight tile every step.  Here the workgroups stay resident for
// the whole sequence with their weight tiles and cell state in REGISTERS and hand the hidden state around themselves.
//
// What makes that cheap is WHERE the workgroups sit (tools/handoff3.hip, profiles/r02_handoff_one_xcd.txt): an all-to-all
// among workgroups spread over the chip needs write-through stores and fabric-served loads and costs as much as the kernel
// boundary it replaces (tools/handoff.hip; measured on this very kernel: 5.6 us / step against 5.3 us for the launches); the
// same exchange among the 32 workgroups of ONE XCD goes through that XCD's L2, which is coherent for them -- plain
// stores, loads that only have to miss the reader's L1 (sc1) -- and costs 1.15 us, flags and a 32 KiB read included.
// The recurrences of different utterances are independent, so the job is cut into GROUPS = (direction, 16-utterance M-tile),
// each wholly on one XCD (2 directions x up to 4 M-tiles = 64 utterances on the 8 XCDs), 32 members per group, each member
// owning 2 gate tiles (8 hidden units x 4 gates) for its group's 16 rows.  Groups never talk to each other.  The launch is
// 512 workgroups, which the dispatcher deals round-robin to the XCDs; a workgroup reads the XCD it landed on from the
// hardware (XCC_ID) -- that is its group -- and takes the next free member slot of that group from a counter; groups that
// have nothing to do and arrivals beyond the 32nd exit at once.  (Nothing is assumed about blockIdx -> XCD except that
// every XCD receives at least 32 of the 512.)
//
// Hand-off per step inside a group (round 6: the state carries its own validity, no flags): |h| < 1, so bit 30 of an h word -- the top
// exponent bit, set only from 2.0 up -- is free.  The owner lanes store h_t with a TAG in that bit into slot t % 3 of the group's blocked
// state (plain 4-byte stores: the XCD's L2 is coherent for its own CUs), and that is all a producer does: no drain, no barrier, no flag.
// A consumer wave polls its two k-blocks of the state ITSELF (two 16-byte sc1 loads per lane) until all eight words of every lane show
// the tag of the step it waits for, and strips it: ONE L2 round trip per step where the flag form took two (poll the 32 flags, then
// fetch the state) plus the producers' store acknowledgement and a barrier -- 1.9 -> see DESIGN 3.5 us per time step.  Tags: slot s is written
// at steps s, s + 3, s + 6, ... with tag ((t / 3) & 1) ^ 1 (the launch starts from a zeroed state: tag 0 = "never written"), so what a
// slot held before differs in the tag.  Three slots, not two: a member overwrites slot t % 3 at step t + 3, for which it needs step
// t + 2 of its four source members, who needed step t + 1 of EVERY member, who had therefore all finished reading step t (with two
// slots nothing orders a producer behind a reader it does not depend on).  The LDS partial sums alternate between two buffers by step
// parity, so the one barrier per step (spill -> reduce) also orders their re-use.  Waits are bounded (error word instead of a hung GPU) --
// which is also what a NaN / Inf state ends in (their exponent has bit 30 set: the tag can no longer be told; a model that produces them
// is reported as a give-up instead of silently propagating them).
//
// Arithmetic = gt_bilstm_lean_kernel<8, 2>'s exactly (k-block kb on wave kb % 8, ascending; partial sums added over waves in
// ascending order after the hoisted input half): bitwise equal outputs, which is what the GPU test checks.
__device__ __forceinline__ void gt_ld2x4_sc1(const float* p0, const float* p1, float4& a, float4& b) {
    f32x4 ra, rb;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\t"
                 "global_load_dwordx4 %1, %3, off sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(ra), "=&v"(rb)
                 : "v"(p0), "v"(p1)
                 : "memory");
    a = make_float4(ra[0], ra[1], ra[2], ra[3]); b = make_float4(rb[0], rb[1], rb[2], rb[3]);
}

constexpr int kPersistLds = 96 * 1024;
// BF16 (Use_Mixed_Precision): W_h as the bf16 pack [tile][32-k block][lane][8], the state rounded to bf16 on its way into
// v_mfma_f32_16x16x32_bf16 (wave w owns 32-k block w = the 16-blocks 2w, 2w + 1) -- the operand roundings of the general bf16
// step kernel; accumulation, gates and state stay fp32.
template <bool BF16>
__global__ __launch_bounds__(512) void gt_bilstm_persist_kernel(BiLstmPersistArgs A) {
    constexpr int NW = 8, KPW = 2, NT = 2, NM = 32;                // waves, k-blocks per wave, tiles per member, members per group
    // (launched with kPersistLds bytes of dynamic LDS, more than half a CU's: one workgroup per CU, so that the 32 members of
    // a group sit on the 32 CUs of their XCD instead of sharing matrix cores in pairs)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int s_slot;
    __shared__ int s_abort;
    float (*part)[NT][16][17] = reinterpret_cast<float (*)[NT][16][17]>(lds);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // which XCD am I on?  (the hardware's answer, not blockIdx % 8: under graph replay the round-robin has been seen to start
    // elsewhere.)  The XCD is the group; the first 32 arrivals on it are its members, later ones have nothing to do.
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const int g = (int)(xcc & 7);
    if (g >= 2 * A.MT) return;
    if (tid == 0) { s_slot = (int)atomicAdd(A.flags + g, 1u); s_abort = 0; }
    __syncthreads();
    const int rank = s_slot;
    if (rank >= NM) return;
    if (rank == A.debug_drop_member) return;                       // fault injection (tests): this member never shows up
    const int d = g & 1, mt = g >> 1, H = A.H;
    // this thread's element of the member's 16 rows x (2 tiles x 16 gate columns)
    const int row = tid >> 5, j_own = (tid >> 4) & 1, col = tid & 15;
    const int grow = mt * 16 + row, tile0 = rank * NT, unit = (tile0 + j_own) * 4 + col;
    const bool owner = col < 4;                                    // lanes that own a hidden unit (rows beyond M publish zeros)
    const bool real = grow < A.M;
    // weights: k-blocks wave and wave + 8 of both tiles (bf16: 32-k block `wave`), resident for the whole sequence
    float4 b[KPW][NT];
    uint4 b32[NT];
    if constexpr (BF16) {
#pragma unroll
        for (int j = 0; j < NT; ++j) b32[j] = (reinterpret_cast<const uint4*>(A.wp[d]) + ((size_t)(tile0 + j) * NW + wave) * 64)[lane];
    } else {
#pragma unroll
        for (int i = 0; i < KPW; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                b[i][j] = (reinterpret_cast<const float4*>(A.wp[d]) + ((size_t)(tile0 + j) * (NW * KPW) + wave + i * NW) * 64)[lane];
    }
    float* hbuf = A.h + (size_t)g * 3 * 16 * H;                     // [3 slots][H/16 k-blocks][64 lanes][4]
    // (rows beyond M read the group's first row, which exists, and are never used: no load under a branch, no wait at its join)
    const float* zrow = A.zx + (size_t)(real ? grow : mt * 16) * A.ldz + (size_t)d * 4 * H + (tile0 + j_own) * 16 + col;
    float c_state = 0.f;
    float pin = zrow[(size_t)(d == 0 ? 0 : A.T - 1) * 8 * H];
    const int n_valid = (A.row_len && real) ? A.row_len[grow] : A.T;       // masked mode: steps tt >= n_valid do not exist for this row
    int slot_w = 0, slot_r = 2;                                    // slot this step writes (t % 3) / reads ((t - 1) % 3)
    uint32_t tag_w = 1u << 30, tag_r = 0u;                         // ... and their tags
    for (int t = 0; t < A.T; ++t) {
        const int tt = d == 0 ? t : A.T - 1 - t;
        float4 x[KPW];
        if (t == 0) {
#pragma unroll
            for (int i = 0; i < KPW; ++i) x[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            // Bounded wait.  A member that gives up raises the (host-mapped) error word; every waiting wave of every group looks
            // at that word now and then, so the WHOLE launch drains within microseconds of the first give-up instead of every
            // remaining time step spinning its full bound again.  The workgroup leaves together behind its next barrier.
            const float* hp = hbuf + (size_t)slot_r * 16 * H + lane * 4;
            const float* p0 = BF16 ? hp + (size_t)(2 * wave) * 256 : hp + (size_t)wave * 256;
            const float* p1 = BF16 ? hp + (size_t)(2 * wave + 1) * 256 : hp + (size_t)(wave + NW) * 256;
            uint32_t spins = 0;
            for (;;) {
                gt_ld2x4_sc1(p0, p1, x[0], x[1]);
                const uint32_t all = __builtin_bit_cast(uint32_t, x[0].x) & __builtin_bit_cast(uint32_t, x[0].y) & __builtin_bit_cast(uint32_t, x[0].z) &
                                     __builtin_bit_cast(uint32_t, x[0].w) & __builtin_bit_cast(uint32_t, x[1].x) & __builtin_bit_cast(uint32_t, x[1].y) &
                                     __builtin_bit_cast(uint32_t, x[1].z) & __builtin_bit_cast(uint32_t, x[1].w);
                const uint32_t any = __builtin_bit_cast(uint32_t, x[0].x) | __builtin_bit_cast(uint32_t, x[0].y) | __builtin_bit_cast(uint32_t, x[0].z) |
                                     __builtin_bit_cast(uint32_t, x[0].w) | __builtin_bit_cast(uint32_t, x[1].x) | __builtin_bit_cast(uint32_t, x[1].y) |
                                     __builtin_bit_cast(uint32_t, x[1].z) | __builtin_bit_cast(uint32_t, x[1].w);
                // every word carries the wanted tag: all of them set (tag 1) / none of them set (tag 0)
                const bool ok = tag_r ? (all & (1u << 30)) != 0u : (any & (1u << 30)) == 0u;
                if (__builtin_amdgcn_readfirstlane(__popcll(__ballot(ok))) == 64) break;
                ++spins;
                if (spins > (1u << 18)) { if (lane == 0) { atomicOr(A.err, 1u); s_abort = 1; } break; }
                if ((spins & 63u) == 0u && __builtin_amdgcn_readfirstlane(gt_ldu_sc1(A.err)) != 0u) { if (lane == 0) s_abort = 1; break; }
            }
            const uint32_t keep = ~(1u << 30);
#pragma unroll
            for (int i = 0; i < KPW; ++i) {
                x[i].x = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x[i].x) & keep);
                x[i].y = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x[i].y) & keep);
                x[i].z = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x[i].z) & keep);
                x[i].w = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x[i].w) & keep);
            }
        }
        // the next step's hoisted input half: requested now, needed after the next wait
        const float cur = pin;
        {
            const int tn = min(t + 1, A.T - 1);
            pin = zrow[(size_t)(d == 0 ? tn : A.T - 1 - tn) * 8 * H];
        }
        f32x4 acc[NT] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if constexpr (BF16) {
            bf16x8 a;
            a[0] = (__bf16)x[0].x; a[1] = (__bf16)x[0].y; a[2] = (__bf16)x[0].z; a[3] = (__bf16)x[0].w;
            a[4] = (__bf16)x[1].x; a[5] = (__bf16)x[1].y; a[6] = (__bf16)x[1].z; a[7] = (__bf16)x[1].w;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bf16x8 bw;
                __builtin_memcpy(&bw, &b32[j], 16);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bw, acc[j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < KPW; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i].x, b[i][j].x, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i].y, b[i][j].y, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i].z, b[i][j].z, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x[i].w, b[i][j].w, acc[j], 0, 0, 0);
                }
        }
        float (*pt)[NT][16][17] = part + (t & 1) * NW;              // (two buffers by step parity: the one barrier below orders their re-use)
        {   // C/D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg
            const int r = lane & 15, q = lane >> 4;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int v = 0; v < 4; ++v) pt[wave][j][q * 4 + v][r] = acc[j][v];
        }
        __syncthreads();
        if (s_abort) return;                                        // (uniform: written before the barrier)
        float z = cur;
#pragma unroll
        for (int w = 0; w < NW; ++w) z += pt[w][j_own][row][col];
        const float zf = gt_row_down<4>(z), zg = gt_row_down<8>(z), zo = gt_row_down<12>(z);
        float hv = 0.f;
        if (owner && real && tt < n_valid) {
            const float gi = gt_sigmoid(z), gf = gt_sigmoid(zf), gg = gt_tanh(zg), go = gt_sigmoid(zo);
            c_state = __builtin_fmaf(gf, c_state, gi * gg);
            hv = go * gt_tanh(c_state);
        }
        // the state first (somebody waits for it), tagged; then the encoding
        if (owner && t + 1 < A.T)
            reinterpret_cast<uint32_t*>(hbuf)[(size_t)slot_w * 16 * H + gt_blk_off(row, unit, 1)] = __builtin_bit_cast(uint32_t, hv) | tag_w;
        if (owner && real) A.out[(size_t)grow * A.ldo + (size_t)tt * 2 * H + (size_t)d * H + unit] = hv;
        slot_r = slot_w; tag_r = tag_w;
        if (slot_w == 2) { slot_w = 0; tag_w ^= 1u << 30; } else ++slot_w;
    }
}

// H = 256 (16 k-blocks on 8 waves, 64 tiles on 32 members); 2 directions x ceil(B/16) M-tiles must fit the 8 XCDs, each of
// which must be able to hold its 32 members at once.
bool gt_bilstm_persist_supported(int H, int B, int n_cu) { return H == 256 && B <= 64 && n_cu >= 256; }

hipError_t gt_launch_bilstm_persist(const BiLstmPersistArgs& a, bool bf16, hipStream_t stream) {
    if (bf16) hipLaunchKernelGGL(gt_bilstm_persist_kernel<true>, dim3(512), dim3(512), kPersistLds, stream, a);
    else hipLaunchKernelGGL(gt_bilstm_persist_kernel<false>, dim3(512), dim3(512), kPersistLds, stream, a);
    return hipGetLastError();
}

hipError_t gt_bilstm_persist_init() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gt_bilstm_persist_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kPersistLds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gt_bilstm_persist_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kPersistLds);
}

// workgroups of the persistent kernel one compute unit can hold at once (the design needs exactly 1: the 32 members of a
// group each on their own CU of the group's XCD); 0 = it does not fit at all
int gt_bilstm_persist_blocks_per_cu() {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(gt_bilstm_persist_kernel<false>), 512, kPersistLds) != hipSuccess) return 0;
    return n;
}

bool gt_bilstm_lean_supported(int nkb_h) { return nkb_h == 16; }

hipError_t gt_launch_bilstm_lean(const BiLstmArgs& a, hipStream_t stream) {
    const dim3 grid((a.H + 3) / 4, (a.M + 31) / 32, 2);
    hipLaunchKernelGGL((gt_bilstm_lean_kernel<8, 2>), grid, dim3(512), 0, stream, a);
    return hipGetLastError();
}
