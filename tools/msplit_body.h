// Batches above 32 rows, M-split form of the decode-step GEMM bodies (gst_tacotron_amd/csrc/lean_body.h has the K-split forms the product
// runs, and the history).  A TOOL (tools/msplit_bench.hip): bitwise the K-split bodies and not faster per launch (87 TFLOP/s either way);
// repeated inside one launch it holds 132 TFLOP/s -- what a ~20 us launch loses is its start-up, not its body (EXPERIMENTS round 4,
// profiles/r04_msplit.txt).
//
// The K-split multi-chunk bodies (gt_lean_mc) split a tile's K over the workgroup's waves and walk the batch in 32-row chunks:
// per chunk every wave multiplies, spills its partial sums to LDS, waits at a barrier, and the workgroup reduces -- all waves
// are in the same phase at the same time, so the matrix pipe idles while they reduce (fp32, 128 rows: 5.8 us per (pair of tiles,
// chunk) for 4.1 us of MFMA at the clock the chip sustains).  Here a WAVE owns a 16-row M-tile and the whole K:
//   - the job's weights (a pair of tiles: 2 x NKB KB) are staged ONCE in LDS, straight from memory (buffer_load ... lds: no
//     registers), and every wave reads them from there (ds_read_b128, lane-contiguous: conflict-free);
//   - a wave streams its M-tile's activation fragments through a ring of D registers, D k-blocks ahead;
//   - no partial sums cross waves, so after the staging barrier the waves never meet again: each runs its MFMAs back to back and
//     a wave that waits for a load leaves the pipe to the other wave of its SIMD.
// Bitwise the K-split (and therefore the general) kernels: the K-split wave w of ORDER waves adds k-blocks w, w + ORDER, ... in
// ascending order into one accumulator, and the epilogue adds the ORDER partial sums in ascending wave order; here a wave keeps
// ORDER accumulators per tile, k-block kb goes into accumulator kb % ORDER (same MFMA sequence per accumulator), and the
// epilogue adds them in the same order.
#pragma once
#include "../gst_tacotron_amd/csrc/lean_body.h"

// 16 bytes per lane from a buffer straight into LDS: lane l's data lands at lds_addr + 16 l (lds_addr wave-uniform).  Inline asm,
// not __builtin_amdgcn_raw_ptr_buffer_load_lds: the compiler orders every later LDS read of the kernel behind a DMA it knows of
// (a vmcnt(0) in front of each).  The caller waits (s_waitcnt vmcnt(0)) and synchronises before the data is read.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void gt_lds_dma16(__amdgpu_buffer_rsrc_t rs, const uint32_t lds_addr, const uint32_t voff, const uint32_t soff) {
    // (M0 is a scratch register for the compiler: it sets it right before each of its own uses)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
#pragma clang diagnostic pop

template <int NKB>
struct MSplitLds {
    static constexpr int kFloats = 2 * NKB * 256;       // [k-block][tile of the pair][64 lanes][4]
};

// Stage the pair's weights: DMA instruction n = 2 kb + j moves k-block kb of tile tile0 + j (1 KB) to lds + n KB; the NWAVES waves
// take n = wave, wave + NWAVES, ...  (a pair's second tile beyond the matrix re-reads the first: multiplied, never stored).
template <int NWAVES, int NKB>
__device__ __forceinline__ void gt_msplit_stage(const float* __restrict__ wp, const int tile0, const int ntile, float* lds) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const auto rs = gt_rsrc(wp, 0x7FFFF000u);
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
#pragma unroll
    for (int i = 0; i < (2 * NKB + NWAVES - 1) / NWAVES; ++i) {
        const int n = wave + i * NWAVES;
        if ((2 * NKB) % NWAVES == 0 || n < 2 * NKB) {
            const int kb = n >> 1, j = (n & 1) < ntile ? (n & 1) : 0;
            gt_lds_dma16(rs, base + (uint32_t)n * 1024u, (uint32_t)lane * 16u, (uint32_t)(((tile0 + j) * NKB + kb) * 1024));
        }
    }
}

// One wave's share of a staged job: M-tile `mt`, tiles j0 .. j0 + TPW - 1 of the pair.  acc[t][w]: tile j0 + t, accumulator of the
// K-split kernels' wave w.  D = activation fragments in flight.
template <int ORDER, int NKB, int TPW, int D = 8>
__device__ __forceinline__ void gt_msplit_wave(const float* lds, const int j0, const LeanX X, const int MT, const int mt, f32x4 (&acc)[TPW][ORDER]) {
    const int lane = threadIdx.x & 63;
    const LeanXR XR = gt_x_rsrc(X);
    const float4* wl = reinterpret_cast<const float4*>(lds) + j0 * 64 + lane;
    float4 x[D];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int w = 0; w < ORDER; ++w) acc[t][w] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < D; ++i) x[i] = gt_xload(XR, X, i < NKB ? i : 0, MT, mt);
    // k-block kb: the weight fragments of kb + 1 are requested from LDS (b[(kb + 1) & 1]), then the 4 TPW MFMAs of kb, then the
    // activation fragment of kb + D into the ring slot just consumed.  One scheduling region per k-block: over the whole unrolled K
    // the scheduler hoists the LDS reads of all 64 k-blocks to the top and spills them.
    float4 b[2][TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) b[0][t] = wl[t * 64];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        if (kb + 1 < NKB) {
#pragma unroll
            for (int t = 0; t < TPW; ++t) b[(kb + 1) & 1][t] = wl[((kb + 1) * 2 + t) * 64];
        }
        const float4 xv = x[kb % D];
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t][kb % ORDER] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.x, b[kb & 1][t].x, acc[t][kb % ORDER], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t][kb % ORDER] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.y, b[kb & 1][t].y, acc[t][kb % ORDER], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t][kb % ORDER] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.z, b[kb & 1][t].z, acc[t][kb % ORDER], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t][kb % ORDER] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.w, b[kb & 1][t].w, acc[t][kb % ORDER], 0, 0, 0);
#ifndef GT_MSPLIT_NO_X          // (tools/msplit_bench.hip ablation: the first D fragments reused for the whole K)
        if (kb + D < NKB) x[kb % D] = gt_xload(XR, X, kb + D, MT, mt);
#endif
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Recurrent-half worker job in the M-split form: gt_lean_partial_mc's result for M-tiles [mt0, mt1) of the pair (tile0, tile0 + 1).
// NWAVES waves; TPW = 2: wave w owns M-tile mt0 + w (+ NWAVES, ...) and both tiles; TPW = 1: wave w owns M-tile mt0 + (w >> 1) (+ NWAVES / 2,
// ...) and tile w & 1 (for the 16-wave front launch, whose 128 registers per lane hold one tile's 16 accumulators).
// All waves must call; the caller synchronises before the LDS is reused.
template <int NWAVES, int ORDER, int NKB, int TPW>
__device__ __forceinline__ void gt_msplit_partial(const LeanPartialArgs& A, const int tile0, const int ntile, const int mt0, const int mt1, float* lds) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    gt_msplit_stage<NWAVES, NKB>(A.wp, tile0, ntile, lds);
    const int j0 = TPW == 2 ? 0 : (wave & 1);
    const int mstep = TPW == 2 ? NWAVES : NWAVES / 2;
    float bias_v[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) bias_v[t] = A.bias[(tile0 + (j0 + t < ntile ? j0 + t : 0)) * 16 + (lane & 15)];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int mt = mt0 + (TPW == 2 ? wave : (wave >> 1)); mt < mt1; mt += mstep) {
        f32x4 acc[TPW][ORDER];
        gt_msplit_wave<ORDER, NKB, TPW, (TPW == 2 ? 8 : 6)>(lds, j0, LeanX{A.x, A.x, NKB}, A.MT, mt, acc);
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            if (j0 + t < ntile) {
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    float z = bias_v[t];
#pragma unroll
                    for (int w = 0; w < ORDER; ++w) z += acc[t][w][v];
                    const int row = mt * 16 + (lane >> 4) * 4 + v;
                    A.partial_out[((size_t)(tile0 + j0 + t) * A.MT * 16 + row) * 16 + (lane & 15)] = z;
                }
            }
        }
    }
}
