// Lean decode-step GEMM bodies: the same arithmetic as gt_skinny_body (skinny_body.h) for the shapes the decode loop
// actually runs, with everything that made the general body slow to START resolved at compile time.
//
// Measured on MI355X (tools/stamps.py): the general body, which handles three segments of either layout, both
// precisions and a runtime K, spends 1.7 us of the layer-2 LSTM launch just ISSUING its 24 loads per wave (~40 scalar
// instructions and several branches per k-block, serialised s_load waits on a 560-byte argument block) before the
// first byte is requested.  Here the k-blocks of a wave are a compile-time count (KPW), the operands are blocked
// (MFMA-fragment order) segments, the arguments fit one s_load, and a wave's whole K range -- KPW x (NT weight + 2
// activation) 16-byte loads -- is requested by straight-line code in the first ~100 instructions.
//
// Arithmetic is identical to the general body: wave w owns k-blocks w, w+NW, ... in ascending order on
// v_mfma_f32_16x16x4_f32, partial sums are added over waves in ascending order after the bias / recurrent half, so
// results are bitwise those of gt_skinny_body (tests/test_gpu_parity.py compares the two paths).
#pragma once
#include "device_utils.h"
#include "kernels.h"

template <int NW, int NT>
struct LeanLds {
    static constexpr int kFloats = NT * NW * 32 * 17;
};

// Blocked activation operand of up to two segments: k-blocks [0, nkb_a) from xa, the rest from xb.
struct LeanX {
    const float* xa;
    const float* xb;
    int nkb_a;
    const uint16_t* ha = nullptr;       // bf16 mirrors of the two segments (kernels.h gt_blk_off_h) or NULL
    const uint16_t* hb = nullptr;
};

// Activation fragment of k-block `kb` (wave-uniform), M-tile `mt`: one 16-byte load per lane.  GT_X_SC1 (default): as sc1
// buffer loads -- the activations were written by the previous launch's workgroups all over the chip, and 32 CUs of an XCD
// missing the same fresh lines serialise in that XCD's L2; bypassing it is 0.6 us per launch at 128 KB (device_utils.h
// gt_bload4_sc1, tools/persist_phase.hip).  Build with -DGT_X_SC1=0 for plain loads (same values either way).
#ifndef GT_X_SC1
#define GT_X_SC1 1
#endif
struct LeanXR {
#if GT_X_SC1
    __amdgpu_buffer_rsrc_t a, b;
#endif
};
__device__ __forceinline__ LeanXR gt_x_rsrc(const LeanX& X) {
#if GT_X_SC1
    return LeanXR{gt_rsrc(X.xa, 0x7FFFF000u), gt_rsrc(X.xb, 0x7FFFF000u)};
#else
    return LeanXR{};
#endif
}
// The bf16 mirror's fragment of 32-k block `kb32` (wave-uniform; nkb_a even), M-tile `mt`: 8 bf16 per lane, the MFMA's A operand as is.
__device__ __forceinline__ u32x4 gt_xload_h(const LeanX& X, const int kb32, const int MT, const int mt) {
    const int lane = threadIdx.x & 63;
    const bool first = 2 * kb32 < X.nkb_a;
    const uint16_t* base = first ? X.ha : X.hb;
    const uint32_t soff = (uint32_t)(((first ? kb32 : kb32 - (X.nkb_a >> 1)) * MT + mt) * 1024);
#if GT_X_SC1
    const auto t = __builtin_amdgcn_raw_buffer_load_b128(gt_rsrc(base, 0x7FFFF000u), lane * 16, (int)soff, 16);
    u32x4 r;
    __builtin_memcpy(&r, &t, 16);
    return r;
#else
    return *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(base) + soff + lane * 16);
#endif
}
__device__ __forceinline__ float4 gt_xload(const LeanXR& R, const LeanX& X, const int kb, const int MT, const int mt) {
    const int lane = threadIdx.x & 63;
#if GT_X_SC1
    const bool first = kb < X.nkb_a;
    const uint32_t soff = (uint32_t)(((first ? kb : kb - X.nkb_a) * MT + mt) * 1024);
    return first ? gt_bload4_sc1(R.a, (uint32_t)lane * 16u, soff) : gt_bload4_sc1(R.b, (uint32_t)lane * 16u, soff);
#else
    const float* xs = kb < X.nkb_a ? X.xa + (size_t)kb * MT * 256 : X.xb + (size_t)(kb - X.nkb_a) * MT * 256;
    return *reinterpret_cast<const float4*>(xs + mt * 256 + lane * 4);
#endif
}

// acc0[j] / acc1[j]: rows m0..m0+15 / m0+16..m0+31 of tile (tile0 + j).  All NW*64 threads must call.
// ONE_M: `mchunk` counts 16-row M-tiles and only acc0 (rows mchunk*16 .. +15) is computed -- for GEMMs with fewer tiles
// than CUs (the projection), where a second workgroup per tile halves each one's activation pull and MFMA chain.
template <int NW, int KPW, int NT, bool NTW, bool ONE_M = false>
__device__ __forceinline__ void gt_lean_core(const float* __restrict__ wp, const int tile0, const int ntile, const LeanX X, const int MT,
                                             const int mchunk, f32x4 (&acc0)[NT], f32x4 (&acc1)[NT]) {
    constexpr int NKB = NW * KPW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mt0 = ONE_M ? mchunk : mchunk * 2, mt1 = min(mt0 + 1, MT - 1);
    const float4* wl = reinterpret_cast<const float4*>(wp) + ((size_t)tile0 * NKB + wave) * 64 + lane;
    const LeanXR XR = gt_x_rsrc(X);
    float4 b[KPW][NT], x0[KPW], x1[KPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int kb = wave + i * NW;                       // wave-uniform
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            b[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (NT == 1 || j < ntile) {
                const float4* src = wl + ((size_t)j * NKB + i * NW) * 64;
                if (NTW) {
                    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
                    b[i][j] = make_float4(t[0], t[1], t[2], t[3]);
                } else {
                    b[i][j] = *src;
                }
            }
        }
        x0[i] = gt_xload(XR, X, kb, MT, mt0);
        if (!ONE_M) x1[i] = gt_xload(XR, X, kb, MT, mt1);
    }
    // every load above is requested before the first MFMA (without this the scheduler sinks each load next to its use
    // to save registers, and the wave pays one memory latency per k-block instead of one in all)
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            acc0[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].x, b[i][j].x, acc0[j], 0, 0, 0);
            if (!ONE_M) acc1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].x, b[i][j].x, acc1[j], 0, 0, 0);
            acc0[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].y, b[i][j].y, acc0[j], 0, 0, 0);
            if (!ONE_M) acc1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].y, b[i][j].y, acc1[j], 0, 0, 0);
            acc0[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].z, b[i][j].z, acc0[j], 0, 0, 0);
            if (!ONE_M) acc1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].z, b[i][j].z, acc1[j], 0, 0, 0);
            acc0[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].w, b[i][j].w, acc0[j], 0, 0, 0);
            if (!ONE_M) acc1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].w, b[i][j].w, acc1[j], 0, 0, 0);
        }
    }
}

// Mixed precision (Use_Mixed_Precision): the same core on v_mfma_f32_16x16x32_bf16.  Weights are the bf16 pack
// [tile][32-k block][lane][8] (one 16-byte load per lane per 32 k, slot i <-> k = 32 j + 16 (i>>2) + 4 q + (i&3)); the fp32
// activations of the two 16-blocks (2j, 2j+1) are rounded to bf16 (RNE) on their way into the MFMA -- exactly what the
// general body does, in the same order, so the results are bitwise the same.  KPW32 = 32-k blocks per wave (guarded by
// nkb32, so 36 blocks on 8 waves is fine); nkb_a (in 16-blocks) must be even.
// XH: the activations come from the bf16 mirrors (X.ha / X.hb): one 16-byte fragment per 32 k instead of two, no conversion.
template <int NW, int KPW32, int NT, bool NTW, bool ONE_M = false, bool XH = false>
__device__ __forceinline__ void gt_lean_core_bf16(const float* __restrict__ wp, const int tile0, const int ntile, const LeanX X, const int MT,
                                                  const int mchunk, const int nkb32, f32x4 (&acc0)[NT], f32x4 (&acc1)[NT]) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mt0 = ONE_M ? mchunk : mchunk * 2, mt1 = min(mt0 + 1, MT - 1);
    const uint4* wl = reinterpret_cast<const uint4*>(wp) + ((size_t)tile0 * nkb32 + wave) * 64 + lane;
    const LeanXR XR = gt_x_rsrc(X);
    uint4 b[KPW32][NT];
    float4 x0[XH ? 1 : KPW32][2], x1[XH ? 1 : KPW32][2];
    u32x4 h0[XH ? KPW32 : 1], h1[XH ? KPW32 : 1];
#pragma unroll
    for (int i = 0; i < KPW32; ++i) {
        const int kb32 = wave + i * NW;                     // wave-uniform
#pragma unroll
        for (int j = 0; j < NT; ++j) b[i][j] = make_uint4(0u, 0u, 0u, 0u);
        if constexpr (XH) { h0[i] = u32x4{0u, 0u, 0u, 0u}; h1[i] = u32x4{0u, 0u, 0u, 0u}; }
        else {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) { x0[i][hf] = make_float4(0.f, 0.f, 0.f, 0.f); x1[i][hf] = make_float4(0.f, 0.f, 0.f, 0.f); }
        }
        if (kb32 < nkb32) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (NT == 1 || j < ntile) {
                    const uint4* src = wl + ((size_t)j * nkb32 + i * NW) * 64;
                    if (NTW) {
                        const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src));
                        b[i][j] = make_uint4(t[0], t[1], t[2], t[3]);
                    } else {
                        b[i][j] = *src;
                    }
                }
            }
            if constexpr (XH) {
                h0[i] = gt_xload_h(X, kb32, MT, mt0);
                if (!ONE_M) h1[i] = gt_xload_h(X, kb32, MT, mt1);
            } else {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int kb = 2 * kb32 + hf;
                    x0[i][hf] = gt_xload(XR, X, kb, MT, mt0);
                    if (!ONE_M) x1[i][hf] = gt_xload(XR, X, kb, MT, mt1);
                }
            }
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < KPW32; ++i) {
        if (wave + i * NW < nkb32) {
            bf16x8 a0, a1;
            if constexpr (XH) {
                __builtin_memcpy(&a0, &h0[i], 16);
                __builtin_memcpy(&a1, &h1[i], 16);
            } else {
                a0[0] = (__bf16)x0[i][0].x; a0[1] = (__bf16)x0[i][0].y; a0[2] = (__bf16)x0[i][0].z; a0[3] = (__bf16)x0[i][0].w;
                a0[4] = (__bf16)x0[i][1].x; a0[5] = (__bf16)x0[i][1].y; a0[6] = (__bf16)x0[i][1].z; a0[7] = (__bf16)x0[i][1].w;
                if (!ONE_M) {
                    a1[0] = (__bf16)x1[i][0].x; a1[1] = (__bf16)x1[i][0].y; a1[2] = (__bf16)x1[i][0].z; a1[3] = (__bf16)x1[i][0].w;
                    a1[4] = (__bf16)x1[i][1].x; a1[5] = (__bf16)x1[i][1].y; a1[6] = (__bf16)x1[i][1].z; a1[7] = (__bf16)x1[i][1].w;
                }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bf16x8 bw;
                __builtin_memcpy(&bw, &b[i][j], 16);
                acc0[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bw, acc0[j], 0, 0, 0);
                if (!ONE_M) acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bw, acc1[j], 0, 0, 0);
            }
        }
    }
}

// accumulators -> LDS part[j][wave][32 rows][17] (C/D layout of 16x16x4: col = lane & 15, row = (lane >> 4) * 4 + reg)
template <int NW, int NT>
__device__ __forceinline__ void gt_lean_spill(float* lds, const f32x4 (&acc0)[NT], const f32x4 (&acc1)[NT]) {
    float (*part)[NW][32][17] = reinterpret_cast<float (*)[NW][32][17]>(lds);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            part[j][wave][q * 4 + v][r] = acc0[j][v];
            part[j][wave][16 + q * 4 + v][r] = acc1[j][v];
        }
    }
}

// Recurrent-half worker job: NT adjacent tiles of  h . W_h + b  written as pre-activation partial sums in tile order
// [tile][MT*16 rows][16 cols] (the consumer is gt_lstm_x_kernel).  One pass over the activations for all NT tiles.
template <int NW, int KPW, int NT, bool BF16 = false, bool ONE_M = false>
__device__ __forceinline__ void gt_lean_partial(const LeanPartialArgs& A, const int tile0, const int ntile, const int mchunk, float* lds) {
    // ONE_M: `mchunk` is a 16-row M-tile and the job covers those 16 rows only (half the state pull; for launches with
    // spare CUs, where two lighter workgroups per tile finish sooner than one)
    constexpr int ROWS = ONE_M ? 16 : 32;
    constexpr int NE = (NT * ROWS * 16 + NW * 64 - 1) / (NW * 64);
    const int m0 = mchunk * ROWS;
    float bias_v[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = threadIdx.x + i * NW * 64;
        const int j = e / (ROWS * 16);
        bias_v[i] = (e < NT * ROWS * 16 && j < ntile) ? A.bias[(tile0 + j) * 16 + (e & 15)] : 0.f;
    }
    f32x4 acc0[NT], acc1[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { acc0[j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // fp32: default cache policy, NOT non-temporal, although every weight byte is read by exactly one CU once per step: the
    // whole per-step working set (58 MB of weights) is re-read 500 times and fits the 256 MB Infinity Cache, and with the
    // non-temporal hint the workers' 34 MB per step came from HBM instead (front launch +0.8 us, projection launch +0.5 us).
    // bf16 (half the bytes): measured the other way round, non-temporal 1 % faster.
    if (BF16) gt_lean_core_bf16<NW, KPW, NT, true, ONE_M>(A.wp, tile0, ntile, LeanX{A.x, A.x, 2 * NW * KPW}, A.MT, mchunk, NW * KPW, acc0, acc1);
    else gt_lean_core<NW, KPW, NT, false, ONE_M>(A.wp, tile0, ntile, LeanX{A.x, A.x, NW * KPW}, A.MT, mchunk, acc0, acc1);
    gt_lean_spill<NW, NT>(lds, acc0, acc1);
    __syncthreads();
    const float (*part)[NW][32][17] = reinterpret_cast<const float (*)[NW][32][17]>(lds);
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = threadIdx.x + i * NW * 64;
        const int j = e / (ROWS * 16), row = (e >> 4) & (ROWS - 1), col = e & 15;
        if (e < NT * ROWS * 16 && j < ntile && m0 + row < A.MT * 16) {
            float z = bias_v[i];
#pragma unroll
            for (int w = 0; w < NW; ++w) z += part[j][w][row][col];
            A.partial_out[((size_t)(tile0 + j) * A.MT * 16 + m0 + row) * 16 + col] = z;
        }
    }
}

// ======================================================================================================================
// Batches above 32 rows.  The bodies above own ONE 32-row chunk per workgroup, so a batch of 128 used to run four
// workgroups per tile, each streaming the tile's weights again (4 x 58 MB per decode step: the launches were bound by that
// stream).  Here a workgroup keeps its weight fragments in REGISTERS and loops over chunks [c0, c1) of the batch: weights
// are read once per step at any batch, per chunk only the activations move.
//
// At these sizes the launches are no longer latency-bound: fp32 MFMA time and the activation pull of a CU are of the same
// order (a 16x16x4 MFMA eats 256 B of A operand in 32 cycles: 32 B/clk per CU when every A fragment is used for one tile,
// about what a CU takes in from L2), so a workgroup multiplies each activation fragment with NT = 2 tiles where the grid
// allows it, and the loads run UNDER the MFMAs (see the request order at gt_lean_mc).  Every load is unconditional (the
// last chunk re-reads itself) so the waits stay counted (front_lean.h explains why that matters).
//
// Per chunk the arithmetic is the single-chunk bodies' exactly -- same k-block -> wave assignment, every accumulator sees
// its k-blocks in the same ascending order, same summation order over waves -- so results stay bitwise those of the
// general kernels.
// ======================================================================================================================
#define GT_PIN_ORDER()                     \
    do {                                   \
        asm volatile("" ::: "memory");     \
        __builtin_amdgcn_sched_barrier(0); \
    } while (0)

// Chunks [c0, c1) of 32 rows for tiles tile0 .. tile0 + NT - 1.  `pre(mc)` requests whatever the epilogue of chunk mc needs
// (it is called before that chunk's MFMAs), `epi(mc)` runs after the chunk's partial sums are in LDS (gt_lean_spill layout,
// part[j][wave][32 rows][17]) and a workgroup barrier.  bf16: KPW counts 32-k blocks and nkb32 guards the last ones.
//
// Request order (measured with in-kernel stamps at 128 rows, K = 1024: a CU's load pipe moves 64 B/clk, so the 256 KB a
// workgroup needs before its first chunk is done are ~2.7 us of pipe time, and a wave that requests its whole K range at once
// makes the LAST wave's first byte wait for all of it): every wave requests k-block by k-block -- weights of k-block i, then the
// two M-tiles' activations of k-block i -- so all waves' k-block 0 is served first and the MFMAs of k-block i run while k-block
// i + 1 .. are still arriving; and inside the loop the NEXT chunk's activations of k-block i are requested right behind the
// MFMAs that consumed k-block i, one request per 8 NT MFMAs instead of a burst that would stall every wave's MFMA issue at once.
// Weight fragments of a workgroup's NT tiles for its wave's k-blocks: float4 (fp32) or uint4 (bf16 pack) per (k-block, tile).
template <int KPW, int NT, bool BF16>
struct LeanW {
    float4 f[BF16 ? 1 : KPW][BF16 ? 1 : NT];
    u32x4 h[BF16 ? KPW : 1][BF16 ? NT : 1];
};
template <int NW, int KPW, int NT, bool BF16, bool NTW>
__device__ __forceinline__ void gt_lean_mc_load_w(const float* __restrict__ wp, const int tile0, const int ntile, const int nkb32,
                                                  LeanW<KPW, NT, BF16>& W) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (BF16) {
        const u32x4* wl = reinterpret_cast<const u32x4*>(wp) + ((size_t)tile0 * nkb32 + wave) * 64 + lane;
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            const int ic = (wave + i * NW < nkb32) ? i : 0;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const u32x4* src = wl + ((size_t)(j < ntile ? j : 0) * nkb32 + ic * NW) * 64;
                W.h[i][j] = NTW ? __builtin_nontemporal_load(src) : *src;
            }
        }
    } else {
        constexpr int NKB = NW * KPW;
        const float4* wl = reinterpret_cast<const float4*>(wp) + ((size_t)tile0 * NKB + wave) * 64 + lane;
#pragma unroll
        for (int i = 0; i < KPW; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) W.f[i][j] = wl[((size_t)(j < ntile ? j : 0) * NKB + i * NW) * 64];
    }
}

// PRE: the weight fragments were requested by the caller (gt_lean_mc_load_w) -- the fused LSTM launch does that before it waits
// for the other workgroups' layer-1 state -- and only the activations are requested here.
// XH (bf16 only): the activations come from the bf16 mirrors (X.ha, X.hb).
template <int NW, int KPW, int NT, bool BF16, bool NTW, bool PRE, bool XH, class Pre, class Epi>
__device__ __forceinline__ void gt_lean_mc_impl(const float* __restrict__ wp, const int tile0, const int ntile, const LeanX X, const int nkb32,
                                                const int MT, const int c0, const int c1, float* lds, Pre pre, Epi epi,
                                                unsigned long long* dbg, LeanW<KPW, NT, BF16>& WPRE) {
    // diagnostic stamps of block 0 (tools/stamps_batch.py): 0 = first chunk's loads requested, 2 = its MFMAs issued (+ the next
    // chunk's loads requested), 3 = its partial sums in LDS (barrier passed), 5 = its epilogue done; 6 = end
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if constexpr (BF16) {
        u32x4 (&b)[KPW][NT] = WPRE.h;
        float4 xa[XH ? 1 : KPW][2], xb[XH ? 1 : KPW][2];
        u32x4 ha[XH ? KPW : 1], hb[XH ? KPW : 1];
        const u32x4* wl = reinterpret_cast<const u32x4*>(wp) + ((size_t)tile0 * nkb32 + wave) * 64 + lane;
        const LeanXR XR = gt_x_rsrc(X);
        auto xld = [&](const int i, const int hf, const int mt) {
            const int kb32 = (wave + i * NW < nkb32) ? wave + i * NW : wave;    // wave-uniform; past the end: re-read, never multiplied
            return gt_xload(XR, X, 2 * kb32 + hf, MT, mt);
        };
        // k-block i's fragments of M-tiles ma / mb: two fp32 fragments each, or one bf16 fragment each from the mirrors
        auto xreq = [&](const int i, const int ma, const int mb) {
            if constexpr (XH) {
                const int kb32 = (wave + i * NW < nkb32) ? wave + i * NW : wave;
                ha[i] = gt_xload_h(X, kb32, MT, ma);
                hb[i] = gt_xload_h(X, kb32, MT, mb);
            } else {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) { xa[i][hf] = xld(i, hf, ma); xb[i][hf] = xld(i, hf, mb); }
            }
        };
        {
            const int ma = 2 * c0, mb = min(2 * c0 + 1, MT - 1);
#pragma unroll
            for (int i = 0; i < KPW; ++i) {
                const int ic = (wave + i * NW < nkb32) ? i : 0;
                if constexpr (!PRE) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const u32x4* src = wl + ((size_t)(j < ntile ? j : 0) * nkb32 + ic * NW) * 64;
                        b[i][j] = NTW ? __builtin_nontemporal_load(src) : *src;
                    }
                }
                xreq(i, ma, mb);
            }
        }
        GT_PIN_ORDER();
        GT_STAMP(dbg, 0);
        for (int mc = c0; mc < c1; ++mc) {
            const int mn = min(mc + 1, c1 - 1);
            const int ma = 2 * mn, mb = min(2 * mn + 1, MT - 1);
            pre(mc);
            f32x4 acc0[NT], acc1[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) { acc0[j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            GT_PIN_ORDER();
#pragma unroll
            for (int i = 0; i < KPW; ++i) {
                if (wave + i * NW < nkb32) {
                    bf16x8 a0, a1;
                    if constexpr (XH) {
                        __builtin_memcpy(&a0, &ha[i], 16);
                        __builtin_memcpy(&a1, &hb[i], 16);
                    } else {
                        a0[0] = (__bf16)xa[i][0].x; a0[1] = (__bf16)xa[i][0].y; a0[2] = (__bf16)xa[i][0].z; a0[3] = (__bf16)xa[i][0].w;
                        a0[4] = (__bf16)xa[i][1].x; a0[5] = (__bf16)xa[i][1].y; a0[6] = (__bf16)xa[i][1].z; a0[7] = (__bf16)xa[i][1].w;
                        a1[0] = (__bf16)xb[i][0].x; a1[1] = (__bf16)xb[i][0].y; a1[2] = (__bf16)xb[i][0].z; a1[3] = (__bf16)xb[i][0].w;
                        a1[4] = (__bf16)xb[i][1].x; a1[5] = (__bf16)xb[i][1].y; a1[6] = (__bf16)xb[i][1].z; a1[7] = (__bf16)xb[i][1].w;
                    }
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        bf16x8 bw;
                        __builtin_memcpy(&bw, &b[i][j], 16);
                        acc0[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bw, acc0[j], 0, 0, 0);
                        acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bw, acc1[j], 0, 0, 0);
                    }
                }
                GT_PIN_ORDER();
                xreq(i, ma, mb);
                GT_PIN_ORDER();
            }
            if (mc == c0) GT_STAMP(dbg, 2);
            gt_lean_spill<NW, NT>(lds, acc0, acc1);
            __syncthreads();
            if (mc == c0) GT_STAMP(dbg, 3);
            epi(mc);
            if (mc == c0) GT_STAMP(dbg, 5);
            if (mc + 1 < c1) __syncthreads();
        }
        GT_STAMP(dbg, 6);
    } else {
        constexpr int NKB = NW * KPW;
        float4 (&b)[KPW][NT] = WPRE.f;
        float4 xa[KPW], xb[KPW];
        const float4* wl = reinterpret_cast<const float4*>(wp) + ((size_t)tile0 * NKB + wave) * 64 + lane;
        const LeanXR XR = gt_x_rsrc(X);
        auto xld = [&](const int i, const int mt) { return gt_xload(XR, X, wave + i * NW, MT, mt); };
        {
            const int ma = 2 * c0, mb = min(2 * c0 + 1, MT - 1);
#pragma unroll
            for (int i = 0; i < KPW; ++i) {
                if constexpr (!PRE) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        // (a pair's second tile beyond the matrix re-reads the first: loaded, multiplied, never stored)
                        const float4* src = wl + ((size_t)(j < ntile ? j : 0) * NKB + i * NW) * 64;
                        if (NTW) {
                            const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
                            b[i][j] = make_float4(t[0], t[1], t[2], t[3]);
                        } else {
                            b[i][j] = *src;
                        }
                    }
                }
                xa[i] = xld(i, ma);
                xb[i] = xld(i, mb);
            }
        }
        GT_PIN_ORDER();
        GT_STAMP(dbg, 0);
        for (int mc = c0; mc < c1; ++mc) {
            const int mn = min(mc + 1, c1 - 1);
            const int ma = 2 * mn, mb = min(2 * mn + 1, MT - 1);
            pre(mc);
            f32x4 acc0[NT], acc1[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) { acc0[j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            GT_PIN_ORDER();
#pragma unroll
            for (int i = 0; i < KPW; ++i) {
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc0[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i].x, b[i][j].x, acc0[j], 0, 0, 0);
                    acc1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[i].x, b[i][j].x, acc1[j], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc0[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i].y, b[i][j].y, acc0[j], 0, 0, 0);
                    acc1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[i].y, b[i][j].y, acc1[j], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc0[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i].z, b[i][j].z, acc0[j], 0, 0, 0);
                    acc1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[i].z, b[i][j].z, acc1[j], 0, 0, 0);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc0[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[i].w, b[i][j].w, acc0[j], 0, 0, 0);
                    acc1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(xb[i].w, b[i][j].w, acc1[j], 0, 0, 0);
                }
                GT_PIN_ORDER();
                xa[i] = xld(i, ma);       // the next chunk's k-block i (the last chunk re-reads its own: no branch, counted waits)
                xb[i] = xld(i, mb);
                GT_PIN_ORDER();
            }
            if (mc == c0) GT_STAMP(dbg, 2);
            gt_lean_spill<NW, NT>(lds, acc0, acc1);
            __syncthreads();
            if (mc == c0) GT_STAMP(dbg, 3);
            epi(mc);
            if (mc == c0) GT_STAMP(dbg, 5);
            if (mc + 1 < c1) __syncthreads();
        }
        GT_STAMP(dbg, 6);
    }
}

template <int NW, int KPW, int NT, bool BF16, bool NTW, class Pre, class Epi>
__device__ __forceinline__ void gt_lean_mc(const float* __restrict__ wp, const int tile0, const int ntile, const LeanX X, const int nkb32,
                                           const int MT, const int c0, const int c1, float* lds, Pre pre, Epi epi,
                                           unsigned long long* dbg = nullptr) {
    LeanW<KPW, NT, BF16> W;
    if (BF16 && X.ha) gt_lean_mc_impl<NW, KPW, NT, BF16, NTW, false, BF16>(wp, tile0, ntile, X, nkb32, MT, c0, c1, lds, pre, epi, dbg, W);
    else gt_lean_mc_impl<NW, KPW, NT, BF16, NTW, false, false>(wp, tile0, ntile, X, nkb32, MT, c0, c1, lds, pre, epi, dbg, W);
}

// Recurrent-half worker job over chunks [c0, c1): gt_lean_partial's result for each, the job's weights read once.
// (bf16: KPW counts 32-k blocks, as in gt_lean_partial.)
template <int NW, int KPW, int NT, bool BF16 = false>
__device__ __forceinline__ void gt_lean_partial_mc(const LeanPartialArgs& A, const int tile0, const int ntile, const int c0, const int c1,
                                                   float* lds) {
    constexpr int NE = (NT * 32 * 16 + NW * 64 - 1) / (NW * 64);
    float bias_v[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = threadIdx.x + i * NW * 64;
        const int j = e / (32 * 16);
        bias_v[i] = (e < NT * 32 * 16 && j < ntile) ? A.bias[(tile0 + j) * 16 + (e & 15)] : 0.f;
    }
    const float (*part)[NW][32][17] = reinterpret_cast<const float (*)[NW][32][17]>(lds);
    auto epi = [&](const int mc) {
        const int m0 = mc * 32;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = threadIdx.x + i * NW * 64;
            const int j = e / (32 * 16), row = (e >> 4) & 31, col = e & 15;
            if (e < NT * 32 * 16 && j < ntile && m0 + row < A.MT * 16) {
                float z = bias_v[i];
#pragma unroll
                for (int w = 0; w < NW; ++w) z += part[j][w][row][col];
                A.partial_out[((size_t)(tile0 + j) * A.MT * 16 + m0 + row) * 16 + col] = z;
            }
        }
    };
    // fp32: default cache policy, bf16: non-temporal -- as in gt_lean_partial
    gt_lean_mc<NW, KPW, NT, BF16, BF16>(A.wp, tile0, ntile, LeanX{A.x, A.x, BF16 ? 2 * NW * KPW : NW * KPW, A.xh, A.xh}, NW * KPW, A.MT, c0, c1,
                                        lds, [](int) {}, epi);
}
