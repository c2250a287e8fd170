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
};

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
        const float* xs = kb < X.nkb_a ? X.xa + (size_t)kb * MT * 256 : X.xb + (size_t)(kb - X.nkb_a) * MT * 256;
        x0[i] = *reinterpret_cast<const float4*>(xs + mt0 * 256 + lane * 4);
        if (!ONE_M) x1[i] = *reinterpret_cast<const float4*>(xs + mt1 * 256 + lane * 4);
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
template <int NW, int KPW32, int NT, bool NTW, bool ONE_M = false>
__device__ __forceinline__ void gt_lean_core_bf16(const float* __restrict__ wp, const int tile0, const int ntile, const LeanX X, const int MT,
                                                  const int mchunk, const int nkb32, f32x4 (&acc0)[NT], f32x4 (&acc1)[NT]) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mt0 = ONE_M ? mchunk : mchunk * 2, mt1 = min(mt0 + 1, MT - 1);
    const uint4* wl = reinterpret_cast<const uint4*>(wp) + ((size_t)tile0 * nkb32 + wave) * 64 + lane;
    uint4 b[KPW32][NT];
    float4 x0[KPW32][2], x1[KPW32][2];
#pragma unroll
    for (int i = 0; i < KPW32; ++i) {
        const int kb32 = wave + i * NW;                     // wave-uniform
#pragma unroll
        for (int j = 0; j < NT; ++j) b[i][j] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) { x0[i][hf] = make_float4(0.f, 0.f, 0.f, 0.f); x1[i][hf] = make_float4(0.f, 0.f, 0.f, 0.f); }
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
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int kb = 2 * kb32 + hf;
                const float* xs = kb < X.nkb_a ? X.xa + (size_t)kb * MT * 256 : X.xb + (size_t)(kb - X.nkb_a) * MT * 256;
                x0[i][hf] = *reinterpret_cast<const float4*>(xs + mt0 * 256 + lane * 4);
                if (!ONE_M) x1[i][hf] = *reinterpret_cast<const float4*>(xs + mt1 * 256 + lane * 4);
            }
        }
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < KPW32; ++i) {
        if (wave + i * NW < nkb32) {
            bf16x8 a0, a1;
            a0[0] = (__bf16)x0[i][0].x; a0[1] = (__bf16)x0[i][0].y; a0[2] = (__bf16)x0[i][0].z; a0[3] = (__bf16)x0[i][0].w;
            a0[4] = (__bf16)x0[i][1].x; a0[5] = (__bf16)x0[i][1].y; a0[6] = (__bf16)x0[i][1].z; a0[7] = (__bf16)x0[i][1].w;
            if (!ONE_M) {
                a1[0] = (__bf16)x1[i][0].x; a1[1] = (__bf16)x1[i][0].y; a1[2] = (__bf16)x1[i][0].z; a1[3] = (__bf16)x1[i][0].w;
                a1[4] = (__bf16)x1[i][1].x; a1[5] = (__bf16)x1[i][1].y; a1[6] = (__bf16)x1[i][1].z; a1[7] = (__bf16)x1[i][1].w;
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
