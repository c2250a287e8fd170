// Winograd F(4,5) / F(2,5) five-tap Conv1D (+ folded BN + activation + residual) with its transform-domain GEMMs on the BF16 matrix
// pipe at fp32 accuracy: split-bf16 x6, fp32 accumulation.  Replaces gemm_conv.hip's fp32 Winograd kernel for the postnet
// (reference Taco2.py:131-149,230) where it is faster; same algorithm, same transforms, same epilogue.
//
// Why (tools/split_bf16.hip, profiles/r06_split_bf16.txt): on gfx950 v_mfma_f32_32x32x2_f32 runs at 1/16 the rate of
// v_mfma_f32_32x32x16_bf16 and there is no xf32.  Every fp32 operand is written as three bf16 planes, v = hi + mid + lo with
// hi = rne(v), mid = rne(v - hi), lo = rne(v - hi - mid) (8 + 8 + 8 significand bits: both subtractions are exact), and a product
// a . b as the six plane products down to 2^-24: hh, hm, mh, mm, hl, lh (what is dropped -- ml, lm, ll -- is below 2^-26 of |a b|).
// Six bf16 MFMAs per 16 k against eight fp32 MFMAs of twice their length: measured 2.26 x the fp32 matrix rate with every plane
// re-read from LDS, and on K = 2 048 dot products an error of 3.0 (one accumulator) against 2.7 (the fp32 MFMA chain) units of
// 2^-24 sum|a b|: the bf16 instruction sums 32 products per rounding where the fp32 chain rounds after every four.
//   U planes: formed at finalize from the float64 transform of the weights (gsttaco.cpp add_wino_split), [xi][plane][n][k], k contiguous.
//   V planes: the input transform runs in fp32 as before (wino_common.h); each thread splits its four transformed values on their way
//             into LDS -- a value is split ONCE per (tile, channel, xi) and multiplied with 128 columns.
// Structure (the fp32 kernel's, whose comments say why): one workgroup = 64 tiles x 128 columns, 8 waves as 2 x 4 MFMA tiles of 32 x 32,
// one accumulator tile per transform-domain GEMM, slice-major steps (xi, 32-channel slice) on two LDS stages with one barrier per step,
// the next step's B planes and the next slice's raw taps in flight under the MFMAs, every load unconditional.
// LDS per stage: A planes [3][64 rows][32 k + 8] bf16 (rows 80 bytes apart: the 32 rows of a fragment read hit all banks), B planes
// [3][128 columns][32 k] bf16 with the four 16-byte chunks of a column XOR-swizzled by (column / 8) % 4: a ds_read_b128 is served in four
// groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- over 64 banks, and the column quads {0, 3, 5, 6} /
// {1, 2, 4, 7} of a group then sit at four different chunk positions (unpadded, conflict-free for the reads and the 16-byte stores).
#include "wino_common.h"
#include <type_traits>

// (ablation switches of tools/wino_split_bench.hip: -DWS_NO_MFMA, -DWS_NO_SPLIT; the product compiles the plain forms)

namespace {

constexpr int WS_BMP = 64, WS_BN = 128, WS_BK = 32, WS_LDA = WS_BK + 8;
constexpr int WS_A_STAGE = 3 * WS_BMP * WS_LDA, WS_B_STAGE = 3 * WS_BN * WS_BK;      // bf16 elements
constexpr int WS_LDS_BYTES = (2 * WS_A_STAGE + 3 * WS_B_STAGE) * 2;      // A: two stages; B: three (its planes are requested two steps ahead)

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// 16 bytes per lane from a buffer straight into LDS: lane l's data lands at lds_addr + 16 l (lds_addr wave-uniform).  Inline asm, as in
// gemm_conv.hip's five-tap bf16 kernel (through the builtin the compiler orders every later LDS read behind the DMA with a vmcnt(0));
// the caller waits (s_waitcnt vmcnt(0)) and synchronises before the data is read.
__device__ __forceinline__ void ws_lds_dma16(__amdgpu_buffer_rsrc_t rs, const uint32_t lds_addr, const uint32_t voff, const uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
#pragma clang diagnostic pop

// v = hi + mid + lo (bf16 each, round to nearest even); the two remainders are exact in fp32
__device__ __forceinline__ void ws_split(const float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

// X3 (knob GSTTACO_WINO_SPLIT=3, NOT the default): only the planes hi and mid of both operands and the products hh, hm, mh -- what the review
// priced as "three products = 5.3 x at ~2^-16": half the MFMAs, two thirds of the plane traffic; what is dropped (mm, hl, lh) is ~2^-17 of
// |a b| per product, 9.3 units of 2^-24 sum|a b| on K = 2 048 dot products against x6's 3.0 and the fp32 chain's 2.7 (tools/split_bf16.hip).
template <int MO, bool X3>
__global__ __launch_bounds__(WT, 2) void gt_conv_wino5s_kernel(ConvGemmArgs A, const __bf16* __restrict__ Us, const int npad) {
    constexpr int AL = Wino<MO>::ALPHA;
    constexpr int NPL = X3 ? 2 : 3;             // planes moved and multiplied
    extern __shared__ __attribute__((aligned(16))) __bf16 ws_lds[];
    __bf16* As = ws_lds;                            // [2][3][64][WS_LDA]
    __bf16* Bs = ws_lds + 2 * WS_A_STAGE;           // [3][3][128][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int Pu = (A.T + MO - 1) / MO;                  // tiles per utterance
    const int Ptot = A.B * Pu;
    // workgroup -> (row block, column block): the column blocks of one row block on ONE XCD at the same time (gemm_conv.hip)
    const int ncb = (A.N + WS_BN - 1) / WS_BN;
    const int wi = blockIdx.x >> 3;
    const int rb = (wi / ncb) * 8 + (blockIdx.x & 7), cb = wi % ncb;
    if (rb * WS_BMP >= Ptot) return;
    const int p0 = rb * WS_BMP, n0 = cb * WS_BN;
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.x), 0, (int)((size_t)A.B * A.T * A.Cin * 4), 0x00020000);
    // this thread's A element: tile row tid >> 3, channel quad tid & 7
    int first, len;
    uint32_t voff;
    {
        const int p = p0 + (tid >> 3);
        const bool ok = p < Ptot;
        const int pp = ok ? p : 0;
        const int b = pp / Pu;
        first = MO * (pp - b * Pu) - 2;
        len = ok ? (A.row_len ? min(A.T, A.row_len[b]) : A.T) : 0;      // a tile past the end reads nothing
        voff = (uint32_t)(((int64_t)b * A.T + first) * A.Cin + (tid & 7) * 4) * 4u;
    }
    f32x16 M[AL];
#pragma unroll
    for (int xi = 0; xi < AL; ++xi)
#pragma unroll
        for (int e = 0; e < 16; ++e) M[xi][e] = 0.f;
    const int kh = lane >> 5, l31 = lane & 31;
    const int nsl = A.wino_cin / WS_BK;                   // even, >= 4
    int cur = 0;

    // B planes of a step: straight from memory into LDS (buffer_load ... lds: no staging registers, no ds_write -- the 16-byte LDS stores
    // of 24 KB per step were 300 cycles of the CU's store path).  A DMA instruction writes its 64 lanes' 16 bytes side by side = 16
    // columns of one plane; wave w fills columns 16 w .. 16 w + 15 of each plane; lane l: column l / 4 of the sixteen, position l % 4, for
    // which it FETCHES logical chunk (l % 4) ^ swizzle(column) -- the swizzle costs the DMA nothing and the fragment reads undo it.
    const auto rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Us), 0, (int)((size_t)AL * 3 * npad * A.wino_cin * 2), 0x00020000);
    const int bcol = __builtin_amdgcn_readfirstlane(wave) * 16 + (lane >> 2);
    const uint32_t vb = (uint32_t)((((n0 + bcol) * A.wino_cin) + (((lane & 3) ^ ((bcol >> 3) & 3)) * 8)) * 2);
    const uint32_t lds_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)Bs + (uint32_t)(__builtin_amdgcn_readfirstlane(wave) * 16 * WS_BK * 2);
    const int a_st = (tid >> 3) * WS_LDA + (tid & 7) * 4;
    auto dma_b = [&](const int st, const int xi, const int c0) {
#pragma unroll
        for (int p = 0; p < NPL; ++p)
            ws_lds_dma16(rs_u, lds_b + (uint32_t)((st * WS_B_STAGE + p * WS_BN * WS_BK) * 2), vb, (uint32_t)((((size_t)(xi * 3 + p) * npad) * A.wino_cin + c0) * 2));
    };
    // the NEXT step's A planes, in four pieces that are dealt between the current step's MFMAs below: transform of channels (0, 1) /
    // (2, 3) of this thread's quad, split of each pair, one 8-byte store per plane
    auto xform2 = [&](auto xi_c, auto half_c, const float4 (&d)[AL]) {
        constexpr int XI = decltype(xi_c)::value, HALF = decltype(half_c)::value;
        f32x2 v = {0.f, 0.f};
#pragma unroll
        for (int tap = 0; tap < AL; ++tap) {
            const float cf = Wino<MO>::bt(XI, tap);
            if (cf != 0.f) v = __builtin_elementwise_fma((f32x2){cf, cf}, HALF ? (f32x2){d[tap].z, d[tap].w} : (f32x2){d[tap].x, d[tap].y}, v);
        }
        return v;
    };
    auto split2 = [&](const f32x2 v, bf16x2& h, bf16x2& m, bf16x2& l) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            __bf16 he, me, le;
#ifdef WS_NO_SPLIT
            he = (__bf16)v[e]; me = he; le = he;
#else
            ws_split(v[e], he, me, le);
#endif
            h[e] = he; m[e] = me; l[e] = le;
        }
    };
    auto store_planes = [&](const int st, const bf16x2 (&h)[2], const bf16x2 (&m)[2], const bf16x2 (&l)[2]) {
        __bf16* as = As + st * WS_A_STAGE + a_st;
        *reinterpret_cast<bf16x4*>(as) = bf16x4{h[0][0], h[0][1], h[1][0], h[1][1]};
        *reinterpret_cast<bf16x4*>(as + WS_BMP * WS_LDA) = bf16x4{m[0][0], m[0][1], m[1][0], m[1][1]};
        if (!X3) *reinterpret_cast<bf16x4*>(as + 2 * WS_BMP * WS_LDA) = bf16x4{l[0][0], l[0][1], l[1][0], l[1][1]};
    };
    // one step's products: per 16 k the three A and three B plane fragments, then hh, hm, mh, mm, hl, lh into the GEMM's accumulator
    const int a_rd = (wm * 32 + l31) * WS_LDA + kh * 8;
    const int bc = wn * 32 + l31, b_sw = (bc >> 3) & 3;
#define WS_RD(ks, p, WHICH)                                                                                        \
        do { if (WHICH & 1) a_[ks][p] = *reinterpret_cast<const bf16x8*>(ab_ + p * WS_BMP * WS_LDA + ks * 16);     \
             if (WHICH & 2) b_[ks][p] = *reinterpret_cast<const bf16x8*>(bb_ + p * WS_BN * WS_BK + (((2 * ks + kh) ^ b_sw) * 8)); } while (0)
#ifdef WS_NO_MFMA
#define WS_MFMA(ACC, x, y) ACC[0] += (float)(x)[0] + (float)(y)[1]
#else
#define WS_MFMA(ACC, x, y) ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, ACC, 0, 0, 0)
#endif
#define WS_MFMA6(ACC, x, y) do { if (!X3) WS_MFMA(ACC, x, y); } while (0)      /* (the products x6 has and x3 drops) */
#define WS_FENCE() __builtin_amdgcn_sched_barrier(0)
    // The step as four fenced segments of three dependent MFMAs each (96 cycles of the matrix pipe) with a quarter of the NEXT step's
    // transform + split (~10 vector instructions, independent of them) behind the segment's MFMAs: on the bf16 pipe a wave's vector
    // instructions run under its own MFMAs (under fp32 MFMAs they do not: tools/mfma_rate.hip).  Left to the scheduler -- also with
    // sched_group_barrier pipelines -- all of it lands behind the twelfth MFMA.
    // (Tried and removed, round 6: the workgroup as 32 tiles x 128 columns on 4 waves, B planes one step ahead in two stages = 63 KB of LDS, so
    // that TWO unsynchronised workgroups share a CU and one's MFMA phase runs beside the other's loads and LDS traffic -- inside one
    // workgroup the per-step barrier keeps every wave in the same phase and the step is the SUM of its MFMA, LDS and vector time.  Bitwise
    // the same results, 300 against 250 us per 512 -> 512 layer (postnet 1.15 against 0.99 ms): every B byte is then fetched for 32 tile
    // rows instead of 64, and the B planes' way into LDS is what the skeleton is bound by.)
    // (Tried and removed, tools/wino_split_bench -DWS_STAGGER, profiles/r06_wino_stagger.txt: the two waves of a SIMD running the step's halves
    // in opposite order -- one its twelve MFMAs first, the other its transform + split + store first -- instead of both interleaving them:
    // 457 against 250 us per 512 -> 512 layer.)
#define WS_BODY_INTERLEAVED(XI, XI1, DX)                                                                           \
        WS_RD(0, 0, 3); WS_RD(0, 1, 3); if (!X3) WS_RD(0, 2, 3);                                                   \
        WS_FENCE();                                                                                               \
        WS_MFMA(M[XI], a_[0][0], b_[0][0]); WS_MFMA(M[XI], a_[0][0], b_[0][1]); WS_MFMA(M[XI], a_[0][1], b_[0][0]); \
        const f32x2 v0_ = xform2(std::integral_constant<int, XI1>{}, std::integral_constant<int, 0>{}, DX);        \
        WS_FENCE();                                                                                               \
        WS_MFMA6(M[XI], a_[0][1], b_[0][1]); WS_MFMA6(M[XI], a_[0][0], b_[0][2]); WS_MFMA6(M[XI], a_[0][2], b_[0][0]); \
        WS_RD(1, 0, 3); WS_RD(1, 1, 3);       /* (the second 16 k: the low planes a segment later -- 16 registers less at the peak) */ \
        split2(v0_, h_[0], m_[0], l_[0]);                                                                          \
        WS_FENCE();                                                                                               \
        WS_MFMA(M[XI], a_[1][0], b_[1][0]); WS_MFMA(M[XI], a_[1][0], b_[1][1]); WS_MFMA(M[XI], a_[1][1], b_[1][0]); \
        if (!X3) WS_RD(1, 2, 3);                                                                                   \
        const f32x2 v1_ = xform2(std::integral_constant<int, XI1>{}, std::integral_constant<int, 1>{}, DX);        \
        WS_FENCE();                                                                                               \
        WS_MFMA6(M[XI], a_[1][1], b_[1][1]); WS_MFMA6(M[XI], a_[1][0], b_[1][2]); WS_MFMA6(M[XI], a_[1][2], b_[1][0]); \
        split2(v1_, h_[1], m_[1], l_[1]);                                                                          \
        store_planes(cur ^ 1, h_, m_, l_);                                                                         \
        WS_FENCE();
#define WS_BODY(XI, XI1, DX) WS_BODY_INTERLEAVED(XI, XI1, DX)
    float4 dE[AL], dO[AL];
#define WS_STEP(XI, DCUR, DNXT, DX, s_)                                                                            \
    {                                                                                                             \
        constexpr int XI1 = (XI + 1) % AL;                                                                         \
        /* (past the last step: a valid address whose data is never used) */                                      \
        constexpr int XI2 = (XI + 2) % AL;                                                                         \
        /* B planes of step g + 2 into the stage step g - 1 read (three stages: a step is ~0.5 us, shorter than an L2 round trip under */ \
        /* load -- requested one step ahead, the wait at the end of every step exposed it: 169 us of the 253 per 512 -> 512 layer) */ \
        const __bf16* ab_ = As + cur * WS_A_STAGE + a_rd;                                                          \
        const __bf16* bb_ = Bs + bcur * WS_B_STAGE + bc * WS_BK;                                                   \
        bf16x8 a_[2][3], b_[2][3];                                                                                 \
        bf16x2 h_[2], m_[2], l_[2];                                                                                \
        WS_BODY(XI, XI1, DX)                                                                                       \
        /* The step's requests go out at its END: the compiler counts only the loads it knows (the taps) and asks for "all but the n */ \
        /* youngest" when it consumes them -- with this step's DMAs already in the queue that wait covered them too (vmcnt(1) in */ \
        /* front of the slice's last transform: a DMA's whole latency, exposed in two or three of eight steps). */ \
        dma_b(bnx, XI2, min((s_) + (XI + 2 >= AL ? 1 : 0), nsl - 1) * WS_BK);                                      \
        if constexpr (XI == 0) wino_issue_taps<MO>(A, rs_x, voff, first, len, min((s_) + 1, nsl - 1) * WS_BK, (s_) + 1 < nsl, DNXT); \
        /* the NEXT step's B planes have landed (requested a step ago; loads complete in order): what may still be in flight is this */ \
        /* step's three DMAs and, at XI == 0 and 1, the next slice's AL tap loads requested behind XI == 0's DMAs */ \
        if constexpr (XI <= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPL + AL) : "memory");                    \
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPL) : "memory");                                           \
        __syncthreads();                                                                                          \
        cur ^= 1;                                                                                                 \
        bcur = bcur == 2 ? 0 : bcur + 1;                                                                           \
        bnx = bnx == 2 ? 0 : bnx + 1;                                                                              \
    }
    /* (DX: the tap set the NEXT step's transform reads -- this slice's until its last step, then the next slice's) */ \

#define WS_SLICE(DCUR, DNXT, s_)                                                                                   \
    WS_STEP(0, DCUR, DNXT, DCUR, s_) WS_STEP(1, DCUR, DNXT, DCUR, s_) WS_STEP(2, DCUR, DNXT, DCUR, s_) WS_STEP(3, DCUR, DNXT, DCUR, s_) \
    WS_STEP(4, DCUR, DNXT, DCUR, s_)                                                                               \
    if constexpr (AL == 8) { WS_STEP(5, DCUR, DNXT, DCUR, s_) WS_STEP(6 % AL, DCUR, DNXT, DCUR, s_) WS_STEP(7 % AL, DCUR, DNXT, DNXT, s_) } \
    else { WS_STEP(5, DCUR, DNXT, DNXT, s_) }
    wino_issue_taps<MO>(A, rs_x, voff, first, len, 0, true, dE);
    dma_b(0, 0, 0);
    dma_b(1, 1, 0);
    int bcur = 0, bnx = 2;                  // B stage of the current step / of the step two ahead (scalar, cycling 0 1 2)
    {
        bf16x2 h_[2], m_[2], l_[2];
        split2(xform2(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, dE), h_[0], m_[0], l_[0]);
        split2(xform2(std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{}, dE), h_[1], m_[1], l_[1]);
        store_planes(0, h_, m_, l_);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < nsl; s += 2) {
        WS_SLICE(dE, dO, s)
        WS_SLICE(dO, dE, s + 1)
    }
#undef WS_SLICE
#undef WS_STEP
#undef WS_BODY
#undef WS_BODY_INTERLEAVED
#undef WS_RD
#undef WS_MFMA
#undef WS_MFMA6
#undef WS_FENCE

    // epilogue (the fp32 kernel's); 32x32 C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5); tile row -> MO output rows
    const int n = n0 + wn * 32 + l31;
    if (n < A.N) {
        const float sc = A.scale ? A.scale[n] : 1.f;
        const float sh = A.shift ? A.shift[n] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int p = p0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            if (p >= Ptot) continue;
            const int b = p / Pu, t0 = MO * (p - b * Pu);
#pragma unroll
            for (int o = 0; o < MO; ++o) {
                const int t = t0 + o;
                if (t >= A.T) continue;
                const int64_t m = (int64_t)b * A.T + t;
                float y = 0.f;                         // output transform: y_o = sum_xi AT[o][xi] M_xi
#pragma unroll
                for (int xi = 0; xi < AL; ++xi)
                    if (Wino<MO>::at(o, xi) != 0.f) y += Wino<MO>::at(o, xi) * M[xi][e];
                float v = y * sc + sh;
                if (A.rowbias) v += A.rowbias[(int64_t)b * A.N + n];
                if (A.act == ACT_RELU) v = fmaxf(v, 0.f);
                else if (A.act == ACT_TANH) v = gt_tanh(v);
                if (A.res) v += A.res[m * A.ldo + n];
                A.out[m * A.ldo + n] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------- plain GEMM, split-bf16 x6
// out[M, N] = act(scale . (x[M, K] . W[K, N]) + shift + rowbias) for the path's tall plain GEMMs -- the BiLSTM's hoisted input halves
// (4 096 x 512 x 2 048 at the headline shape: 109 us on the fp32 matrix pipe) and the attention's Value projection (Steps.py:123,
// hoisted) -- on the bf16 pipe at fp32 accuracy, with the Winograd kernel's machinery minus its transforms: a workgroup = 64 rows x 128
// columns, 8 waves as 2 x 4 MFMA tiles of 32 x 32, steps of 32 k.  Everything that comes from memory comes by LDS-DMA: the step's three
// W planes (24 KB, pre-split at finalize, [plane][n][k]) two steps ahead into a three-slot ring, and the step's RAW fp32 x tile (8 KB)
// THREE steps ahead into a four-slot ring -- a thread reads back the 16 bytes it fetched itself (behind its own vmcnt wait: no barrier)
// one step BEFORE the step they belong to, splits them into the three bf16 planes and stores those for that step, dealt between the
// current step's twelve MFMAs as in the Winograd kernel.  (The end-of-step wait leaves exactly the step's own four DMAs in flight, so
// what it guarantees is everything requested a step earlier: x at distance two was read while possibly still on its way -- one call
// in sixteen differed under four contexts on four streams.)
constexpr int GS_ARAW = WS_BMP * WS_BK;                                 // floats per raw x slot
constexpr int GS_LDS_BYTES = 4 * GS_ARAW * 4 + (2 * WS_A_STAGE + 3 * WS_B_STAGE) * 2;

__global__ __launch_bounds__(WT, 2) void gt_gemm_split_kernel(ConvGemmArgs A, const __bf16* __restrict__ Ws, const int npad) {
    extern __shared__ __attribute__((aligned(16))) __bf16 ws_lds[];
    float* Ar = reinterpret_cast<float*>(ws_lds);                        // [4][64 rows][32 k] raw fp32
    __bf16* As = ws_lds + 4 * GS_ARAW * 2;                                // [2][3 planes][64][WS_LDA]
    __bf16* Bs = As + 2 * WS_A_STAGE;                                     // [3][3 planes][128][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave >> 2, wn = wave & 3;
    const int M = A.B * A.T, K = A.Cin;
    const int ncb = (A.N + WS_BN - 1) / WS_BN;
    const int wi = blockIdx.x >> 3;
    const int rb = (wi / ncb) * 8 + (blockIdx.x & 7), cb = wi % ncb;
    if (rb * WS_BMP >= M) return;
    const int p0 = rb * WS_BMP, n0 = cb * WS_BN;
    const int kh = lane >> 5, l31 = lane & 31;
    const int nsl = K / WS_BK;
    // x by DMA: wave w fetches rows 8 w .. 8 w + 7 of the tile (lane l: row l / 8, 16-byte chunk l % 8); rows past M get an out-of-range
    // offset and land as zeros.  W planes as in the Winograd kernel (wave w: columns 16 w .. 16 w + 15 of each plane, chunks swizzled).
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.x), 0, (int)((size_t)M * K * 4), 0x00020000);
    const int xrow = p0 + wv * 8 + (lane >> 3);
    const uint32_t vx = xrow < M ? (uint32_t)(((size_t)xrow * K + (lane & 7) * 4) * 4) : 0x80000000u;
    const uint32_t lds_x = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)Ar + (uint32_t)(wv * 8 * WS_BK * 4);
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Ws), 0, (int)((size_t)3 * npad * K * 2), 0x00020000);
    const int bcol = wv * 16 + (lane >> 2);
    const uint32_t vb = (uint32_t)((((n0 + bcol) * K) + (((lane & 3) ^ ((bcol >> 3) & 3)) * 8)) * 2);
    const uint32_t lds_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)Bs + (uint32_t)(wv * 16 * WS_BK * 2);
    auto dma_x = [&](const int slot, const int c0) { ws_lds_dma16(rs_x, lds_x + (uint32_t)(slot * GS_ARAW * 4), vx, (uint32_t)(c0 * 4)); };
    auto dma_b = [&](const int slot, const int c0) {
#pragma unroll
        for (int p = 0; p < 3; ++p)
            ws_lds_dma16(rs_w, lds_b + (uint32_t)((slot * WS_B_STAGE + p * WS_BN * WS_BK) * 2), vb, (uint32_t)((((size_t)p * npad) * K + c0) * 2));
    };
    const int a_own = (wave * 8 + (lane >> 3)) * WS_BK + (lane & 7) * 4;           // this thread's 16 bytes of a raw slot (floats)
    const int a_st = (tid >> 3) * WS_LDA + (tid & 7) * 4;                          // ... and of a plane (row tid / 8 = 8 wave + lane / 8)
    const int a_rd = (wm * 32 + l31) * WS_LDA + kh * 8;
    const int bc = wn * 32 + l31, b_sw = (bc >> 3) & 3;
    auto split2 = [&](const float v0, const float v1, bf16x2& h, bf16x2& m, bf16x2& l) {
        __bf16 he, me, le;
        ws_split(v0, he, me, le); h[0] = he; m[0] = me; l[0] = le;
        ws_split(v1, he, me, le); h[1] = he; m[1] = me; l[1] = le;
    };
    auto store_planes = [&](const int st, const bf16x2 (&h)[2], const bf16x2 (&m)[2], const bf16x2 (&l)[2]) {
        __bf16* as = As + st * WS_A_STAGE + a_st;
        *reinterpret_cast<bf16x4*>(as) = bf16x4{h[0][0], h[0][1], h[1][0], h[1][1]};
        *reinterpret_cast<bf16x4*>(as + WS_BMP * WS_LDA) = bf16x4{m[0][0], m[0][1], m[1][0], m[1][1]};
        *reinterpret_cast<bf16x4*>(as + 2 * WS_BMP * WS_LDA) = bf16x4{l[0][0], l[0][1], l[1][0], l[1][1]};
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // prologue: x of steps 0..2 and W of steps 0, 1 requested; step 0's x split
    dma_x(0, 0);
    dma_x(1, min(1, nsl - 1) * WS_BK);
    dma_x(2, min(2, nsl - 1) * WS_BK);
    dma_b(0, 0);
    dma_b(1, min(1, nsl - 1) * WS_BK);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        const float4 v = *reinterpret_cast<const float4*>(Ar + a_own);
        bf16x2 h_[2], m_[2], l_[2];
        split2(v.x, v.y, h_[0], m_[0], l_[0]);
        split2(v.z, v.w, h_[1], m_[1], l_[1]);
        store_planes(0, h_, m_, l_);
    }
    __syncthreads();
    int cur = 0, s3 = 0, s3nn = 2, x4n = 1, x4nnn = 3;     // plane stage of the step; W ring slots of steps s / s + 2; x ring slots of steps s + 1 / s + 3
#define GS_RD(ks, p, WHICH)                                                                                        \
        do { if (WHICH & 1) a_[ks][p] = *reinterpret_cast<const bf16x8*>(ab_ + p * WS_BMP * WS_LDA + ks * 16);     \
             if (WHICH & 2) b_[ks][p] = *reinterpret_cast<const bf16x8*>(bb_ + p * WS_BN * WS_BK + (((2 * ks + kh) ^ b_sw) * 8)); } while (0)
#define GS_MFMA(x, y) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc, 0, 0, 0)
    for (int s = 0; s < nsl; ++s) {
        dma_x(x4nnn, min(s + 3, nsl - 1) * WS_BK);  // (past the end: a valid address whose data is never used)
        dma_b(s3nn, min(s + 2, nsl - 1) * WS_BK);
        const __bf16* ab_ = As + cur * WS_A_STAGE + a_rd;
        const __bf16* bb_ = Bs + s3 * WS_B_STAGE + bc * WS_BK;
        bf16x8 a_[2][3], b_[2][3];
        bf16x2 h_[2], m_[2], l_[2];
        GS_RD(0, 0, 3); GS_RD(0, 1, 3); GS_RD(0, 2, 3);
        const float4 v = *reinterpret_cast<const float4*>(Ar + x4n * GS_ARAW + a_own);       // step s + 1's x: requested two steps ago, landed behind the last step's wait
        __builtin_amdgcn_sched_barrier(0);
        GS_MFMA(a_[0][0], b_[0][0]); GS_MFMA(a_[0][0], b_[0][1]); GS_MFMA(a_[0][1], b_[0][0]);
        split2(v.x, v.y, h_[0], m_[0], l_[0]);
        __builtin_amdgcn_sched_barrier(0);
        GS_MFMA(a_[0][1], b_[0][1]); GS_MFMA(a_[0][0], b_[0][2]); GS_MFMA(a_[0][2], b_[0][0]);
        GS_RD(1, 0, 3); GS_RD(1, 1, 3);
        split2(v.z, v.w, h_[1], m_[1], l_[1]);
        __builtin_amdgcn_sched_barrier(0);
        GS_MFMA(a_[1][0], b_[1][0]); GS_MFMA(a_[1][0], b_[1][1]); GS_MFMA(a_[1][1], b_[1][0]);
        GS_RD(1, 2, 3);
        store_planes(cur ^ 1, h_, m_, l_);
        __builtin_amdgcn_sched_barrier(0);
        GS_MFMA(a_[1][1], b_[1][1]); GS_MFMA(a_[1][0], b_[1][2]); GS_MFMA(a_[1][2], b_[1][0]);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");       // all but this step's four DMAs: the next step's x and W have landed
        __syncthreads();
        cur ^= 1;
        s3 = s3 == 2 ? 0 : s3 + 1; s3nn = s3nn == 2 ? 0 : s3nn + 1;
        x4n = (x4n + 1) & 3; x4nnn = (x4nnn + 1) & 3;
    }
#undef GS_RD
#undef GS_MFMA
    // epilogue; 32x32 C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    const int n = n0 + wn * 32 + l31;
    if (n < A.N) {
        const float sc = A.scale ? A.scale[n] : 1.f;
        const float sh = A.shift ? A.shift[n] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = p0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
            if (row >= M) continue;
            float v = acc[e] * sc + sh;
            if (A.rowbias) v += A.rowbias[(int64_t)(row / A.T) * A.N + n];
            if (A.act == ACT_RELU) v = fmaxf(v, 0.f);
            else if (A.act == ACT_TANH) v = gt_tanh(v);
            if (A.res) v += A.res[(int64_t)row * A.ldo + n];
            A.out[(int64_t)row * A.ldo + n] = v;
        }
    }
}

}  // namespace

hipError_t gt_conv_wino5s_init() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gt_gemm_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, GS_LDS_BYTES);
#define WS_ATTR(MO, X3) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gt_conv_wino5s_kernel<MO, X3>), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES);
    WS_ATTR(4, false) WS_ATTR(2, false) WS_ATTR(4, true) WS_ATTR(2, true)
#undef WS_ATTR
    return e;
}

// mo = 4 / 2: which transform (the caller -- gt_launch_conv_gemm -- has applied the grid-fill rule); planes = a.wino_s4 / a.wino_s
hipError_t gt_launch_conv_wino5s(const ConvGemmArgs& a, int mo, hipStream_t stream) {
    const int nb = (a.N + WS_BN - 1) / WS_BN;
    const int P = a.B * ((a.T + mo - 1) / mo);
    const dim3 grid(8 * (((P + WS_BMP - 1) / WS_BMP + 7) / 8) * nb);
    const __bf16* planes = reinterpret_cast<const __bf16*>(mo == 4 ? a.wino_s4 : a.wino_s);
    if (a.wino_x3) {        // (the knob's reduced form: two planes, three products)
        if (mo == 4) hipLaunchKernelGGL((gt_conv_wino5s_kernel<4, true>), grid, dim3(WT), WS_LDS_BYTES, stream, a, planes, a.wino_npad);
        else hipLaunchKernelGGL((gt_conv_wino5s_kernel<2, true>), grid, dim3(WT), WS_LDS_BYTES, stream, a, planes, a.wino_npad);
        return hipGetLastError();
    }
    if (mo == 4) hipLaunchKernelGGL((gt_conv_wino5s_kernel<4, false>), grid, dim3(WT), WS_LDS_BYTES, stream, a, planes, a.wino_npad);
    else hipLaunchKernelGGL((gt_conv_wino5s_kernel<2, false>), grid, dim3(WT), WS_LDS_BYTES, stream, a, planes, a.wino_npad);
    return hipGetLastError();
}

// plain GEMM (taps == 1, no gather): a.gemm_s = the W planes [3][a.wino_npad][Cin]; Cin a multiple of 32, >= 64
bool gt_gemm_split_applies(const ConvGemmArgs& a) {
    return a.gemm_s && a.taps == 1 && !a.tokens && !a.row_len && !a.pool2 && !a.conv2d && !a.wt_bf16 && a.Cin % WS_BK == 0 && a.Cin >= 2 * WS_BK &&
           (size_t)a.B * a.T * a.Cin * 4 < 0x7FFFFFFFull && a.wino_npad >= (a.N + WS_BN - 1) / WS_BN * WS_BN;
}

hipError_t gt_launch_gemm_split(const ConvGemmArgs& a, hipStream_t stream) {
    const int nb = (a.N + WS_BN - 1) / WS_BN;
    const int M = a.B * a.T;
    const dim3 grid(8 * (((M + WS_BMP - 1) / WS_BMP + 7) / 8) * nb);
    hipLaunchKernelGGL(gt_gemm_split_kernel, grid, dim3(WT), GS_LDS_BYTES, stream, a, reinterpret_cast<const __bf16*>(a.gemm_s), a.wino_npad);
    return hipGetLastError();
}
