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
// [3][128 columns][32 k] bf16 with the four 16-byte chunks of a column XOR-swizzled by (column / 2) % 4 (unpadded, conflict-free both ways).
#include "wino_common.h"

namespace {

constexpr int WS_BMP = 64, WS_BN = 128, WS_BK = 32, WS_LDA = WS_BK + 8;
constexpr int WS_A_STAGE = 3 * WS_BMP * WS_LDA, WS_B_STAGE = 3 * WS_BN * WS_BK;      // bf16 elements
constexpr int WS_LDS_BYTES = 2 * (WS_A_STAGE + WS_B_STAGE) * 2;

// v = hi + mid + lo (bf16 each, round to nearest even); the two remainders are exact in fp32
__device__ __forceinline__ void ws_split(const float v, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)v;
    const float r1 = v - (float)h;
    m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    l = (__bf16)r2;
}

template <int MO>
__global__ __launch_bounds__(WT, 2) void gt_conv_wino5s_kernel(ConvGemmArgs A, const __bf16* __restrict__ Us, const int npad) {
    constexpr int AL = Wino<MO>::ALPHA;
    extern __shared__ __attribute__((aligned(16))) __bf16 ws_lds[];
    __bf16* As = ws_lds;                            // [2][3][64][WS_LDA]
    __bf16* Bs = ws_lds + 2 * WS_A_STAGE;           // [2][3][128][32]
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

    // this thread's piece of a step's B planes: column tid >> 2 of the 128, 16-byte chunk tid & 3 of its 32 k -- the same for every step
    // and plane, so the per-thread offset is one register and (xi, plane, slice) ride in the scalar offset
    const auto rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(Us), 0, (int)((size_t)AL * 3 * npad * A.wino_cin * 2), 0x00020000);
    const int bcol = tid >> 2, bchk = tid & 3;
    const uint32_t vb = (uint32_t)((((n0 + bcol) * A.wino_cin) + bchk * 8) * 2);
    const int b_st = bcol * WS_BK + ((bchk ^ ((bcol >> 1) & 3)) * 8);           // element offset inside a plane of a stage
    const int a_st = (tid >> 3) * WS_LDA + (tid & 7) * 4;
    u32x4 bP[3];
    auto issue_b = [&](const int xi, const int c0) {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs_u, (int)vb, (int)((((size_t)(xi * 3 + p) * npad) * A.wino_cin + c0) * 2), 0);
            __builtin_memcpy(&bP[p], &t, 16);
        }
    };
    auto store_slice = [&](const int st, const float4 v) {
        bf16x4 h, m, l;
        const float ve[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            __bf16 he, me, le;
            ws_split(ve[e], he, me, le);
            h[e] = he; m[e] = me; l[e] = le;
        }
        __bf16* as = As + st * WS_A_STAGE + a_st;
        *reinterpret_cast<bf16x4*>(as) = h;
        *reinterpret_cast<bf16x4*>(as + WS_BMP * WS_LDA) = m;
        *reinterpret_cast<bf16x4*>(as + 2 * WS_BMP * WS_LDA) = l;
        __bf16* bs = Bs + st * WS_B_STAGE + b_st;
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<u32x4*>(bs + p * WS_BN * WS_BK) = bP[p];
    };
    // one step's products: per 16 k the three A and three B plane fragments, then hh, hm, mh, mm, hl, lh into the GEMM's accumulator
    const int a_rd = (wm * 32 + l31) * WS_LDA + kh * 8;
    const int bc = wn * 32 + l31, b_sw = (bc >> 1) & 3;
#define WS_MMA(ACC) do {                                                                                           \
        const __bf16* ab_ = As + cur * WS_A_STAGE + a_rd;                                                          \
        const __bf16* bb_ = Bs + cur * WS_B_STAGE + bc * WS_BK;                                                    \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                         \
            bf16x8 a_[3], b_[3];                                                                                   \
            _Pragma("unroll") for (int p = 0; p < 3; ++p) {                                                        \
                a_[p] = *reinterpret_cast<const bf16x8*>(ab_ + p * WS_BMP * WS_LDA + ks * 16);                     \
                b_[p] = *reinterpret_cast<const bf16x8*>(bb_ + p * WS_BN * WS_BK + (((2 * ks + kh) ^ b_sw) * 8));  \
            }                                                                                                      \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], b_[0], ACC, 0, 0, 0);                             \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], b_[1], ACC, 0, 0, 0);                             \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1], b_[0], ACC, 0, 0, 0);                             \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[1], b_[1], ACC, 0, 0, 0);                             \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[0], b_[2], ACC, 0, 0, 0);                             \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_[2], b_[0], ACC, 0, 0, 0);                             \
            if (ks == 0) __builtin_amdgcn_sched_barrier(0);                                                        \
        }                                                                                                          \
    } while (0)
    float4 dE[AL], dO[AL];
#define WS_STEP(XI, DCUR, DNXT, s_)                                                                                \
    {                                                                                                             \
        constexpr int XI1 = (XI + 1) % AL;                                                                         \
        /* (past the last step: a valid address whose data is never used) */                                      \
        issue_b(XI1, min((s_) + (XI + 1 >= AL ? 1 : 0), nsl - 1) * WS_BK);                                         \
        if constexpr (XI == 0) wino_issue_taps<MO>(A, rs_x, voff, first, len, min((s_) + 1, nsl - 1) * WS_BK, (s_) + 1 < nsl, DNXT); \
        WS_MMA(M[XI]);                                                                                            \
        if constexpr (XI + 1 < AL) store_slice(cur ^ 1, wino_xform<MO, XI1>(DCUR));                               \
        else store_slice(cur ^ 1, wino_xform<MO, 0>(DNXT));                                                       \
        __syncthreads();                                                                                          \
        cur ^= 1;                                                                                                 \
    }
#define WS_SLICE(DCUR, DNXT, s_)                                                                                   \
    WS_STEP(0, DCUR, DNXT, s_) WS_STEP(1, DCUR, DNXT, s_) WS_STEP(2, DCUR, DNXT, s_) WS_STEP(3, DCUR, DNXT, s_)   \
    WS_STEP(4, DCUR, DNXT, s_) WS_STEP(5, DCUR, DNXT, s_)                                                          \
    if constexpr (AL == 8) { WS_STEP(6 % AL, DCUR, DNXT, s_) WS_STEP(7 % AL, DCUR, DNXT, s_) }
    wino_issue_taps<MO>(A, rs_x, voff, first, len, 0, true, dE);
    issue_b(0, 0);
    store_slice(0, wino_xform<MO, 0>(dE));
    __syncthreads();
    for (int s = 0; s < nsl; s += 2) {
        WS_SLICE(dE, dO, s)
        WS_SLICE(dO, dE, s + 1)
    }
#undef WS_SLICE
#undef WS_STEP
#undef WS_MMA

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

}  // namespace

hipError_t gt_conv_wino5s_init() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gt_conv_wino5s_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(gt_conv_wino5s_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, WS_LDS_BYTES);
    return e;
}

// mo = 4 / 2: which transform (the caller -- gt_launch_conv_gemm -- has applied the grid-fill rule); planes = a.wino_s4 / a.wino_s
hipError_t gt_launch_conv_wino5s(const ConvGemmArgs& a, int mo, hipStream_t stream) {
    const int nb = (a.N + WS_BN - 1) / WS_BN;
    const int P = a.B * ((a.T + mo - 1) / mo);
    const dim3 grid(8 * (((P + WS_BMP - 1) / WS_BMP + 7) / 8) * nb);
    if (mo == 4) hipLaunchKernelGGL(gt_conv_wino5s_kernel<4>, grid, dim3(WT), WS_LDS_BYTES, stream, a, reinterpret_cast<const __bf16*>(a.wino_s4), a.wino_npad);
    else hipLaunchKernelGGL(gt_conv_wino5s_kernel<2>, grid, dim3(WT), WS_LDS_BYTES, stream, a, reinterpret_cast<const __bf16*>(a.wino_s), a.wino_npad);
    return hipGetLastError();
}
