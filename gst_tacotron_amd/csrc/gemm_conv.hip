// Tall fp32 implicit-GEMM on MFMA: Conv1D(k, stride 1, SAME, no bias) + folded BatchNorm + activation
// (+ residual), also used as a plain Dense over [B*T] rows.
//
//   out[(b,t), n] = act( scale[n] * sum_{tap,c} x[b, t+tap-pad, c] * w[tap, c, n] + shift[n] + rowbias[b,n] ) + res
//
// Replaces: encoder Embedding+3x(Conv1D,BN,ReLU) (reference Modules/Taco2.py:18-38), the postnet
// 5x(Conv1D,BN[,tanh]) + residual (Taco2.py:131-149,230) and the hoisted attention Value projection of
// the encoder memory (Steps.py:123) with the GST half of the concat (GST.py:121-124) folded
// into `rowbias` so the [B,Tv,640] tensor is never materialised.
//
// gfx950 mapping: M = B*T rows on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD), a
// workgroup of 4 waves computes a (WAVES_M*RM*32) x (WAVES_N*RN*32) tile; A (im2col rows, gathered on the
// fly -- the embedding lookup is just another row indirection) and W slices of BK=32 are register-staged
// into LDS k-major so both MFMA operands are conflict-free ds_read_b32 (lanes 0-31 / 32-63 are separate
// bank groups); the next slice's global loads are issued before the current slice's MFMAs.
#include "device_utils.h"
#include "kernels.h"
#include <stdlib.h>

#define BK 32

template <int WAVES_M, int WAVES_N, int RM, int RN, bool C2D = false>
__global__ __launch_bounds__(256) void gt_conv_gemm_kernel(ConvGemmArgs A) {
    constexpr int BM = WAVES_M * RM * 32;
    constexpr int BN = WAVES_N * RN * 32;
    constexpr int LDA = BM + 1;
    constexpr int LDB = BN + 4;
    constexpr int A_F4 = BM * (BK / 4) / 256;      // float4 per thread for the A slice
    constexpr int B_F4 = (BK * BN / 4 + 255) / 256;
    __shared__ float As[BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int Mtot = A.B * A.T;
    const int K = A.taps * A.Cin;
    const int ldw = A.ldw ? A.ldw : A.N;

    // per-thread A rows (constant over the K loop)
    int a_b[A_F4], a_t[A_F4], a_len[A_F4];
    bool a_ok[A_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        const int f = tid + i * 256;
        const int m = m0 + (f >> 3);
        a_ok[i] = m < Mtot;
        const int mm = a_ok[i] ? m : 0;
        a_b[i] = mm / A.T;
        a_t[i] = mm - a_b[i] * A.T;
        a_len[i] = A.row_len ? min(A.T, A.row_len[a_b[i]]) : A.T;      // masked mode: rows >= length read as zero
    }

    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[A_F4], rb[B_F4];
    // 2-D mode: per piece the output position's input origin and batch offset; per slice (load_slice) this thread's tap
    int c2_h0[C2D ? A_F4 : 1], c2_w0[C2D ? A_F4 : 1];
    int64_t c2_b[C2D ? A_F4 : 1];
    int c2_ti = 0, c2_tj = 0, c2_c = 0;
    const auto c2_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.x), 0, C2D ? (int)min((int64_t)0x7FFFFFFF, (int64_t)A.B * A.xb * 4) : 0, 0x00020000);
    if (C2D) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int ho = a_t[i] / A.Wo, wo = a_t[i] - ho * A.Wo;
            c2_h0[i] = ho * A.stride - A.pad_h;
            c2_w0[i] = wo * A.stride - A.pad_w;
            c2_b[i] = (int64_t)a_b[i] * A.xb;
        }
    }

    auto load_slice = [&](int k0) {
        if (C2D) {
            const int kk = k0 + (tid & 7) * 4;
            const int tap = kk / A.Cin;
            c2_c = kk - tap * A.Cin;
            c2_ti = tap / A.kw;
            c2_tj = tap - c2_ti * A.kw;
        }
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int f = tid + i * 256;
            const int kk = k0 + (f & 7) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (C2D) {                                           // 2-D taps, stride, NHWC input (ConvGemmArgs::conv2d)
                // (this thread's k quad is the same for all of its pieces: the tap decomposition is done once per slice, the output
                // position's once per kernel -- c2_h0 / c2_w0; the piece is ONE unconditional buffer load, out of range = zero.  These
                // launches are a few workgroups with 18-36 dependent slices each: the divisions and branches WERE the slice time.)
                const int hi = c2_h0[i] + c2_ti, wi = c2_w0[i] + c2_tj;
                const bool ok = a_ok[i] && kk < K && hi >= 0 && hi < A.H && wi >= 0 && wi < A.W;
                const uint32_t vo = ok ? (uint32_t)((c2_b[i] + ((int64_t)hi * A.W + wi) * A.Cin + c2_c) * 4) : 0x80000000u;
                const auto t4 = __builtin_amdgcn_raw_buffer_load_b128(c2_rs, (int)vo, 0, 0);
                __builtin_memcpy(&v, &t4, 16);
            } else if (a_ok[i] && kk < K) {
                const int tap = kk / A.Cin;
                const int c = kk - tap * A.Cin;
                const int ts = a_t[i] + tap - A.pad_before;
                if (ts >= 0 && ts < a_len[i]) {
                    const int64_t rowi = (int64_t)a_b[i] * A.T + ts;
                    const float* rp = A.tokens ? A.x + (int64_t)A.tokens[rowi] * A.Cin : A.x + rowi * A.Cin;
                    v = *reinterpret_cast<const float4*>(rp + c);
                    if (A.pool2 && ts + 1 < a_len[i]) {         // MaxPool1D(2,1,'same'): padding never wins the max
                        const float4 v2 = *reinterpret_cast<const float4*>(rp + A.Cin + c);
                        v.x = fmaxf(v.x, v2.x); v.y = fmaxf(v.y, v2.y); v.z = fmaxf(v.z, v2.z); v.w = fmaxf(v.w, v2.w);
                    }
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int f = tid + i * 256;
            const int kr = f / (BN / 4);
            const int nq = f - kr * (BN / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int kk = k0 + kr, n = n0 + nq * 4;
            if (kr < BK && kk < K && n < A.N) v = *reinterpret_cast<const float4*>(A.w + (int64_t)kk * ldw + n);
            rb[i] = v;
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int f = tid + i * 256;
            const int row = f >> 3, kq = (f & 7) * 4;
            As[(kq + 0) * LDA + row] = ra[i].x;
            As[(kq + 1) * LDA + row] = ra[i].y;
            As[(kq + 2) * LDA + row] = ra[i].z;
            As[(kq + 3) * LDA + row] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int f = tid + i * 256;
            const int kr = f / (BN / 4);
            const int nq = f - kr * (BN / 4);
            if (kr < BK) *reinterpret_cast<float4*>(&Bs[kr * LDB + nq * 4]) = rb[i];
        }
    };

    const int nslices = (K + BK - 1) / BK;
    load_slice(0);
    const int kh = lane >> 5, l31 = lane & 31;
    for (int s = 0; s < nslices; ++s) {
        __syncthreads();            // previous slice's reads are done
        store_slice();
        __syncthreads();
        if (s + 1 < nslices) load_slice((s + 1) * BK);
#pragma unroll 4
        for (int kp = 0; kp < BK / 2; ++kp) {
            float av[RM], bv[RN];
            const int krow = kp * 2 + kh;
#pragma unroll
            for (int i = 0; i < RM; ++i) av[i] = As[krow * LDA + (wm * RM + i) * 32 + l31];
#pragma unroll
            for (int j = 0; j < RN; ++j) bv[j] = Bs[krow * LDB + (wn * RN + j) * 32 + l31];
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }

    // epilogue; 32x32 C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int n = n0 + (wn * RN + j) * 32 + l31;
        if (n >= A.N) continue;
        const float sc = A.scale ? A.scale[n] : 1.f;
        const float sh = A.shift ? A.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < RM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (wm * RM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                if (m >= Mtot) continue;
                float v = acc[i][j][e] * sc + sh;
                if (A.rowbias) v += A.rowbias[(int64_t)(m / A.T) * A.N + n];
                if (A.act == ACT_RELU) v = fmaxf(v, 0.f);
                else if (A.act == ACT_TANH) v = gt_tanh(v);
                if (A.res) v += A.res[(int64_t)m * A.ldo + n];
                A.out[(int64_t)m * A.ldo + n] = v;
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------- mixed precision
// Same GEMM with bf16 operands on v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  Workgroup tile (RM*64) x 128, 4 waves as
// 2 x 2, BK = 64.  A rows are gathered as in the fp32 kernel (float4 pieces of an im2col row), rounded to bf16 (RNE) and
// written k-contiguous into LDS; the weights are stored transposed ([n][k], k contiguous) at finalize so both operands are
// one ds_read_b128 per lane per MFMA.  Row pitch BKH + 8 halves (144 B): the 16 lanes of a b128 phase hit 16 distinct
// 4-bank groups.
#define BKH 64
#define LDH (BKH + 8)

// 4 consecutive channels of an activation row as fp32, from an fp32 or a bf16 buffer (element index `idx`, a multiple of 4)
__device__ __forceinline__ float4 gt_act_load4(const float* x, const int64_t idx, const bool bf16) {
    if (!bf16) return *reinterpret_cast<const float4*>(x + idx);
    const uint2 h = *reinterpret_cast<const uint2*>(reinterpret_cast<const __bf16*>(x) + idx);
    return make_float4(__builtin_bit_cast(float, h.x << 16), __builtin_bit_cast(float, h.x & 0xFFFF0000u),
                       __builtin_bit_cast(float, h.y << 16), __builtin_bit_cast(float, h.y & 0xFFFF0000u));
}
// Epilogue store of two vertically adjacent elements (rows m0 / m1, column n) of a 32x32 accumulator tile as bf16: adjacent lanes
// exchange so that each owns a PAIR of adjacent columns of ONE of the two rows -- one 4-byte store instead of two 2-byte ones
// (even lanes keep row m0, odd lanes row m1).  All 64 lanes must call.
__device__ __forceinline__ void gt_store_pair_bf16(float* out, const int64_t ldo, const int n, const bool n_ok, const float v0, const float v1,
                                                   const int64_t m0, const int64_t m1, const int64_t row_limit) {
    const bool odd = threadIdx.x & 1;
    const float recv = __shfl_xor(odd ? v0 : v1, 1, 64);
    const float lo = odd ? recv : v0, hi = odd ? v1 : recv;       // columns (n & ~1, n | 1) of row m0 (even lanes) / m1 (odd lanes)
    const int64_t m = odd ? m1 : m0;
    if (m < row_limit && n_ok) {
        bf16x2 h;
        h[0] = (__bf16)lo; h[1] = (__bf16)hi;
        *reinterpret_cast<bf16x2*>(reinterpret_cast<__bf16*>(out) + m * ldo + (n & ~1)) = h;
    }
}

template <int RM, bool XB = false, bool OB = false>
__global__ __launch_bounds__(256) void gt_conv_gemm_bf16_kernel(ConvGemmArgs A) {
    constexpr int BM = 2 * RM * 32, BN = 128, RN = 2;
    constexpr int A_F4 = BM * (BKH / 4) / 256;          // float4 (4 k of one row) per thread per slice
    constexpr int B_U4 = BN * (BKH / 8) / 256;          // 16-byte pieces (8 k of one column) per thread per slice
    __shared__ __attribute__((aligned(16))) __bf16 As[BM * LDH];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[BN * LDH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int Mtot = A.B * A.T;
    const int K = A.taps * A.Cin;

    int a_b[A_F4], a_t[A_F4], a_len[A_F4];
    bool a_ok[A_F4];
#pragma unroll
    for (int i = 0; i < A_F4; ++i) {
        const int f = tid + i * 256;
        const int m = m0 + (f >> 4);                    // 16 float4 per row
        a_ok[i] = m < Mtot;
        const int mm = a_ok[i] ? m : 0;
        a_b[i] = mm / A.T;
        a_t[i] = mm - a_b[i] * A.T;
        a_len[i] = A.row_len ? min(A.T, A.row_len[a_b[i]]) : A.T;
    }
    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[A_F4];
    u32x4 rb[B_U4];       // (ext_vector_type, NOT HIP's uint4 struct: an array of those is not promoted to registers here -- it went through scratch)
    const __bf16* wt = reinterpret_cast<const __bf16*>(A.wt_bf16);
    // The gather: every piece is ONE unconditional buffer load whose offset is out of range (-> zero) where the im2col element does not
    // exist (SAME padding, masked rows, the M / K tails); with a token gather the (clamped) token is an unconditional load in front of it.
    // (As loads under `if (valid)` each piece was waited for at the end of its branch -- the bf16 form converts right there.)  Only the
    // pooled form (MaxPool1D fused into the gather: the vocoder's CBHG) keeps the conditional loads.
    constexpr int AE = XB ? 2 : 4;
    const bool buf_ok = !A.pool2 && (A.tokens != nullptr || (size_t)Mtot * A.Cin * AE < 0x7FFFFFFFull);      // (uniform)
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.x), 0, A.tokens ? 0x7FFFFFFF : (int)((size_t)Mtot * A.Cin * AE), 0x00020000);
    auto load_slice = [&](int k0) {
        if (buf_ok) {
#pragma unroll
            for (int i = 0; i < A_F4; ++i) {
                const int f = tid + i * 256;
                const int kk = k0 + (f & 15) * 4;
                const int tap = kk / A.Cin;
                const int c = kk - tap * A.Cin;
                const int ts = a_t[i] + tap - A.pad_before;
                const bool ok = a_ok[i] && kk < K && ts >= 0 && ts < a_len[i];
                const int64_t rowi = ok ? (int64_t)a_b[i] * A.T + ts : 0;
                const int64_t ri = A.tokens ? (int64_t)A.tokens[rowi] : rowi;
                const uint32_t vo = ok ? (uint32_t)((ri * A.Cin + c) * AE) : 0x80000000u;
                if constexpr (XB) {
                    const auto t = __builtin_amdgcn_raw_buffer_load_b64(rs_x, (int)vo, 0, 0);
                    uint2 h;
                    __builtin_memcpy(&h, &t, 8);
                    ra[i] = make_float4(__builtin_bit_cast(float, h.x << 16), __builtin_bit_cast(float, h.x & 0xFFFF0000u),
                                        __builtin_bit_cast(float, h.y << 16), __builtin_bit_cast(float, h.y & 0xFFFF0000u));
                } else {
                    const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)vo, 0, 0);
                    __builtin_memcpy(&ra[i], &t, 16);
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int f = tid + i * 256;
            const int kk = k0 + (f & 15) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_ok[i] && kk < K) {
                const int tap = kk / A.Cin;
                const int c = kk - tap * A.Cin;
                const int ts = a_t[i] + tap - A.pad_before;
                if (ts >= 0 && ts < a_len[i]) {
                    const int64_t rowi = (int64_t)a_b[i] * A.T + ts;
                    const int64_t ri = A.tokens ? (int64_t)A.tokens[rowi] * A.Cin : rowi * A.Cin;
                    v = gt_act_load4(A.x, ri + c, XB);
                    if (A.pool2 && ts + 1 < a_len[i]) {
                        const float4 v2 = gt_act_load4(A.x, ri + A.Cin + c, XB);
                        v.x = fmaxf(v.x, v2.x); v.y = fmaxf(v.y, v2.y); v.z = fmaxf(v.z, v2.z); v.w = fmaxf(v.w, v2.w);
                    }
                }
            }
            ra[i] = v;
        }
        }
#pragma unroll
        for (int i = 0; i < B_U4; ++i) {
            const int f = tid + i * 256;
            const int n = f >> 3, kq = f & 7;           // 8 pieces of 8 k per column
            rb[i] = *reinterpret_cast<const u32x4*>(wt + (size_t)(n0 + n) * A.ldk + k0 + kq * 8);   // padded: always in range
        }
    };
    auto store_slice = [&]() {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int f = tid + i * 256;
            const int row = f >> 4, kq = (f & 15) * 4;
            bf16x4 h;
            h[0] = (__bf16)ra[i].x; h[1] = (__bf16)ra[i].y; h[2] = (__bf16)ra[i].z; h[3] = (__bf16)ra[i].w;
            *reinterpret_cast<bf16x4*>(&As[row * LDH + kq]) = h;
        }
#pragma unroll
        for (int i = 0; i < B_U4; ++i) {
            const int f = tid + i * 256;
            const int n = f >> 3, kq = f & 7;
            *reinterpret_cast<u32x4*>(&Bs[n * LDH + kq * 8]) = rb[i];
        }
    };

    const int nslices = (K + BKH - 1) / BKH;
    load_slice(0);
    const int kh = lane >> 5, l31 = lane & 31;
    for (int s = 0; s < nslices; ++s) {
        __syncthreads();
        store_slice();
        __syncthreads();
        if (s + 1 < nslices) load_slice((s + 1) * BKH);
#pragma unroll
        for (int ks = 0; ks < BKH / 16; ++ks) {
            bf16x8 av[RM], bv[RN];
#pragma unroll
            for (int i = 0; i < RM; ++i)
                av[i] = *reinterpret_cast<const bf16x8*>(&As[((wm * RM + i) * 32 + l31) * LDH + ks * 16 + kh * 8]);
#pragma unroll
            for (int j = 0; j < RN; ++j)
                bv[j] = *reinterpret_cast<const bf16x8*>(&Bs[((wn * RN + j) * 32 + l31) * LDH + ks * 16 + kh * 8]);
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }

#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int n = n0 + (wn * RN + j) * 32 + l31;
        const bool n_ok = n < A.N;
        const int nc = n_ok ? n : 0;                    // (columns beyond N: computed on column 0's parameters, never stored)
        const float sc = A.scale ? A.scale[nc] : 1.f;
        const float sh = A.shift ? A.shift[nc] : 0.f;
#pragma unroll
        for (int i = 0; i < RM; ++i) {
            auto row_of = [&](const int e) { return (int64_t)(m0 + (wm * RM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh); };
            auto value = [&](const int e) {
                const int64_t m = min(row_of(e), (int64_t)Mtot - 1);
                float x = acc[i][j][e] * sc + sh;
                if (A.rowbias) x += A.rowbias[(m / A.T) * A.N + nc];
                if (A.act == ACT_RELU) x = fmaxf(x, 0.f);
                else if (A.act == ACT_TANH) x = gt_tanh(x);
                if (A.res) x += A.res[m * A.ldo + nc];
                return x;
            };
#pragma unroll
            for (int e = 0; e < 16; e += 2) {
                const float v0 = value(e), v1 = value(e + 1);
                if (OB) {
                    gt_store_pair_bf16(A.out, A.ldo, n, n_ok, v0, v1, row_of(e), row_of(e + 1), (int64_t)Mtot);
                } else {
                    if (n_ok && row_of(e) < Mtot) A.out[row_of(e) * A.ldo + n] = v0;
                    if (n_ok && row_of(e + 1) < Mtot) A.out[row_of(e + 1) * A.ldo + n] = v1;
                }
            }
        }
    }
}


// ------------------------------------------------------------------------------------- five-tap Conv1D on bf16 MFMA
// The implicit GEMM above treats K = taps x Cin as one axis, so every input row is gathered again for each of the five taps
// (and for each 128-column block): at 64 utterances x 1000 frames a 512 -> 512 postnet layer pulled 2.6 GB through the CUs'
// load pipes and ran at 252 TF, a tenth of the bf16 MFMA rate.  Here a workgroup owns 256 output frames of ONE utterance x
// 256 (or 128) output channels, stages the 260 input rows a 64-channel slice needs ONCE (fp32 -> bf16 on their way into LDS)
// and runs the five taps from it -- tap j of output frame i reads staged row i + j -- so only the 32 KB weight slab of a
// (tap, slice) step moves per 2048 MFMA-cycles.  Both LDS operands are double-buffered: one barrier per step, the next
// step's weights and the next slice's rows are in flight under the MFMAs.  Same operand roundings and the same fp32
// accumulation as gt_conv_gemm_bf16_kernel (the order of the K terms differs: tap-major inside a slice).
// Weight slabs (round 5): straight from memory into LDS (buffer_load ... lds: no staging registers, no ds_write, no VALU between the
// request and the MFMAs).  A DMA instruction writes its 64 lanes' 16 bytes side by side, so the slab is UNPADDED -- 128 bytes per
// column -- and swizzled instead: 16-byte chunk c of column r sits at chunk position c ^ (r % 8), which the DMA gets for free (a
// lane chooses what it FETCHES: lane l of an instruction fills position l % 8 of column r0 + l / 8 with chunk (l % 8) ^ (l / 8)) and
// the fragment reads undo (eight consecutive columns at one logical chunk hit eight different positions = all 32 banks).
#ifndef GT_C5_DMA
#define GT_C5_DMA 1
#endif
constexpr int C5_BM = 256, C5_BK = 64, C5_LD = C5_BK + 8, C5_AR = C5_BM + 4;
constexpr int C5_LDB = GT_C5_DMA ? C5_BK : C5_LD;         // elements per column of the weight slab in LDS
template <int RN>       // 32-column tiles per wave: the workgroup tile is 256 frames x (2 * RN * 32) channels
constexpr int c5_lds_bytes() { return 2 * (C5_AR * C5_LD + 2 * RN * 32 * C5_LDB) * 2; }

#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// 16 bytes per lane from a buffer straight into LDS: lane l's data lands at lds_addr + 16 l (lds_addr wave-uniform).  Inline asm, not
// __builtin_amdgcn_raw_ptr_buffer_load_lds: the compiler orders every later LDS read of the kernel behind a DMA it knows of (a
// vmcnt(0) in front of each).  The caller waits (s_waitcnt vmcnt(0)) and synchronises before the data is read.  (The compiler's own
// counted waits stay safe: loads complete in order, an unknown one behind a known one only makes its wait longer.)
__device__ __forceinline__ void gt_lds_dma16(__amdgpu_buffer_rsrc_t rs, const uint32_t lds_addr, const uint32_t voff, const uint32_t soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
#pragma clang diagnostic pop

// XB / OB: the input / output activations are stored as bf16 (compile-time: as run-time flags both forms' registers were live at once)
template <int RN, bool XB = false, bool OB = false>
__global__ __launch_bounds__(512) void gt_conv5_bf16_kernel(ConvGemmArgs A) {
    constexpr int BN = 2 * RN * 32, RM = 2;
    constexpr int A_F4 = (C5_AR * (C5_BK / 4) + 511) / 512;      // float4 pieces (4 channels of one row) per thread per slice
    constexpr int B_U4 = BN * (C5_BK / 8) / 512;                 // 16-byte pieces (8 k of one column) per thread per step
    extern __shared__ __attribute__((aligned(16))) __bf16 c5_lds[];
    __bf16* As = c5_lds;                                         // [2][C5_AR][C5_LD]
    __bf16* Bs = c5_lds + 2 * C5_AR * C5_LD;                     // [2][BN][C5_LDB]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;                     // 4 x 2 waves: 64 frames x (RN * 32) channels each
    const int tiles_t = (A.T + C5_BM - 1) / C5_BM;
    // 1-D grid: the column blocks of one frame tile sit 8 block indices apart = on ONE XCD under the round-robin block -> XCD deal, at
    // the same time (speed only, never correctness): the second read of the tile's input rows is an L2 hit
    const int ncb = (A.N + BN - 1) / BN;
    const int bx = ((int)blockIdx.x / (8 * ncb)) * 8 + ((int)blockIdx.x & 7), by = ((int)blockIdx.x >> 3) % ncb;
    if (bx >= A.B * tiles_t) return;
    const int b = bx / tiles_t, t0 = (bx % tiles_t) * C5_BM;
    const int n0 = by * BN;
    const int len = A.row_len ? min(A.T, A.row_len[b]) : A.T;    // masked mode: input rows >= len read as zero
    const __bf16* wt = reinterpret_cast<const __bf16*>(A.wt_bf16);

    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // (the next slice's 260 rows are staged in two halves -- requested at taps 0 / 2, stored at taps 1 / 3 -- so that only five
    // pieces of them are ever live beside the 128 accumulator registers: with all nine the weight staging registers went
    // through scratch, and a scratch store waits for the loads it saves)
    // Every row piece is an UNCONDITIONAL buffer load: a row outside the utterance (SAME padding, masked mode, the tile's tail) gets an
    // out-of-range offset and reads as zero.  As loads under `if (row in range)` each one was waited for on the spot (vmcnt(0) at the
    // end of its branch: the bf16 form converts right there) -- five dependent memory latencies at taps 0 and 2 of every slice, more
    // than the slice's MFMA time (round 4: 205 -> see DESIGN 3.4 per 512 -> 512 layer).  bf16 input: the 8 bytes go to LDS as they are.
    constexpr int A_H0 = (A_F4 + 1) / 2;
    constexpr int AE = XB ? 2 : 4;                               // bytes per input element
    // (the resource starts pad_before rows BEFORE the tensor, so that the offset of frame -pad_before of utterance 0 is 0 and no
    // per-thread base is negative; those rows are never requested: their pieces get the out-of-range offset)
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(A.x)) - (size_t)A.pad_before * A.Cin * AE, 0,
                                                        (int)(((size_t)A.B * A.T + A.pad_before) * A.Cin * AE), 0x00020000);
    // this thread's pieces: rows (tid >> 4) + 32 j, channel quad tid & 15; the row part of the address is a per-thread base + a scalar
    const int row0 = tid >> 4, tt0 = t0 - A.pad_before + row0;
    const uint32_t abase = (uint32_t)((((int64_t)b * A.T + t0 + row0) * A.Cin + (tid & 15) * 4) * AE);
    float4 ra[XB ? 1 : A_H0];
    uint2 rh[XB ? A_H0 : 1];
    u32x4 rb[GT_C5_DMA ? 1 : B_U4];
    // weight slab DMA: BN / 8 instructions of 8 columns x 128 bytes per step, dealt to the 8 waves; lane l: column l / 8 of the eight,
    // logical chunk (l % 8) ^ (l / 8) -> position l % 8
    const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(wt), 0, 0x7FFFF000, 0x00020000);
    const uint32_t lds_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)Bs;
    // (round 6: the swizzle is (column / 2) % 8, not column % 8 -- a ds_read_b128 is served in four groups of 16 NON-contiguous lanes
    // ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32) over 64 banks: with column % 8 every fragment read of the slab was 2-way
    // conflicted, which is what "bound by its fragment reads" (round 5) really was.  A group of 8 columns starts at a multiple of 8, so
    // its swizzles are ((r0 / 2) % 8) + (lane / 16): two per-thread offsets, by the parity of the instruction's column group.)
    const uint32_t wvoff0 = (uint32_t)(((lane >> 3) * A.ldk + (((lane & 7) ^ (lane >> 4)) * 8)) * 2);
    const uint32_t wvoff1 = (uint32_t)(((lane >> 3) * A.ldk + (((lane & 7) ^ (4 + (lane >> 4))) * 8)) * 2);
    auto dma_b = [&](const int buf, const int tap, const int k0) {
#pragma unroll
        for (int i = 0; i < BN / 64; ++i) {
            const int g8 = __builtin_amdgcn_readfirstlane(wave) * (BN / 64) + i, r0 = g8 * 8;
            gt_lds_dma16(rs_w, lds_b + (uint32_t)((buf * BN + r0) * C5_LDB * 2), (g8 & 1) ? wvoff1 : wvoff0,
                         (uint32_t)((((size_t)(n0 + r0)) * A.ldk + tap * A.Cin + k0) * 2));
        }
    };
    auto load_a = [&](const int k0, const int half) {
#pragma unroll
        for (int i = 0; i < A_H0; ++i) {
            const int j = half * A_H0 + i;
            const int row = row0 + 32 * j, t = tt0 + 32 * j;
            // (Cin not a multiple of the 64-channel slice -- the postnet's 80 -> 512 layer: channels past Cin read as zero, so the
            // weights they meet, the next tap's or the K padding, do not matter)
            const uint32_t vo = (row < C5_AR && t >= 0 && t < len && k0 + (tid & 15) * 4 < A.Cin) ? abase : 0x80000000u;
            const int so = (32 * j * A.Cin + k0) * AE;
            if constexpr (XB) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b64(rs_x, (int)vo, so, 0);
                __builtin_memcpy(&rh[i], &v, 8);
            } else {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)vo, so, 0);
                __builtin_memcpy(&ra[i], &v, 16);
            }
        }
    };
    auto store_a = [&](const int buf, const int half) {
#pragma unroll
        for (int i = 0; i < A_H0; ++i) {
            const int row = row0 + 32 * (half * A_H0 + i);
            if (row < C5_AR) {
                if constexpr (XB) {
                    *reinterpret_cast<uint2*>(&As[(buf * C5_AR + row) * C5_LD + (tid & 15) * 4]) = rh[i];
                } else {
                    bf16x4 h;
                    h[0] = (__bf16)ra[i].x; h[1] = (__bf16)ra[i].y; h[2] = (__bf16)ra[i].z; h[3] = (__bf16)ra[i].w;
                    *reinterpret_cast<bf16x4*>(&As[(buf * C5_AR + row) * C5_LD + (tid & 15) * 4]) = h;
                }
            }
        }
    };
    auto load_b = [&](const int tap, const int k0) {
        if (GT_C5_DMA) return;
#pragma unroll
        for (int i = 0; i < B_U4; ++i) {
            const int f = tid + i * 512;
            rb[i] = *reinterpret_cast<const u32x4*>(wt + (size_t)(n0 + (f >> 3)) * A.ldk + tap * A.Cin + k0 + (f & 7) * 8);   // padded rows: in range
        }
    };
    auto store_b = [&](const int buf) {
        if (GT_C5_DMA) return;
#pragma unroll
        for (int i = 0; i < B_U4; ++i) {
            const int f = tid + i * 512;
            *reinterpret_cast<u32x4*>(&Bs[(buf * BN + (f >> 3)) * C5_LD + (f & 7) * 8]) = rb[i];
        }
    };

#ifdef C5_NO_LOOP
    const int nslices = 0, nsteps = 0;       // (tools/conv5_bench.hip: the kernel's fixed part -- prologue + epilogue -- alone)
#else
    const int nslices = (A.Cin + C5_BK - 1) / C5_BK, nsteps = nslices * 5;
#endif
    load_a(0, 0);
    store_a(0, 0);
    load_a(0, 1);
    store_a(0, 1);
    load_b(0, 0);
    store_b(0);
    if (GT_C5_DMA) { dma_b(0, 0, 0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __syncthreads();
    const int kh = lane >> 5, l31 = lane & 31;
    for (int s = 0; s < nsteps; ++s) {
        const int sl = s / 5, tap = s - sl * 5;
        // requests for the NEXT step's weight slab (the last step re-requests its own: no load under a branch, so the staging
        // registers stay registers and the waits stay counted), and at tap 0 for the next slice's input rows (stored at tap 1:
        // the other A buffer was last read in the previous slice)
        const int sn = min(s + 1, nsteps - 1);
        load_b(sn % 5, (sn / 5) * C5_BK);
#ifndef C5_NO_A
        if (tap == 0 || tap == 2) load_a(min(sl + 1, nslices - 1) * C5_BK, tap >> 1);
#endif
        // (DMA: into the other slab, last read in step s - 1 -- every wave is past that step's barrier.  BEHIND the row requests since round 6:
        // before the compiler re-uses the rows' registers it waits for the loads it knows with vmcnt(4) .. vmcnt(0) -- which, with this step's
        // four DMAs already queued, waited for THEM: a DMA's whole latency exposed at taps 0 and 2 of every slice.  Also measured: the requests
        // behind the step's first / second / last eight MFMAs instead of in front of its fragment reads -- no difference, tools/conv5_bench.hip)
#ifndef C5_NO_DMA
        if (GT_C5_DMA) dma_b((s + 1) & 1, sn % 5, (sn / 5) * C5_BK);
#endif
        const __bf16* Ab = As + ((sl & 1) * C5_AR + wm * 64 + l31 + tap) * C5_LD + kh * 8;
        const __bf16* Bb = Bs + ((s & 1) * BN + wn * RN * 32 + l31) * C5_LDB + (GT_C5_DMA ? 0 : kh * 8);
#pragma unroll
        for (int ks = 0; ks < C5_BK / 16; ++ks) {
            bf16x8 av[RM], bv[RN];
            // (ablation switches of tools/conv5_bench.hip: -DC5_NO_READS_A / _B -- one fragment read per step instead of one per 16 k --,
            // -DC5_NO_MFMA, -DC5_NO_DMA, -DC5_NO_A; the product compiles the plain forms)
#ifdef C5_NO_READS_A
#pragma unroll
            for (int i = 0; i < RM; ++i) av[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * C5_LD);
#else
#pragma unroll
            for (int i = 0; i < RM; ++i) av[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * C5_LD + ks * 16);
#endif
#ifdef C5_NO_READS_B
#pragma unroll
            for (int j = 0; j < RN; ++j) bv[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * C5_LDB);
#else
#pragma unroll
            for (int j = 0; j < RN; ++j)        // (DMA: chunk 2 ks + kh of column .. + l31 sits at position (2 ks + kh) ^ ((l31 / 2) % 8): j * 32 keeps the column's low bits)
                bv[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 32 * C5_LDB + (GT_C5_DMA ? (((2 * ks + kh) ^ ((l31 >> 1) & 7)) * 8) : ks * 16));
#endif
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j)
#ifdef C5_NO_MFMA
                    acc[i][j][ks] += (float)av[i][0] + (float)bv[j][1];
#else
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
#endif
        }
        store_b((s + 1) & 1);           // (after the last step: written, never read)
#ifndef C5_NO_A
        if (tap == 1 || tap == 3) store_a((sl + 1) & 1, tap >> 1);
#endif
        if (GT_C5_DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // (the next slab has landed -- requested a step's MFMAs ago)
        __syncthreads();
    }

    if constexpr (!OB) {
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int n = n0 + (wn * RN + j) * 32 + l31;
            if (n >= A.N) continue;
            const float sc = A.scale ? A.scale[n] : 1.f;
            const float sh = A.shift ? A.shift[n] : 0.f;
#pragma unroll
            for (int i = 0; i < RM; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int t = t0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;
                    if (t >= A.T) continue;
                    const int64_t m = (int64_t)b * A.T + t;
                    float v = acc[i][j][e] * sc + sh;
                    if (A.rowbias) v += A.rowbias[(int64_t)b * A.N + n];
                    if (A.act == ACT_RELU) v = fmaxf(v, 0.f);
                    else if (A.act == ACT_TANH) v = gt_tanh(v);
                    if (A.res) v += A.res[m * A.ldo + n];
                    A.out[m * A.ldo + n] = v;
                }
            }
        }
    } else {
        // bf16 output (an intermediate of a bf16 chain: no residual here): pairs of rows, adjacent lanes exchange (gt_store_pair_bf16)
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int n = n0 + (wn * RN + j) * 32 + l31;
            const bool n_ok = n < A.N;
            const int nc = n_ok ? n : 0;
            const float sc = A.scale ? A.scale[nc] : 1.f;
            const float sh = (A.shift ? A.shift[nc] : 0.f) + (A.rowbias ? A.rowbias[(int64_t)b * A.N + nc] : 0.f);
            const int64_t m_end = (int64_t)(b + 1) * A.T;
#pragma unroll
            for (int i = 0; i < RM; ++i) {
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    const int64_t m0r = (int64_t)b * A.T + t0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kh;     // (frames past T land beyond m_end)
                    float v0 = acc[i][j][e] * sc + sh, v1 = acc[i][j][e + 1] * sc + sh;
                    if (A.act == ACT_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
                    else if (A.act == ACT_TANH) { v0 = gt_tanh(v0); v1 = gt_tanh(v1); }
                    gt_store_pair_bf16(A.out, A.ldo, n, n_ok, v0, v1, m0r, m0r + 1, m_end);
                }
            }
        }
    }
}

static bool gt_conv5_bf16_applies(const ConvGemmArgs& a) {
    return a.wt_bf16 && a.taps == 5 && a.Cin % 4 == 0 && (a.Cin % C5_BK == 0 || a.ldk >= 4 * a.Cin + (a.Cin + C5_BK - 1) / C5_BK * C5_BK) && !a.tokens && !a.pool2 && !a.conv2d && a.N >= 64 && a.T >= 64 &&
           ((size_t)a.B * a.T + a.pad_before) * a.Cin * 4 < 0x7FFFFFFFull;         // (the input as one buffer resource)
}

hipError_t gt_conv5_bf16_init() {
    hipError_t e = hipSuccess;
#define GT_C5_ATTR(RN, XB, OB)                                                                                                                    \
    if (e == hipSuccess)                                                                                                                          \
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(gt_conv5_bf16_kernel<RN, XB, OB>), hipFuncAttributeMaxDynamicSharedMemorySize, c5_lds_bytes<RN>());
    GT_C5_ATTR(4, false, false) GT_C5_ATTR(4, true, false) GT_C5_ATTR(4, false, true) GT_C5_ATTR(4, true, true)
    GT_C5_ATTR(2, false, false) GT_C5_ATTR(2, true, false) GT_C5_ATTR(2, false, true) GT_C5_ATTR(2, true, true)
#undef GT_C5_ATTR
    return e;
}

// ---------------------------------------------------------------------------------------------------- Winograd F(2, 5)
// Conv1D(k = 5, stride 1, SAME) as the minimal-filtering algorithm F(2, 5): two outputs from six inputs with 6
// multiplications per (input channel, output channel) instead of 10, i.e. 0.6x the MFMAs of the implicit GEMM above for the
// same result (fp32 throughout; Cook-Toom points 0, +-1, +-1/2, infinity).  For a tile p (outputs t = 2p, 2p + 1; inputs
// d_i = x[2p - 2 + i], i = 0..5):
//     V_xi = sum_i BT[xi][i] d_i          (input transform, on the fly in the A gather: 3-4 rows per element)
//     M_xi = V_xi . U_xi                  (six GEMMs over the input channels, U_xi = sum_k G[xi][k] w[k], formed in float64
//                                          at finalize)
//     y_0 = sum_xi M_xi (xi < 5),   y_1 = M_1 - M_2 + (M_3 - M_4) / 2 + M_5
// One workgroup owns 64 tiles (128 output rows) x 128 columns; each of the six GEMMs has its own 32 x 32 accumulator tile per wave,
// advanced one 32-channel slice at a time in turn (the slice's input rows are fetched once for all six), and the output
// transform is the epilogue's (V never touches memory, no cross-workgroup reduction).
// Measured error on the postnet's shapes: 2-3.6e-6 max-abs at |y| ~ 3 (the direct fp32 sum: 1.1e-6) --
// tests/test_gpu_configs.py::test_winograd_postnet_matches_the_oracle_and_the_implicit_gemm prints all three variants' errors.
// Applies to: taps == 5, pad_before == 2, Cin % 32 == 0, no pooling, no 2-D mode.
// Transform rows at compile time.  MO = outputs per tile: 2 -> F(2,5), points 0, +-1, +-1/2, inf (6 GEMMs per 2 outputs, 0.6x
// the multiplications of the direct sum); 4 -> F(4,5), points 0, +-1, +-1/2, +-2, inf (8 GEMMs per 4 outputs, 0.4x).
#include "wino_common.h"

#ifdef GT_WINO_STAMPS          // tools/wino_bench.hip: one cycle stamp per step of workgroup 0, wave 0
__device__ unsigned long long gt_wino_stamp[1024];
#endif

template <int MO>
__global__ __launch_bounds__(WT, 2) void gt_conv_wino5_kernel(ConvGemmArgs A, const float* __restrict__ U) {
    constexpr int AL = Wino<MO>::ALPHA;
    constexpr int BMP = 64, BN = 128, LDA = BMP + 1, LDB = BN + 4;
    // two LDS stages: slice g + 1 is written while slice g is read, ONE barrier per slice (50 KB per workgroup)
    __shared__ float As[2][BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int Pu = (A.T + MO - 1) / MO;                  // tiles per utterance
    const int Ptot = A.B * Pu;
    // Workgroup -> (row block, column block): the column blocks of one row block read the same input rows, and an XCD's L2
    // (4 MB) holds the rows of about eight row blocks -- so they must run on the SAME XCD at the SAME time.  Workgroups are
    // dealt to the XCDs round-robin by linear id: XCD x's i-th workgroup takes column block i % ncb of row block
    // (i / ncb) * 8 + x.  (Pass-major kernel: 1.9 GB fetched from the memory side per 512 -> 512 layer for 73 MB of operands,
    // L2 hit rate 0.39; with this mapping 0.55 GB and 0.80.  The launch is 1-D, rounded up to whole groups of 8 row blocks.)
    const int ncb = (A.N + BN - 1) / BN;
    const int wi = blockIdx.x >> 3;
    const int rb = (wi / ncb) * 8 + (blockIdx.x & 7), cb = wi % ncb;
    if (rb * BMP >= Ptot) return;
    const int p0 = rb * BMP, n0 = cb * BN;
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
        // (the first rows of utterance 0 give a negative row index: those taps are out of range anyway, the wrapped offset is unused)
        voff = (uint32_t)(((int64_t)b * A.T + first) * A.Cin + (tid & 7) * 4) * 4u;
    }
    // one accumulator tile per transform-domain GEMM (8 x 16 registers for F(4,5)); the output transform is the epilogue's
    f32x16 M[AL];
#pragma unroll
    for (int xi = 0; xi < AL; ++xi)
#pragma unroll
        for (int e = 0; e < 16; ++e) M[xi][e] = 0.f;
    const int kh = lane >> 5, l31 = lane & 31;
    // wino_cin = Cin rounded up to TWO slice widths: U holds zero rows for the padding channels, whose x operand is whatever
    // follows in memory (the next row's first channels, or zero past the tensor's end: the descriptor covers exactly B*T*Cin)
    const int nsl = A.wino_cin / BK;                      // even, >= 4 (gt_conv_wino5_applies)
    int cur = 0;                                          // LDS stage the current step's operands sit in

    auto store_slice = [&](int st, const float4 ra, const float4 rb0, const float4 rb1) {
        const int row = tid >> 3, kq = (tid & 7) * 4;
        As[st][(kq + 0) * LDA + row] = ra.x;
        As[st][(kq + 1) * LDA + row] = ra.y;
        As[st][(kq + 2) * LDA + row] = ra.z;
        As[st][(kq + 3) * LDA + row] = ra.w;
        *reinterpret_cast<float4*>(&Bs[st][(tid >> 5) * LDB + (tid & 31) * 4]) = rb0;
        *reinterpret_cast<float4*>(&Bs[st][(16 + (tid >> 5)) * LDB + (tid & 31) * 4]) = rb1;
    };
    // A step's 32 operand words are requested from LDS before its first MFMA (fully unrolled: counted lgkmcnt waits), so that
    // the 16 dependent MFMAs run back to back instead of paying an LDS latency every other one.
#define WINO_MMA(ACC) do {                                                                                         \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_) {                                                         \
            float av_[BK / 4], bv_[BK / 4];                                                                        \
            _Pragma("unroll") for (int kp = 0; kp < BK / 4; ++kp) {                                                \
                const int krow = (h_ * (BK / 4) + kp) * 2 + kh;                                                    \
                av_[kp] = As[cur][krow * LDA + wm * 32 + l31];                                                     \
                bv_[kp] = Bs[cur][krow * LDB + wn * 32 + l31];                                                     \
            }                                                                                                      \
            _Pragma("unroll") for (int kp = 0; kp < BK / 4; ++kp) ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(av_[kp], bv_[kp], ACC, 0, 0, 0); \
        }                                                                                                          \
    } while (0)
    // The work is ONE stream of steps g = s * ALPHA + XI, SLICE-major: the ALPHA transform-domain GEMMs of a 32-channel slice
    // follow each other, each on its own accumulator, so that the slice's raw input rows are fetched ONCE (into dE / dO, even /
    // odd slices, seven steps ahead) instead of once per GEMM -- the pass-major order moved 61 KB per step and CU through the
    // L1 (1 630 cycles per step with the MFMAs taken out, against 2 050 of MFMA), this one 24 KB.  Per step: request the B
    // slice of the next step (one register set, bP0 / bP1), at XI = 0 the next slice's taps; the MFMAs on the current LDS
    // stage; transform + store the operands of step g + 1 into the other stage; barrier.  Nothing pins the order inside a
    // step: left to the scheduler, the transform's VALU work and the LDS writes land between the dependent MFMAs.  Slices are
    // unrolled by two so that the tap sets are named statically; every load is unconditional, so every wait is a counted one.
    // (Tried and dropped: the two waves that share a SIMD doing MFMAs / transform + store in opposite order -- one wave's chain
    // of dependent 32x32x2 MFMAs alone issues every ~135 cycles, two waves' interleaved chains every ~60 = the pipe's rate; two
    // 4-wave workgroups per CU instead of one 8-wave one; three LDS stages with the next step's operands read ahead of the
    // barrier -- none faster while the L1 traffic was the limit.)
    float4 dE[AL], dO[AL], bP0, bP1;
    const auto rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, (int)((size_t)AL * A.wino_cin * A.N * 4), 0x00020000);
    const uint32_t vb0 = (uint32_t)(((tid >> 5) * A.N + min(n0 + (tid & 31) * 4, A.N - 4)) * 4);   // (columns past N are never stored)
    const uint32_t vb1 = vb0 + (uint32_t)(16 * A.N * 4);
    // (ablation switches of tools/wino_bench.hip; the product compiles the plain forms)
#ifdef GT_WINO_NO_MFMA
#define WINO_ABL_MMA(ACC) do { ACC[0] += As[cur][tid & 31] + Bs[cur][tid & 31]; } while (0)
#else
#define WINO_ABL_MMA(ACC) WINO_MMA(ACC)
#endif
#ifdef GT_WINO_NO_XSTORE
    auto abl_sink = [&](int, const float4 ra_, const float4 b0_, const float4 b1_) { M[0][1] += ra_.x + ra_.y + ra_.z + ra_.w + b0_.x + b1_.y; };
#define WINO_ABL_STORE(...) abl_sink(__VA_ARGS__)
#else
#define WINO_ABL_STORE(...) store_slice(__VA_ARGS__)
#endif
#define WINO_STEP(XI, DCUR, DNXT, s_)                                                                              \
    {                                                                                                             \
        constexpr int XI1 = (XI + 1) % AL;                                                                         \
        /* (past the last step: a valid address whose data is never used) */                                      \
        wino_issue_b(A, rs_u, vb0, vb1, XI1, min((s_) + (XI + 1 >= AL ? 1 : 0), nsl - 1) * BK, bP0, bP1);          \
        if constexpr (XI == 0) wino_issue_taps<MO>(A, rs_x, voff, first, len, min((s_) + 1, nsl - 1) * BK, (s_) + 1 < nsl, DNXT); \
        WINO_ABL_MMA(M[XI]);                                                                                      \
        if constexpr (XI + 1 < AL) WINO_ABL_STORE(cur ^ 1, wino_xform<MO, XI1>(DCUR), bP0, bP1);                  \
        else WINO_ABL_STORE(cur ^ 1, wino_xform<MO, 0>(DNXT), bP0, bP1);                                          \
        __syncthreads();                                                                                          \
        cur ^= 1;                                                                                                 \
        WINO_STAMP_STEP();                                                                                        \
    }
#define WINO_SLICE(DCUR, DNXT, s_)                                                                                 \
    WINO_STEP(0, DCUR, DNXT, s_) WINO_STEP(1, DCUR, DNXT, s_) WINO_STEP(2, DCUR, DNXT, s_) WINO_STEP(3, DCUR, DNXT, s_) \
    WINO_STEP(4, DCUR, DNXT, s_) WINO_STEP(5, DCUR, DNXT, s_)                                                      \
    if constexpr (AL == 8) { WINO_STEP(6 % AL, DCUR, DNXT, s_) WINO_STEP(7 % AL, DCUR, DNXT, s_) }
#ifdef GT_WINO_STAMPS
    const bool stamp_on = blockIdx.x == 0 && tid == 0;
    int nstamp = 0;
#define WINO_STAMP_STEP() do { if (stamp_on && nstamp < 1023) gt_wino_stamp[nstamp++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WINO_STAMP_STEP() do { } while (0)
#endif
    // B(g) lives in set g & 1 (ALPHA is even).  Prologue: taps of slice 0, B of steps 0 and 1, operands of step 0 into stage 0.
    wino_issue_taps<MO>(A, rs_x, voff, first, len, 0, true, dE);
    wino_issue_b(A, rs_u, vb0, vb1, 0, 0, bP0, bP1);
    store_slice(0, wino_xform<MO, 0>(dE), bP0, bP1);
    __syncthreads();
    for (int s = 0; s < nsl; s += 2) {
        WINO_SLICE(dE, dO, s)
        WINO_SLICE(dO, dE, s + 1)
    }
#undef WINO_SLICE
#undef WINO_STEP
#undef WINO_MMA
#undef WINO_ABL_MMA
#undef WINO_ABL_STORE
#undef WINO_STAMP_STEP

    // epilogue; 32x32 C/D layout: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5); tile row -> MO output rows
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

bool gt_conv_wino5_applies(const ConvGemmArgs& a) {
    return a.wino_u && !a.wt_bf16 && !a.conv2d && !a.pool2 && !a.tokens && a.N % 4 == 0 && a.N >= 4 && a.taps == 5 &&
           (size_t)a.B * a.T * a.Cin * 4 < 0x7FFFFFFFull && a.pad_before == 2 && a.Cin % 4 == 0 && a.wino_cin % (2 * BK) == 0 &&
           a.wino_cin >= 4 * BK && a.wino_cin >= a.Cin && (a.ldw == 0 || a.ldw == a.N);
}

hipError_t gt_launch_conv_gemm(const ConvGemmArgs& a, hipStream_t stream) {
    const int M = a.B * a.T;
    if (gt_conv5_bf16_applies(a)) {         // five taps sharing one staged input tile (256 frames of one utterance per workgroup)
        const int tiles_t = (a.T + C5_BM - 1) / C5_BM;
        const int nx8 = (a.B * tiles_t + 7) / 8 * 8;         // (1-D grids: frame tiles rounded up to 8 x column blocks, see the kernel)
        const dim3 g4(nx8 * ((a.N + 255) / 256)), g2(nx8);
#define GT_C5_LAUNCH(XB, OB)                                                                                                       \
        do {                                                                                                                       \
            if (a.N > 128) hipLaunchKernelGGL((gt_conv5_bf16_kernel<4, XB, OB>), g4, dim3(512), c5_lds_bytes<4>(), stream, a);     \
            else hipLaunchKernelGGL((gt_conv5_bf16_kernel<2, XB, OB>), g2, dim3(512), c5_lds_bytes<2>(), stream, a);               \
        } while (0)
        if (a.x_bf16 && a.out_bf16) GT_C5_LAUNCH(true, true);
        else if (a.x_bf16) GT_C5_LAUNCH(true, false);
        else if (a.out_bf16) GT_C5_LAUNCH(false, true);
        else GT_C5_LAUNCH(false, false);
#undef GT_C5_LAUNCH
        return hipGetLastError();
    }
    if (a.wt_bf16) {
        const int nb = (a.N + 127) / 128;
        const bool big = ((M + 127) / 128) * nb >= 256;
        const dim3 gb((M + 127) / 128, nb), gs((M + 63) / 64, nb);
#define GT_CG_LAUNCH(XB, OB)                                                                                              \
        do {                                                                                                              \
            if (big) hipLaunchKernelGGL((gt_conv_gemm_bf16_kernel<2, XB, OB>), gb, dim3(256), 0, stream, a);              \
            else hipLaunchKernelGGL((gt_conv_gemm_bf16_kernel<1, XB, OB>), gs, dim3(256), 0, stream, a);                  \
        } while (0)
        if (a.x_bf16 && a.out_bf16) GT_CG_LAUNCH(true, true);
        else if (a.x_bf16) GT_CG_LAUNCH(true, false);
        else if (a.out_bf16) GT_CG_LAUNCH(false, true);
        else GT_CG_LAUNCH(false, false);
#undef GT_CG_LAUNCH
        return hipGetLastError();
    }
    if (gt_gemm_split_applies(a) && ((M + 63) / 64) * ((a.N + 127) / 128) >= 48) return gt_launch_gemm_split(a, stream);
    if (gt_conv_wino5_applies(a)) {
        // worth it when the grid (nearly) fills the chip: the 4096-row encoder convs would leave half of it idle (one round of
        // 128 Winograd workgroups ~275 us against 136 us for the implicit GEMM); 250 workgroups (the 512 -> 80 layer) do pay.
        // F(4,5) where its (half as large) grid still does, else F(2,5).
        const int min_wgs = a.wino_min_wgs > 0 ? a.wino_min_wgs : 240;
        const int nb = (a.N + 127) / 128;
        const int P4 = a.B * ((a.T + 3) / 4), P2 = a.B * ((a.T + 1) / 2);
        // (1-D grids: 8 XCDs x ceil(row blocks / 8) x column blocks, see the kernel)
        if (a.wino_u4 && ((P4 + 63) / 64) * nb >= min_wgs) {
            if (a.wino_s4) return gt_launch_conv_wino5s(a, 4, stream);
            hipLaunchKernelGGL(gt_conv_wino5_kernel<4>, dim3(8 * (((P4 + 63) / 64 + 7) / 8) * nb), dim3(WT), 0, stream, a, a.wino_u4);
            return hipGetLastError();
        }
        if (((P2 + 63) / 64) * nb >= min_wgs) {
            if (a.wino_s) return gt_launch_conv_wino5s(a, 2, stream);
            hipLaunchKernelGGL(gt_conv_wino5_kernel<2>, dim3(8 * (((P2 + 63) / 64 + 7) / 8) * nb), dim3(WT), 0, stream, a, a.wino_u);
            return hipGetLastError();
        }
    }
    if (a.conv2d) {
        if ((size_t)a.B * a.xb * 4 >= 0x7FFFFFFFull) {
            // the 2-D gather addresses the input as ONE buffer resource (< 2 GiB): a larger input goes slice by slice over the batch
            if ((size_t)a.xb * 4 >= 0x7FFFFFFFull) return hipErrorInvalidValue;        // (one utterance alone is too large)
            const int per = (int)(0x7FFFFFFEull / ((size_t)a.xb * 4));
            for (int b0 = 0; b0 < a.B; b0 += per) {
                ConvGemmArgs s2 = a;
                s2.B = a.B - b0 < per ? a.B - b0 : per;
                s2.x = a.x + (size_t)b0 * a.xb;
                s2.out = a.out + (size_t)b0 * a.T * a.ldo;
                if (a.res) s2.res = a.res + (size_t)b0 * a.T * a.ldo;
                if (a.rowbias) s2.rowbias = a.rowbias + (size_t)b0 * a.N;
                if (a.row_len) s2.row_len = a.row_len + b0;
                const hipError_t e = gt_launch_conv_gemm(s2, stream);
                if (e != hipSuccess) return e;
            }
            return hipSuccess;
        }
        if (a.N > 64) hipLaunchKernelGGL((gt_conv_gemm_kernel<1, 4, 1, 1, true>), dim3((M + 31) / 32, (a.N + 127) / 128), dim3(256), 0, stream, a);
        else if (a.N > 32) hipLaunchKernelGGL((gt_conv_gemm_kernel<4, 1, 1, 2, true>), dim3((M + 127) / 128, 1), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL((gt_conv_gemm_kernel<4, 1, 1, 1, true>), dim3((M + 127) / 128, 1), dim3(256), 0, stream, a);
        return hipGetLastError();
    }
    if (a.N > 96) {
        const int wg128 = ((M + 127) / 128) * ((a.N + 127) / 128);
        if (wg128 >= 256) {
            dim3 grid((M + 127) / 128, (a.N + 127) / 128);
            hipLaunchKernelGGL((gt_conv_gemm_kernel<2, 2, 2, 2>), grid, dim3(256), 0, stream, a);
        } else if (((M + 63) / 64) * ((a.N + 127) / 128) <= 256) {
            // few rows (the 4096-row encoder convs): 32 x 128 tiles give two workgroups per CU instead of one wave per SIMD
            dim3 grid((M + 31) / 32, (a.N + 127) / 128);
            hipLaunchKernelGGL((gt_conv_gemm_kernel<1, 4, 1, 1>), grid, dim3(256), 0, stream, a);
        } else {
            dim3 grid((M + 63) / 64, (a.N + 127) / 128);
            hipLaunchKernelGGL((gt_conv_gemm_kernel<2, 2, 1, 2>), grid, dim3(256), 0, stream, a);
        }
    } else if (a.N > 64) {
        dim3 grid((M + 127) / 128, 1);
        hipLaunchKernelGGL((gt_conv_gemm_kernel<4, 1, 1, 3>), grid, dim3(256), 0, stream, a);
    } else if (a.N > 32) {
        dim3 grid((M + 127) / 128, 1);
        hipLaunchKernelGGL((gt_conv_gemm_kernel<4, 1, 1, 2>), grid, dim3(256), 0, stream, a);
    } else {
        dim3 grid((M + 127) / 128, 1);
        hipLaunchKernelGGL((gt_conv_gemm_kernel<4, 1, 1, 1>), grid, dim3(256), 0, stream, a);
    }
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void gt_highway_kernel(const float4* z, const float4* x, float4* out, int64_t M, int S4) {
    const int64_t total = M * S4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / S4;
        const int c = (int)(i - m * S4);
        const float4 h = z[m * 2 * S4 + c], t = z[m * 2 * S4 + S4 + c], xv = x[i];
        float4 o;
        float g;
        g = 1.f / (1.f + expf(-t.x)); o.x = fmaxf(h.x, 0.f) * g + xv.x * (1.f - g);
        g = 1.f / (1.f + expf(-t.y)); o.y = fmaxf(h.y, 0.f) * g + xv.y * (1.f - g);
        g = 1.f / (1.f + expf(-t.z)); o.z = fmaxf(h.z, 0.f) * g + xv.z * (1.f - g);
        g = 1.f / (1.f + expf(-t.w)); o.w = fmaxf(h.w, 0.f) * g + xv.w * (1.f - g);
        out[i] = o;
    }
}

hipError_t gt_launch_highway(const float* z, const float* x, float* out, int64_t M, int S, hipStream_t stream) {
    if (S & 3) return hipErrorInvalidValue;
    const int64_t total = M * (S / 4);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gt_highway_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const float4*>(z),
                       reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(out), M, S / 4);
    return hipGetLastError();
}
