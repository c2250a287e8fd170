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

    auto load_slice = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int f = tid + i * 256;
            const int kk = k0 + (f & 7) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (C2D) {                                           // 2-D taps, stride, NHWC input (ConvGemmArgs::conv2d)
                if (a_ok[i] && kk < K) {
                    const int tap = kk / A.Cin;
                    const int c = kk - tap * A.Cin;
                    const int ti = tap / A.kw, tj = tap - ti * A.kw;
                    const int ho = a_t[i] / A.Wo, wo = a_t[i] - ho * A.Wo;
                    const int hi = ho * A.stride + ti - A.pad_h, wi = wo * A.stride + tj - A.pad_w;
                    if (hi >= 0 && hi < A.H && wi >= 0 && wi < A.W)
                        v = *reinterpret_cast<const float4*>(A.x + (int64_t)a_b[i] * A.xb + ((int64_t)hi * A.W + wi) * A.Cin + c);
                }
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

template <int RM>
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
    uint4 rb[B_U4];
    const __bf16* wt = reinterpret_cast<const __bf16*>(A.wt_bf16);
    auto load_slice = [&](int k0) {
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
                    const float* rp = A.tokens ? A.x + (int64_t)A.tokens[rowi] * A.Cin : A.x + rowi * A.Cin;
                    v = *reinterpret_cast<const float4*>(rp + c);
                    if (A.pool2 && ts + 1 < a_len[i]) {
                        const float4 v2 = *reinterpret_cast<const float4*>(rp + A.Cin + c);
                        v.x = fmaxf(v.x, v2.x); v.y = fmaxf(v.y, v2.y); v.z = fmaxf(v.z, v2.z); v.w = fmaxf(v.w, v2.w);
                    }
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_U4; ++i) {
            const int f = tid + i * 256;
            const int n = f >> 3, kq = f & 7;           // 8 pieces of 8 k per column
            rb[i] = *reinterpret_cast<const uint4*>(wt + (size_t)(n0 + n) * A.ldk + k0 + kq * 8);   // padded: always in range
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
            *reinterpret_cast<uint4*>(&Bs[n * LDH + kq * 8]) = rb[i];
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

hipError_t gt_launch_conv_gemm(const ConvGemmArgs& a, hipStream_t stream) {
    const int M = a.B * a.T;
    if (a.wt_bf16) {
        const int nb = (a.N + 127) / 128;
        if (((M + 127) / 128) * nb >= 256) {
            hipLaunchKernelGGL((gt_conv_gemm_bf16_kernel<2>), dim3((M + 127) / 128, nb), dim3(256), 0, stream, a);
        } else {
            hipLaunchKernelGGL((gt_conv_gemm_bf16_kernel<1>), dim3((M + 63) / 64, nb), dim3(256), 0, stream, a);
        }
        return hipGetLastError();
    }
    if (a.conv2d) {
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
