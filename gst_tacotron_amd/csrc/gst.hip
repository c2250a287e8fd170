// GST (global style token) path: reference-encoder Conv2D stack and the per-utterance tail.
//
// gt_conv2d_bn_relu_kernel  <- reference Modules/GST.py:23-31,55-56: Conv2D 3x3 stride 2 padding 'same'
//     (TF asymmetric pads) no bias + BatchNorm (folded) + ReLU, NHWC = [B, time, freq, C].
// gt_gst_tail_kernel        <- GST.py:57-70 (reshape, GRU reset_after, gather at ceil(len/64)-1, Dense tanh)
//     + GST.py:100-109 / Layers.py:172-214,230-237,280-283 (4-head unscaled attention over tanh(tokens),
//     residual with the projected query, LayerNorm eps 1e-8).  The token-side projection
//     tanh(tokens).Wv+bv is batch-invariant and precomputed at finalize.
//
// The whole GST path is ~25 MMAC per utterance, run once per batch: these are plain VALU kernels with
// coalesced channel-fastest accesses; the decode loop is where the time goes.
#include <algorithm>
#include "device_utils.h"
#include "kernels.h"

// One thread per (b, ho, wo, 4 consecutive output channels): the input pixel is a wave-wide broadcast load, the
// weights one coalesced 16-byte load per (tap, ci) shared by 4 FMAs.  Cout % 4 == 0 (checked on the host side).
__global__ __launch_bounds__(256) void gt_conv2d_bn_relu_kernel(Conv2dArgs P) {
    const int C4 = P.Cout >> 2;
    const int64_t total = (int64_t)P.B * P.Ho * P.Wo * C4;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = idx % C4;
        int64_t rest = idx / C4;
        const int wo = rest % P.Wo; rest /= P.Wo;
        const int ho = rest % P.Ho;
        const int b = rest / P.Ho;
        const float* xb = P.x + (int64_t)b * P.xb;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int i = 0; i < P.k; ++i) {
            const int hi = ho * P.stride + i - P.pad_h;
            if (hi < 0 || hi >= P.H) continue;
            for (int j = 0; j < P.k; ++j) {
                const int wi = wo * P.stride + j - P.pad_w;
                if (wi < 0 || wi >= P.W) continue;
                const float* xp = xb + ((int64_t)hi * P.W + wi) * P.Cin;
                const float4* wp = reinterpret_cast<const float4*>(P.w + ((int64_t)(i * P.k + j) * P.Cin) * P.Cout) + c4;
#pragma unroll 4
                for (int ci = 0; ci < P.Cin; ++ci) {
                    const float xv = xp[ci];
                    const float4 wv = wp[(int64_t)ci * C4];
                    acc.x += xv * wv.x; acc.y += xv * wv.y; acc.z += xv * wv.z; acc.w += xv * wv.w;
                }
            }
        }
        const float4 sc = reinterpret_cast<const float4*>(P.scale)[c4];
        const float4 sh = reinterpret_cast<const float4*>(P.shift)[c4];
        float4 o;
        o.x = fmaxf(acc.x * sc.x + sh.x, 0.f); o.y = fmaxf(acc.y * sc.y + sh.y, 0.f);
        o.z = fmaxf(acc.z * sc.z + sh.z, 0.f); o.w = fmaxf(acc.w * sc.w + sh.w, 0.f);
        reinterpret_cast<float4*>(P.out)[idx] = o;
    }
}

hipError_t gt_launch_conv2d_bn_relu(const Conv2dArgs& a, hipStream_t stream) {
    if (a.Cout & 3) return hipErrorInvalidValue;
    const int64_t total = (int64_t)a.B * a.Ho * a.Wo * (a.Cout >> 2);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gt_conv2d_bn_relu_kernel, dim3(blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

#define TAIL_THREADS 1024
#define TAIL_MAXT 8            // compressed reference frames whose input halves are computed in one pass

// partial[kp][N] <- x[K] . W[K, N] split over KP = TAIL_THREADS / (N/4) k-parts; lane = 4 consecutive columns (one coalesced
// 16-byte load per row), 8 rows requested at a time.  NT input vectors (xs + t * ldx) share every weight load.
// N % 4 == 0, N / 4 <= TAIL_THREADS.  The caller syncs, then sums partial[0..KP)[j].
template <int NT>
__device__ __forceinline__ int tail_gemv(const float* __restrict__ W, int K, int N, const float* xs, int ldx, int nt, float* partial) {
    const int n4 = N >> 2, KP = TAIL_THREADS / n4;
    const int c4 = threadIdx.x % n4, kp = threadIdx.x / n4;
    const int rows = (K + KP - 1) / KP;
    if (kp < KP) {
        float4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int k0 = kp * rows, k1 = min(K, k0 + rows);
        for (int k = k0; k < k1; k += 8) {
            float4 w[8];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                w[i] = (k + i < k1) ? *reinterpret_cast<const float4*>(W + (size_t)(k + i) * N + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (k + i < k1) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        if (t < nt) {
                            const float x = xs[t * ldx + k + i];
                            acc[t].x += x * w[i].x; acc[t].y += x * w[i].y; acc[t].z += x * w[i].z; acc[t].w += x * w[i].w;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
            if (t < nt) *reinterpret_cast<float4*>(partial + ((size_t)t * KP + kp) * N + c4 * 4) = acc[t];
    }
    return KP;
}

__device__ __forceinline__ float tail_sum(const float* partial, int KP, int N, int j) {
    float z = 0.f;
    for (int p = 0; p < KP; ++p) z += partial[(size_t)p * N + j];
    return z;
}

// one 1024-thread workgroup per utterance.  The GRU's input halves x_t . W + b_i of every needed frame are one pass over W
// (they do not depend on the state); each recurrent step is then one K = u GEMV split over all lanes.
__global__ __launch_bounds__(TAIL_THREADS) void gt_gst_tail_kernel(GstTailArgs P) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int u = P.u, G = 3 * u;
    const int b = blockIdx.x, tid = threadIdx.x;
    // index of the last valid compressed frame: ceil(len / prod(strides)) - 1  (GST.py:38-40,65-68)
    int last = (P.mel_len[b] + P.stride_prod - 1) / P.stride_prod - 1;
    last = max(0, min(last, P.T2 - 1));
    const int KPG = TAIL_THREADS / (G >> 2);
    float* xs = sm;                                 // [TAIL_MAXT][gru_in]
    float* hs = xs + TAIL_MAXT * P.gru_in;          // [u]
    float* mx = hs + u;                             // [TAIL_MAXT][3u]  input halves + input bias
    float* mh = mx + TAIL_MAXT * G;                 // [3u]
    float* ref = mh + G;                            // [D]
    float* qv = ref + P.D;                          // [A]
    float* ov = qv + P.A;                           // [A]
    float* red = ov + P.A;                          // [64]
    float* partial = red + 64;                      // [TAIL_MAXT][KPG][3u] (>= every other GEMV's partials)

    for (int i = tid; i < u; i += TAIL_THREADS) hs[i] = 0.f;
    // GRU (Keras reset_after=True; gate order z,r,h; bias[0]=input bias, bias[1]=recurrent bias).
    // Steps after `last` cannot influence the gathered output, so the loop stops there.
    for (int t0 = 0; t0 <= last; t0 += TAIL_MAXT) {
        const int nt = min(TAIL_MAXT, last + 1 - t0);
        __syncthreads();
        for (int i = tid; i < nt * P.gru_in; i += TAIL_THREADS)
            xs[i] = P.x[((int64_t)b * P.T2 + t0) * P.gru_in + i];
        __syncthreads();
        const int KP = tail_gemv<TAIL_MAXT>(P.gru_w, P.gru_in, G, xs, P.gru_in, nt, partial);
        __syncthreads();
        for (int i = tid; i < nt * G; i += TAIL_THREADS) {
            const int t = i / G, j = i - t * G;
            mx[i] = P.gru_b[j] + tail_sum(partial + (size_t)t * KP * G, KP, G, j);
        }
        for (int t = 0; t < nt; ++t) {
            __syncthreads();
            const int KPh = tail_gemv<1>(P.gru_u, u, G, hs, 0, 1, partial);
            __syncthreads();
            for (int j = tid; j < G; j += TAIL_THREADS) mh[j] = P.gru_b[G + j] + tail_sum(partial, KPh, G, j);
            __syncthreads();
            for (int i = tid; i < u; i += TAIL_THREADS) {
                const float* m = mx + (size_t)t * G;
                const float z = 1.f / (1.f + expf(-(m[i] + mh[i])));
                const float r = 1.f / (1.f + expf(-(m[u + i] + mh[u + i])));
                const float hh = tanhf(m[2 * u + i] + r * mh[2 * u + i]);
                hs[i] = z * hs[i] + (1.f - z) * hh;
            }
        }
    }
    __syncthreads();
    // Dense tanh (GST.py:42-45)
    {
        const int KP = tail_gemv<1>(P.dense_w, u, P.D, hs, 0, 1, partial);
        __syncthreads();
        for (int j = tid; j < P.D; j += TAIL_THREADS) ref[j] = tanhf(P.dense_b[j] + tail_sum(partial, KP, P.D, j));
        __syncthreads();
    }
    // query projection (Layers.py:174)
    {
        const int KP = tail_gemv<1>(P.q_w, P.D, P.A, ref, 0, 1, partial);
        __syncthreads();
        for (int j = tid; j < P.A; j += TAIL_THREADS) qv[j] = P.q_b[j] + tail_sum(partial, KP, P.A, j);
        __syncthreads();
    }
    // per head: scores = q_h . v_h^T (no scaling, F13), softmax over tokens, out_h = P . v_h
    const int dh = P.A / P.heads;
    float* sc = partial;                            // [heads][ntok]
    for (int i = tid; i < P.heads * P.ntok; i += TAIL_THREADS) {
        const int h = i / P.ntok, n = i - h * P.ntok;
        float s = 0.f;
        for (int d = 0; d < dh; ++d) s += qv[h * dh + d] * P.v_tok[n * P.A + h * dh + d];
        sc[i] = s;
    }
    __syncthreads();
    for (int j = tid; j < P.A; j += TAIL_THREADS) {
        const int h = j / dh;
        float mxs = -INFINITY;
        for (int n = 0; n < P.ntok; ++n) mxs = fmaxf(mxs, sc[h * P.ntok + n]);
        float den = 0.f, num = 0.f;
        for (int n = 0; n < P.ntok; ++n) {
            const float e = expf(sc[h * P.ntok + n] - mxs);
            den += e;
            num += e * P.v_tok[n * P.A + j];
        }
        ov[j] = num / den + qv[j];                 // residual adds the PROJECTED query (Layers.py:211)
    }
    __syncthreads();
    // LayerNorm, population variance, eps inside the sqrt (Layers.py:280-283): wave 0 reduces
    if (tid < 64) {
        float m = 0.f;
        for (int j = tid; j < P.A; j += 64) m += ov[j];
        for (int d = 32; d > 0; d >>= 1) m += __shfl_xor(m, d, 64);
        m /= P.A;
        float v = 0.f;
        for (int j = tid; j < P.A; j += 64) v += (ov[j] - m) * (ov[j] - m);
        for (int d = 32; d > 0; d >>= 1) v += __shfl_xor(v, d, 64);
        v /= P.A;
        if (tid == 0) { red[0] = m; red[1] = 1.f / sqrtf(v + 1e-8f); }
    }
    __syncthreads();
    for (int j = tid; j < P.A; j += TAIL_THREADS)
        P.gst[(int64_t)b * P.A + j] = P.ln_g[j] * ((ov[j] - red[0]) * red[1]) + P.ln_b[j];
}

static size_t tail_lds_bytes(const GstTailArgs& a) {
    const int G = 3 * a.u;
    const size_t KPG = TAIL_THREADS / (G / 4);
    size_t part = (size_t)TAIL_MAXT * KPG * G;
    auto need = [&](int N) { return (size_t)(TAIL_THREADS / (N / 4)) * N; };
    part = std::max<size_t>({part, need(a.D), need(a.A), (size_t)a.heads * a.ntok});
    return ((size_t)TAIL_MAXT * a.gru_in + a.u + (size_t)TAIL_MAXT * G + G + a.D + 2 * (size_t)a.A + 64 + part) * sizeof(float);
}

hipError_t gt_launch_gst_tail(const GstTailArgs& a, hipStream_t stream) {
    const int G = 3 * a.u;
    if ((G & 3) || (a.D & 3) || (a.A & 3) || G / 4 > TAIL_THREADS || a.D / 4 > TAIL_THREADS || a.A / 4 > TAIL_THREADS || (a.gru_in & 3))
        return hipErrorInvalidValue;
    const size_t lds = tail_lds_bytes(a);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(gt_gst_tail_kernel, dim3(a.B), dim3(TAIL_THREADS), lds, stream, a);
    return hipGetLastError();
}

hipError_t gt_gst_init() {     // opt in to >64 KiB dynamic LDS; call once outside stream capture
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gt_gst_tail_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
