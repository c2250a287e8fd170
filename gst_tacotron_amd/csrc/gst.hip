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

#define TAIL_THREADS 384

// one workgroup per utterance; thread j owns gate column j (3u columns, looped if 3u > threads)
__global__ __launch_bounds__(TAIL_THREADS) void gt_gst_tail_kernel(GstTailArgs P) {
    extern __shared__ float sm[];
    const int u = P.u, G = 3 * u;
    float* xs = sm;                 // [gru_in]
    float* hs = xs + P.gru_in;      // [u]
    float* mx = hs + u;             // [3u]
    float* mh = mx + G;             // [3u]
    float* ref = mh + G;            // [D]
    float* qv = ref + P.D;          // [A]
    float* ov = qv + P.A;           // [A]
    float* red = ov + P.A;          // [2]
    const int b = blockIdx.x, tid = threadIdx.x;

    // index of the last valid compressed frame: ceil(len / prod(strides)) - 1  (GST.py:38-40,65-68)
    int last = (P.mel_len[b] + P.stride_prod - 1) / P.stride_prod - 1;
    last = max(0, min(last, P.T2 - 1));

    for (int i = tid; i < u; i += TAIL_THREADS) hs[i] = 0.f;
    __syncthreads();
    // GRU (Keras reset_after=True; gate order z,r,h; bias[0]=input bias, bias[1]=recurrent bias).
    // Steps after `last` cannot influence the gathered output, so the loop stops there.
    for (int t = 0; t <= last; ++t) {
        const float* xt = P.x + ((int64_t)b * P.T2 + t) * P.gru_in;
        for (int i = tid; i < P.gru_in; i += TAIL_THREADS) xs[i] = xt[i];
        __syncthreads();
        for (int j = tid; j < G; j += TAIL_THREADS) {
            // 4 independent accumulators: the weight loads of 4 consecutive k are in flight together
            float a0 = P.gru_b[j], a1 = 0.f, a2 = 0.f, a3 = 0.f;
            int k = 0;
            for (; k + 4 <= P.gru_in; k += 4) {
                a0 += xs[k] * P.gru_w[(int64_t)k * G + j];
                a1 += xs[k + 1] * P.gru_w[(int64_t)(k + 1) * G + j];
                a2 += xs[k + 2] * P.gru_w[(int64_t)(k + 2) * G + j];
                a3 += xs[k + 3] * P.gru_w[(int64_t)(k + 3) * G + j];
            }
            for (; k < P.gru_in; ++k) a0 += xs[k] * P.gru_w[(int64_t)k * G + j];
            mx[j] = (a0 + a1) + (a2 + a3);
            float r0 = P.gru_b[G + j], r1 = 0.f, r2 = 0.f, r3 = 0.f;
            for (k = 0; k + 4 <= u; k += 4) {
                r0 += hs[k] * P.gru_u[(int64_t)k * G + j];
                r1 += hs[k + 1] * P.gru_u[(int64_t)(k + 1) * G + j];
                r2 += hs[k + 2] * P.gru_u[(int64_t)(k + 2) * G + j];
                r3 += hs[k + 3] * P.gru_u[(int64_t)(k + 3) * G + j];
            }
            for (; k < u; ++k) r0 += hs[k] * P.gru_u[(int64_t)k * G + j];
            mh[j] = (r0 + r1) + (r2 + r3);
        }
        __syncthreads();
        for (int i = tid; i < u; i += TAIL_THREADS) {
            const float z = 1.f / (1.f + expf(-(mx[i] + mh[i])));
            const float r = 1.f / (1.f + expf(-(mx[u + i] + mh[u + i])));
            const float hh = tanhf(mx[2 * u + i] + r * mh[2 * u + i]);
            hs[i] = z * hs[i] + (1.f - z) * hh;
        }
        __syncthreads();
    }
    // Dense tanh (GST.py:42-45)
    for (int j = tid; j < P.D; j += TAIL_THREADS) {
        float a = P.dense_b[j];
        for (int k = 0; k < u; ++k) a += hs[k] * P.dense_w[(int64_t)k * P.D + j];
        ref[j] = tanhf(a);
    }
    __syncthreads();
    // query projection (Layers.py:174)
    for (int j = tid; j < P.A; j += TAIL_THREADS) {
        float a = P.q_b[j];
        for (int k = 0; k < P.D; ++k) a += ref[k] * P.q_w[(int64_t)k * P.A + j];
        qv[j] = a;
    }
    __syncthreads();
    // per head: scores = q_h . v_h^T (no scaling, F13), softmax over tokens, out_h = P . v_h
    const int dh = P.A / P.heads;
    for (int j = tid; j < P.A; j += TAIL_THREADS) {
        const int h = j / dh;
        float mxs = -INFINITY;
        for (int n = 0; n < P.ntok; ++n) {
            float s = 0.f;
            for (int d = 0; d < dh; ++d) s += qv[h * dh + d] * P.v_tok[n * P.A + h * dh + d];
            mxs = fmaxf(mxs, s);
        }
        float den = 0.f, num = 0.f;
        for (int n = 0; n < P.ntok; ++n) {
            float s = 0.f;
            for (int d = 0; d < dh; ++d) s += qv[h * dh + d] * P.v_tok[n * P.A + h * dh + d];
            const float e = expf(s - mxs);
            den += e;
            num += e * P.v_tok[n * P.A + j];
        }
        ov[j] = num / den + qv[j];                 // residual adds the PROJECTED query (Layers.py:211)
    }
    __syncthreads();
    // LayerNorm, population variance, eps inside the sqrt (Layers.py:280-283)
    if (tid == 0) {
        float m = 0.f;
        for (int j = 0; j < P.A; ++j) m += ov[j];
        m /= P.A;
        float v = 0.f;
        for (int j = 0; j < P.A; ++j) v += (ov[j] - m) * (ov[j] - m);
        v /= P.A;
        red[0] = m;
        red[1] = 1.f / sqrtf(v + 1e-8f);
    }
    __syncthreads();
    for (int j = tid; j < P.A; j += TAIL_THREADS)
        P.gst[(int64_t)b * P.A + j] = P.ln_g[j] * ((ov[j] - red[0]) * red[1]) + P.ln_b[j];
}

hipError_t gt_launch_gst_tail(const GstTailArgs& a, hipStream_t stream) {
    size_t lds = (size_t)(a.gru_in + a.u + 6 * a.u + a.D + 2 * a.A + 8) * sizeof(float);
    hipLaunchKernelGGL(gt_gst_tail_kernel, dim3(a.B), dim3(TAIL_THREADS), lds, stream, a);
    return hipGetLastError();
}
