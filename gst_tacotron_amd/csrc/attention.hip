// One decoder-step of the monotonic attentions, one workgroup per utterance.
//
// Replaces reference Modules/Attention/Steps.py:126-166 (score, _apply_scores, context) with
//   SMA  StepwiseMonotonicAttention._monotonic_probability_fn  Steps.py:215-229
//   BMA  BahdanauMonotonicAttention._monotonic_probability_fn  Steps.py:168-199 (safe_cumprod)
// on the HOISTED processed memory pm = Dense_Value(memory) (Steps.py:123 is loop-invariant).
//
//   score[t] = sum_a v[a] * tanh(q[a] + pm[t,a]) + b  (+ sigmoid_noise * N(0,1))
//   p = sigmoid(score);  align = SMA/BMA(p, prev);  ctx[a] = sum_t align[t] * pm[t,a]
//
// gfx950 mapping: the utterance's processed memory (Tv x A fp32 = 64 KiB at 128x128) is staged once per
// step into LDS with coalesced 16-byte loads (row stride A+1 floats so row-per-lane reads for the score and
// column-per-lane reads for the context are both bank-conflict-free) and is read twice from there.
// Scores: one lane per memory row (no cross-lane reduction over a); BMA prefix sums: one wave, serial
// chunk per lane + wave scan; context: one lane per output channel.
#include "device_utils.h"
#include "kernels.h"
#include "../../include/gsttaco.h"

#define ATT_THREADS 256

size_t gt_attn_lds_bytes(int Tv, int A, int loc_f, int loc_k, int* rows_lds) {
    int rows = Tv < 256 ? Tv : 256;
    // keep the tile (+ the LSA location features of its rows) <= 96 KiB
    while ((size_t)rows * (A + 1 + loc_f + 1) * 4 > 96 * 1024 && rows > 16) rows /= 2;
    if (rows_lds) *rows_lds = rows;
    // tile + q + v + score/p + prev + align + partials (+ LSA: location features, dense and conv weights, biases)
    size_t fl = (size_t)rows * (A + 1) + 2 * (size_t)A + 3 * (size_t)Tv + 4 * 256 + 64;
    if (loc_f > 0) fl += (size_t)rows * (loc_f + 1) + (size_t)loc_f * A + (size_t)loc_k * loc_f + loc_f + 2 * (size_t)A;
    return fl * sizeof(float);
}

__device__ __forceinline__ float wave_incl_scan(float x, int lane) { return gt_wave_incl_scan(x, lane); }

__global__ __launch_bounds__(ATT_THREADS) void gt_attn_step_kernel(AttnStepArgs P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int TvFull = P.Tv, A = P.A, LD = A + 1;
    // masked mode: only the first tok_len[b] memory positions exist for this utterance
    const int Tv = P.tok_len ? max(1, min(TvFull, P.tok_len[blockIdx.x])) : TvFull;
    const int rows_lds = P.rows_lds;
    float* tile = smem;                              // [rows_lds][LD]
    float* qs = tile + (size_t)rows_lds * LD;        // [A]
    float* vs = qs + A;                              // [A]
    float* sc = vs + A;                              // [Tv] score -> p
    float* pv = sc + TvFull;                         // [Tv] previous alignment
    float* al = pv + TvFull;                         // [Tv] new alignment
    float* partial = al + TvFull;                    // [4*256]
    // LSA extension scratch
    const bool lsa = P.type == GSTTACO_ATT_LSA;
    const int LF = lsa ? P.loc_f : 0, LK = lsa ? P.loc_k : 0;
    // (row stride LF + 1: in the score pass a lane is a memory ROW and reads its row's features one after the other -- at a stride of
    // 32 floats every lane of a wave hit the same LDS bank, a 64-way conflict on each of 64 x 32 reads per thread: the step-wise LSA
    // extension ran at 325 us per decode step, 165 ms per Inference_Step at the headline shape, tools/lsa_time.py)
    const int LFS = LF + 1;
    float* lfeat = partial + 4 * 256;                // [rows_lds][LF + 1]  conv(state)+bias
    float* ldw = lfeat + (size_t)rows_lds * LFS;     // [LF][A]
    float* lcw = ldw + (size_t)LF * A;               // [LK][LF]
    float* lcb = lcw + (size_t)LK * LF;              // [LF]
    float* labias = lcb + LF;                        // [A]  location-dense bias + additive bias

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const float* pm = P.pm + (size_t)b * TvFull * A;

    for (int a = tid; a < A; a += ATT_THREADS) {
        qs[a] = P.q[(size_t)b * P.ldq + a];
        vs[a] = P.type == GSTTACO_ATT_LSA ? 0.f : P.v[a];      // LSA has no attention_v (Layers.py:407)
    }
    for (int t = tid; t < Tv; t += ATT_THREADS) {
        if (lsa) pv[t] = P.lsa_state[(size_t)b * TvFull + t];       // cumulative (or last) alignment, zeros at step 0
        else pv[t] = P.prev ? P.prev[(size_t)b * P.ldprev + t] : (t == 0 ? 1.f : 0.f);
    }
    if (lsa) {
        for (int i = tid; i < LF * A; i += ATT_THREADS) ldw[i] = P.loc_dw[i];
        for (int i = tid; i < LK * LF; i += ATT_THREADS) lcw[i] = P.loc_cw[i];
        for (int i = tid; i < LF; i += ATT_THREADS) lcb[i] = P.loc_cb[i];
        for (int i = tid; i < A; i += ATT_THREADS) labias[i] = P.loc_db[i] + P.att_bias[i];
    }

    const float bias = lsa ? 0.f : P.score_bias[0];
    const int lpad = (LK - 1) / 2;                   // TF 'same', stride 1: before = (k-1)//2
    const int nchunks = (Tv + rows_lds - 1) / rows_lds;
    const int a4 = A >> 2;

    auto stage = [&](int c) {
        const int r0 = c * rows_lds;
        const int nr = min(rows_lds, Tv - r0);
        const float4* src = reinterpret_cast<const float4*>(pm + (size_t)r0 * A);
        for (int f = tid; f < nr * a4; f += ATT_THREADS) {
            const float4 v = src[f];
            const int row = f / a4, col = (f - row * a4) * 4;
            float* d = tile + row * LD + col;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        return nr;
    };

    // ---- pass 1: scores
    // thread -> (row = tid % RP, part = tid / RP); RP = rows rounded to {64,128,256}
    const int RP = rows_lds <= 64 ? 64 : (rows_lds <= 128 ? 128 : 256);
    const int nparts = ATT_THREADS / RP;
    const int arange = (A + nparts - 1) / nparts;
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();
        const int nr = stage(c);
        __syncthreads();
        if (lsa) {
            // location features of this chunk's rows: Conv1D(state) + bias  (Layers.py:362-363)
            for (int i = tid; i < nr * LF; i += ATT_THREADS) {
                const int rr = i / LF, f = i - rr * LF;
                const int t = c * rows_lds + rr;
                float acc = lcb[f];
                for (int j = 0; j < LK; ++j) {
                    const int ts = t + j - lpad;
                    if (ts >= 0 && ts < Tv) acc += pv[ts] * lcw[j * LF + f];
                }
                lfeat[rr * LFS + f] = acc;
            }
            __syncthreads();
        }
        const int row = tid % RP, part = tid / RP;
        float s = 0.f;
        if (row < nr) {
            const float* tr = tile + row * LD;
            const int abeg = part * arange, aend = min(A, abeg + arange);
            if (lsa) {
                // score = sum_a tanh(q + key + Dense(loc) + bias)   (Layers.py:364, 407; no v vector, scale 1)
                const float* lf = lfeat + (size_t)row * LFS;
                for (int a = abeg; a < aend; ++a) {
                    float loc = labias[a];
                    for (int f = 0; f < LF; ++f) loc += lf[f] * ldw[f * A + a];
                    s += gt_tanh(qs[a] + tr[a] + loc);
                }
            } else {
                for (int a = abeg; a < aend; ++a) s += vs[a] * gt_tanh(qs[a] + tr[a]);
            }
        }
        partial[part * 256 + row] = s;
        __syncthreads();
        if (tid < nr) {
            float z = partial[tid];
            for (int p = 1; p < nparts; ++p) z += partial[p * 256 + tid];
            sc[c * rows_lds + tid] = z + bias;
        }
    }
    __syncthreads();

    if (lsa) {
        // softmax (or smoothing normalisation, Layers.py:426-444) over the Tv positions: one wave, serial chunk per lane
        if (tid < 64) {
            const int per = (Tv + 63) / 64;
            const int t0 = tid * per, t1 = min(Tv, t0 + per);
            float mx = -INFINITY;
            for (int t = t0; t < t1; ++t) mx = fmaxf(mx, sc[t]);
            mx = gt_wave_max(mx);
            float sum = 0.f;
            for (int t = t0; t < t1; ++t) {
                const float e = P.lsa_smoothing ? 1.f / (1.f + expf(-sc[t])) : expf(sc[t] - mx);
                al[t] = e;
                sum += e;
            }
            sum = gt_wave_sum(sum);
            const float inv = 1.f / sum;
            for (int t = t0; t < t1; ++t) {
                al[t] *= inv;
                P.lsa_state[(size_t)b * TvFull + t] = P.lsa_cumulate ? pv[t] + al[t] : al[t];
            }
        }
    } else {
    // ---- noise + sigmoid
    for (int t = tid; t < Tv; t += ATT_THREADS) {
        float s = sc[t];
        if (P.sigmoid_noise > 0.f) {
            float nz;
            if (P.noise) nz = P.noise[(size_t)b * P.ldnoise + t];
            else {
                Philox4 r = gt_philox(*P.seed_ptr, (uint32_t)(b * TvFull + t), P.rng_step, 0u, GT_RNG_NOISE);
                nz = gt_normal(r.x, r.y);
            }
            s += P.sigmoid_noise * nz;
        }
        sc[t] = gt_sigmoid(s);
    }
    __syncthreads();

    // ---- alignment
    if (P.type == GSTTACO_ATT_SMA) {
        for (int t = tid; t < Tv; t += ATT_THREADS) {
            float v = pv[t] * sc[t];
            if (t > 0) v = __builtin_fmaf(pv[t - 1], 1.f - sc[t - 1], v);       // (explicit: a*b + c*d can contract either way)
            al[t] = v;
        }
    } else {
        // BMA: cp = exp(exclusive_cumsum(log(clip(1-p, tiny, 1)))); align = p*cp*cumsum(prev/clip(cp,1e-10,1))
        if (tid < 64) {
            const int per = (Tv + 63) / 64;
            const int t0 = tid * per, t1 = min(Tv, t0 + per);
            float run = 0.f;
            for (int t = t0; t < t1; ++t) run += logf(fminf(fmaxf(1.f - sc[t], 1.17549435e-38f), 1.f));
            float incl = wave_incl_scan(run, tid);
            float base = incl - run;
            for (int t = t0; t < t1; ++t) {
                const float lg = logf(fminf(fmaxf(1.f - sc[t], 1.17549435e-38f), 1.f));
                al[t] = expf(base);            // exclusive cumprod
                base += lg;
            }
            // second scan: cumsum(prev / clip(cp, 1e-10, 1))
            run = 0.f;
            for (int t = t0; t < t1; ++t) run += pv[t] / fminf(fmaxf(al[t], 1e-10f), 1.f);
            incl = wave_incl_scan(run, tid);
            base = incl - run;
            for (int t = t0; t < t1; ++t) {
                base += pv[t] / fminf(fmaxf(al[t], 1e-10f), 1.f);
                al[t] = sc[t] * al[t] * base;
            }
        }
    }
    }   // !lsa
    __syncthreads();
    for (int t = tid; t < TvFull; t += ATT_THREADS) P.align[(size_t)b * P.ldalign + t] = t < Tv ? al[t] : 0.f;

    // ---- pass 2: context  ctx[a] = sum_t al[t] * pm[t][a]; thread -> (a = tid % AP, part = tid / AP)
    const int AP = A <= 64 ? 64 : (A <= 128 ? 128 : 256);
    const int cparts = ATT_THREADS / AP;
    float cacc = 0.f;
    const int ca = tid % AP, cpart = tid / AP;
    for (int c = 0; c < nchunks; ++c) {
        int nr;
        if (nchunks > 1) {
            __syncthreads();
            nr = stage(c);
            __syncthreads();
        } else {
            nr = Tv;
        }
        if (ca < A) {
            const float* alc = al + c * rows_lds;
            for (int t = cpart; t < nr; t += cparts) cacc += alc[t] * tile[t * LD + ca];
        }
    }
    __syncthreads();
    partial[cpart * 256 + ca] = cacc;
    __syncthreads();
    if (tid < A && A <= 256) {
        float z = partial[tid];
        for (int p = 1; p < cparts; ++p) z += partial[p * 256 + tid];
        if (P.ctx_mt > 0) P.ctx[gt_blk_off(b, tid, P.ctx_mt)] = z;
        else P.ctx[(size_t)b * P.ldctx + tid] = z;
    }
}

hipError_t gt_attn_init() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gt_attn_step_kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

__global__ void gt_set_seed_kernel(uint64_t* dst, uint64_t seed) { *dst = seed; }

// Throughput-mode randomness for a whole decode (reference: tf.nn.dropout in the always-on prenet, Taco2.py:283, and the
// SMA sigmoid noise, Steps.py:220-221), generated ONCE per Inference_Step instead of inside every step's dependent chain
// (3 Philox evaluations per thread cost ~0.6 us of the front kernel's 13 us).  Same Philox4x32-10 counters the step
// kernels use, same buffer layout as injected tensors: masks [steps][mask0 B*P0 | mask1 B*P1] of 0/1, noise [steps][B][Tv].
__global__ __launch_bounds__(256) void gt_rng_fill_kernel(const uint64_t* seed_ptr, float* masks, float* noise, int steps, int B,
                                                          int P0, int P1, int Tv, float drop_rate) {
    const uint64_t seed = *seed_ptr;
    const int64_t per_step = (int64_t)B * (P0 + P1);
    const int64_t nmask = masks ? (int64_t)steps * per_step : 0;
    const int64_t nnoise = noise ? (int64_t)steps * B * Tv : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nmask + nnoise; i += (int64_t)gridDim.x * blockDim.x) {
        if (i < nmask) {
            const int t = (int)(i / per_step);
            const int64_t e = i - (int64_t)t * per_step;
            const bool second = e >= (int64_t)B * P0;
            const uint32_t idx = (uint32_t)(second ? e - (int64_t)B * P0 : e);         // b * P + column
            const uint32_t P = (uint32_t)(second ? P1 : P0);
            masks[i] = gt_drop_keep(seed, (uint32_t)t, second ? 1u : 0u, idx / P, idx % P, P, drop_rate);
        } else {
            const int64_t j = i - nmask;
            const int t = (int)(j / ((int64_t)B * Tv));
            const uint32_t idx = (uint32_t)(j - (int64_t)t * B * Tv);                   // b * Tv + position
            const Philox4 r = gt_philox(seed, idx, (uint32_t)t, 0u, GT_RNG_NOISE);
            noise[j] = gt_normal(r.x, r.y);
        }
    }
}

hipError_t gt_launch_rng_fill(const uint64_t* seed_ptr, float* masks, float* noise, int steps, int B, int P0, int P1, int Tv,
                              float drop_rate, hipStream_t stream) {
    const int64_t n = (masks ? (int64_t)steps * B * (P0 + P1) : 0) + (noise ? (int64_t)steps * B * Tv : 0);
    if (n == 0) return hipSuccess;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(gt_rng_fill_kernel, dim3(blocks), dim3(256), 0, stream, seed_ptr, masks, noise, steps, B, P0, P1, Tv, drop_rate);
    return hipGetLastError();
}

// State re-initialisation inside captured graphs is a KERNEL node, not a memset node: on ROCm 7.x a
// hipMemsetAsync node was observed to race with the kernel node that follows it (stale LSTM state after a
// replay with a different batch), while kernel->kernel edges are always honoured.
__global__ __launch_bounds__(256) void gt_zero_kernel(float4* p, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

hipError_t gt_launch_zero(float* p, size_t n_floats, hipStream_t stream) {
    const size_t n4 = (n_floats + 3) / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gt_zero_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<float4*>(p), n4);
    return hipGetLastError();
}

// Injected keep masks of a zero-padded decoder (gsttaco.cpp pad_decoder): the caller's [steps][B * p0 | B * p1] tensor re-laid out as the
// padded model's [steps][B * P0 | B * P1].  Padding columns get 1: whatever they keep is an exact zero (never garbage: NaN * 0).
// to_padded = 0: the other way round (gsttaco_debug_randomness hands back the caller's layout).
__global__ __launch_bounds__(256) void gt_relayout_masks_kernel(const float* src, float* dst, int steps, int B, int p0, int p1, int P0, int P1, int to_padded) {
    const int64_t per_big = (int64_t)B * (P0 + P1), per_small = (int64_t)B * (p0 + p1);
    const int64_t n = (int64_t)steps * (to_padded ? per_big : per_small);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t per = to_padded ? per_big : per_small;
        const int t = (int)(i / per);
        const int64_t e = i - (int64_t)t * per;
        const int W0 = to_padded ? P0 : p0, W1 = to_padded ? P1 : p1;
        const bool second = e >= (int64_t)B * W0;
        const int64_t r = second ? e - (int64_t)B * W0 : e;
        const int W = second ? W1 : W0, b = (int)(r / W), col = (int)(r % W);
        const int w_other = to_padded ? (second ? p1 : p0) : (second ? P1 : P0);
        const int64_t base_other = (int64_t)t * (to_padded ? per_small : per_big) + (second ? (int64_t)B * (to_padded ? p0 : P0) : 0);
        dst[i] = col < w_other ? src[base_other + (int64_t)b * w_other + col] : 1.f;
    }
}

hipError_t gt_launch_relayout_masks(const float* src, float* dst, int steps, int B, int p0, int p1, int P0, int P1, int to_padded, hipStream_t stream) {
    const int64_t n = (int64_t)steps * B * (to_padded ? P0 + P1 : p0 + p1);
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(gt_relayout_masks_kernel, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, stream, src, dst, steps, B, p0, p1, P0, P1, to_padded);
    return hipGetLastError();
}

// Embedding lookup as rows (Taco2.py:18-21): out[row] = table[tokens[row]], 16 bytes per thread.  The encoder's first convolution used to
// resolve the lookup inside its im2col gather (implicit GEMM only); as rows in memory the layer can take the Winograd kernel on the bf16
// pipe like the two behind it (4 096 rows x 512 channels: 8 MB written once, read back from L2).
__global__ __launch_bounds__(256) void gt_embed_rows_kernel(const float4* __restrict__ table, const int32_t* __restrict__ tokens, float4* __restrict__ out, int rows, int c4) {
    const int64_t n = (int64_t)rows * c4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int r = (int)(i / c4), q = (int)(i - (int64_t)r * c4);
        out[i] = table[(int64_t)tokens[r] * c4 + q];
    }
}

hipError_t gt_launch_embed_rows(const float* table, const int32_t* tokens, float* out, int rows, int C, hipStream_t stream) {
    if (C & 3) return hipErrorInvalidValue;
    const int64_t n = (int64_t)rows * (C / 4);
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(gt_embed_rows_kernel, dim3(blocks < 1 ? 1 : blocks), dim3(256), 0, stream, reinterpret_cast<const float4*>(table), tokens,
                       reinterpret_cast<float4*>(out), rows, C / 4);
    return hipGetLastError();
}

// Several device-to-device copies (and the call's seed) as ONE launch: gsttaco_inference_step stages its inputs into the workspace the
// cached graphs read and its outputs out of the one they write -- as three hipMemcpyAsync + a seed kernel in front of the graphs and
// three behind them that was seven dependent ~5 us boundaries per call for ~20 MB of copies the chip moves in a few microseconds.
__global__ __launch_bounds__(256) void gt_copy_segments_kernel(GtCopySegs S) {
    if (S.seed_dst && blockIdx.x == 0 && threadIdx.x == 0) *S.seed_dst = S.seed;
    for (int k = 0; k < S.n; ++k) {
        const size_t n = S.words[k];
        const uint32_t* src = reinterpret_cast<const uint32_t*>(S.src[k]);
        uint32_t* dst = reinterpret_cast<uint32_t*>(S.dst[k]);
        if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
            const size_t n4 = n >> 2;
            const uint4* s4 = reinterpret_cast<const uint4*>(src);
            uint4* d4 = reinterpret_cast<uint4*>(dst);
            for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) d4[i] = s4[i];
            for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
        } else {
            for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
        }
    }
}

hipError_t gt_launch_copy_segments(const GtCopySegs& S, hipStream_t stream) {
    size_t total = 0;
    for (int k = 0; k < S.n; ++k) total += S.words[k];
    if (total == 0 && !S.seed_dst) return hipSuccess;
    size_t blocks = (total / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(gt_copy_segments_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, S);
    return hipGetLastError();
}

hipError_t gt_launch_set_seed(uint64_t* dst, uint64_t seed, hipStream_t stream) {
    hipLaunchKernelGGL(gt_set_seed_kernel, dim3(1), dim3(1), 0, stream, dst, seed);
    return hipGetLastError();
}

hipError_t gt_launch_attn_step(const AttnStepArgs& a, hipStream_t stream) {
    if (a.A > 256 || (a.A & 3)) return hipErrorInvalidValue;
    int rows;
    size_t lds = gt_attn_lds_bytes(a.Tv, a.A, a.type == GSTTACO_ATT_LSA ? a.loc_f : 0, a.type == GSTTACO_ATT_LSA ? a.loc_k : 0, &rows);
    AttnStepArgs p = a;
    p.rows_lds = rows;
    hipLaunchKernelGGL(gt_attn_step_kernel, dim3(a.B), dim3(ATT_THREADS), lds, stream, p);
    return hipGetLastError();
}
