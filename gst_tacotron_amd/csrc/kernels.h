// Host-visible launch interface of the gfx950 kernels (internal; the public ABI is include/gsttaco.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---------------------------------------------------------------- skinny_gemm.hip
enum { EPI_LINEAR = 0, EPI_RELU_DROP = 1, EPI_LSTM = 2 };

struct SkinnySeg {
    const float* ptr;   // [M, len] rows, 16-byte aligned
    int64_t ld;         // row stride in floats (multiple of 4)
    int nkb;            // len / 16
    int pad_;
};

struct SkinnyArgs {
    const float* wp;    // packed weights [ntiles][nkb][64 lanes][4]
    const float* bias;  // [ntiles*16] in packed column order
    SkinnySeg seg[3];
    int nkb;            // sum of seg nkb
    int M;              // batch rows
    int N;              // LINEAR: valid output columns; LSTM: hidden units H
    int n_split;        // LINEAR: columns >= n_split go to out2 (projection: mel | stop)
    float* out; int64_t ldo;
    float* out2; int64_t ldo2;
    // EPI_RELU_DROP
    const float* mask; int64_t ldm;     // keep-mask [M, N] or NULL -> Philox
    float drop_rate, drop_scale;
    const uint64_t* seed_ptr; uint32_t rng_step, rng_stream;   // Philox seed lives in HBM so a cached graph can be re-seeded
    // EPI_LSTM
    float* c;           // [M, H] cell state, updated in place
    float* h; int64_t ldh;
};

hipError_t gt_launch_skinny(int epi, const SkinnyArgs& a0, const SkinnyArgs* a1, int ntiles, hipStream_t stream);

// ---------------------------------------------------------------- gemm_conv.hip
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_TANH = 2 };

struct ConvGemmArgs {
    const float* x;         // [B, T, Cin] channels-last (or embedding table when tokens != NULL)
    const int32_t* tokens;  // optional [B, T]: row (b,t) of x is x[tokens[b,t], :]
    const float* w;         // [taps, Cin, N] (TF Conv1D kernel layout) == [taps*Cin, N]
    const float* scale;     // [N] or NULL (=1)   epilogue: y = acc*scale + shift (+rowbias[b]) ; act ; (+res)
    const float* shift;     // [N] or NULL (=0)
    const float* rowbias;   // [B, N] or NULL
    const float* res;       // [B*T, ldo] residual or NULL
    float* out;             // [B*T, ldo]
    int64_t ldo;
    int B, T, Cin, N, taps, pad_before, act;
};

hipError_t gt_launch_conv_gemm(const ConvGemmArgs& a, hipStream_t stream);

// ---------------------------------------------------------------- attention.hip
struct AttnStepArgs {
    const float* q; int64_t ldq;        // [B, A] projected query
    const float* pm;                    // [B, Tv, A] processed memory (Dense_Value(memory), hoisted)
    const float* v;                     // [A] attention_v
    const float* score_bias;            // [1]
    const float* prev;  int64_t ldprev; // previous alignment rows [B, Tv] (NULL -> one-hot(0), Steps.py:201-206)
    const float* noise; int64_t ldnoise;// N(0,1) [B, Tv] or NULL -> Philox
    float* align; int64_t ldalign;      // out [B, Tv]
    float* ctx; int64_t ldctx;          // out [B, A]
    int B, Tv, A, type;                 // type: GSTTACO_ATT_*
    float sigmoid_noise;
    const uint64_t* seed_ptr; uint32_t rng_step;
    int rows_lds;                       // rows of pm staged in LDS per chunk
};

hipError_t gt_launch_attn_step(const AttnStepArgs& a, hipStream_t stream);
hipError_t gt_attn_init();     // opt in to >64 KiB dynamic LDS; call once outside stream capture
hipError_t gt_launch_set_seed(uint64_t* dst, uint64_t seed, hipStream_t stream);
size_t gt_attn_lds_bytes(int Tv, int A, int* rows_lds);

// ---------------------------------------------------------------- gst.hip
struct Conv2dArgs {
    const float* x;         // [B, H, W, Cin] (batch stride xb floats; lets frame 0 be skipped)
    int64_t xb;
    const float* w;         // [kh, kw, Cin, Cout]
    const float* scale;     // [Cout] folded BN
    const float* shift;
    float* out;             // [B, Ho, Wo, Cout]
    int B, H, W, Cin, Cout, Ho, Wo, k, stride, pad_h, pad_w;
};
hipError_t gt_launch_conv2d_bn_relu(const Conv2dArgs& a, hipStream_t stream);

struct GstTailArgs {
    const float* x;         // [B, T2, gru_in] conv stack output
    const int32_t* mel_len; // [B]
    const float* gru_w;     // [gru_in, 3u]
    const float* gru_u;     // [u, 3u]
    const float* gru_b;     // [2, 3u]
    const float* dense_w;   // [u, D]
    const float* dense_b;   // [D]
    const float* q_w;       // [D, A]
    const float* q_b;       // [A]
    const float* v_tok;     // [ntok, A]  = tanh(tokens).Wv + bv  (batch-invariant, precomputed at finalize)
    const float* ln_g;      // [A]
    const float* ln_b;      // [A]
    float* gst;             // [B, A]
    int B, T2, gru_in, u, D, A, ntok, heads, stride_prod;
};
hipError_t gt_launch_gst_tail(const GstTailArgs& a, hipStream_t stream);
