// Host-visible launch interface of the gfx950 kernels (internal; the public ABI is include/gsttaco.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---------------------------------------------------------------- skinny_gemm.hip
enum { EPI_LINEAR = 0, EPI_RELU_DROP = 1, EPI_LSTM = 2, EPI_PARTIAL = 3 };

// Activation ("A operand") layouts of a [M, K] matrix:
//   row-major : element (row, k) at ptr[row*ld + k]
//   blocked   : MFMA-fragment order [k/16][row/16][lane = ((k%16)/4)*16 + row%16][k%4], i.e. one 1-KiB block per
//               (16 k, 16 rows): a wave reads its A fragment of a k-block with ONE coalesced 16-byte-per-lane load
//               instead of 16 rows x 64 B at a power-of-two row stride.  All decode-loop activations use it.
struct SkinnySeg {
    const float* ptr;   // row-major: [M, len] rows, 16-byte aligned; blocked: first k-block of the segment
    int64_t ld;         // row-major: row stride in floats (multiple of 4); blocked: unused
    int nkb;            // len / 16
    int blocked;        // 1 = blocked layout
};
__host__ __device__ inline size_t gt_blk_off(int row, int k, int MT) {
    return (((size_t)(k >> 4) * MT + (row >> 4)) << 8) + (size_t)(((((k & 15) >> 2) << 4) + (row & 15)) * 4 + (k & 3));
}

// bf16 MIRROR of a blocked activation buffer (mixed precision, batches above 32 rows): [K/32][MT][64 lanes][8 bf16], lane l of 32-k
// block j and M-tile mt holds row 16 mt + (l & 15), slot i <-> k = 32 j + 16 (i >> 2) + 4 (l >> 4) + (i & 3): exactly the A operand of
// v_mfma_f32_16x16x32_bf16 the consumers used to assemble from two fp32 fragments (lean_body.h), rounded (RNE) once by the producer
// instead of by every consumer.  Offset in bf16 elements.
__host__ __device__ inline size_t gt_blk_off_h(int row, int k, int MT) {
    return ((((size_t)(k >> 5) * MT + (row >> 4)) << 6) + (size_t)((((k & 15) >> 2) << 4) + (row & 15))) * 8 + (size_t)((((k >> 4) & 1) << 2) + (k & 3));
}

struct SkinnyArgs {
    const float* wp;    // packed weights [ntiles][nkb][64 lanes][4]; when `bf16`: bf16 [ntiles][ceil(nkb/2)][64 lanes][8]
    const float* bias;  // [ntiles*16] in packed column order
    SkinnySeg seg[3];
    int nkb;            // sum of seg nkb
    int M;              // batch rows
    int N;              // LINEAR: valid output columns; LSTM: hidden units H
    int n_split;        // LINEAR: columns >= n_split go to out2 (projection: mel | stop)
    int MT;             // M-tiles (ceil(M/16)) of blocked operands / outputs
    int out_blocked;    // LINEAR: `out` is a blocked activation buffer; LSTM: `h` is
    float* out; int64_t ldo;
    float* out2; int64_t ldo2;
    // LINEAR, optional third region: columns [col3, N) go to out3, [n_split, n_valid2) to out2, [n_valid2, col3) are padding
    float* out3; int64_t ldo3; int col3, n_valid2;
    // EPI_RELU_DROP
    const float* mask; int64_t ldm;     // keep-mask [M, N] or NULL -> Philox
    float drop_rate, drop_scale;
    const uint64_t* seed_ptr; uint32_t rng_step, rng_stream;   // Philox seed lives in HBM so a cached graph can be re-seeded
    // EPI_LSTM
    float* c;           // [M, H] cell state, updated in place
    float* h; int64_t ldh;
    // EPI_PARTIAL writes / EPI_LSTM adds pre-activation partial sums, tile order [tile][MT*16 rows][16 cols]
    float* partial_out;
    const float* partial_in;
    // masked-mode extension: EPI_LSTM rows whose time index t_index >= row_len[row] do not exist --
    // h is written as 0 and the cell state is left untouched (NULL = the reference's unmasked behaviour)
    const int32_t* row_len; int t_index;
    int bf16;           // mixed precision (Use_Mixed_Precision): weights are bf16, activations are rounded to bf16 on load,
                        // products accumulate in fp32 on v_mfma_f32_16x16x32_bf16; everything else stays fp32
    int keep_weights;   // 1: load the weights with the default cache policy (they are re-read every step and small enough
                        // to stay L2-resident) instead of non-temporally
    unsigned long long* dbg;   // diagnostic phase stamps (s_memrealtime, 100 MHz) of block 0, or NULL
};

// Recurrent-half worker job (lean_body.h gt_lean_partial): tiles of  h . W_h + b  as pre-activation partial sums.
struct LeanPartialArgs {
    const float* wp;            // packed weights [tiles][NKB][64][4]
    const float* bias;          // [tiles*16]
    const float* x;             // blocked state [NKB][MT][64][4]
    float* partial_out;         // [tile][MT*16 rows][16 cols]
    int MT;
    const uint16_t* xh = nullptr;   // bf16 mirror of x (gt_blk_off_h) or NULL: read instead of x by the bf16 multi-chunk bodies
};

enum { TAG_GENERIC = 0, TAG_DEC_LSTM1 = 1, TAG_DEC_LSTM2 = 2, TAG_ENC_BILSTM = 3 };
hipError_t gt_launch_skinny(int epi, const SkinnyArgs& a0, const SkinnyArgs* a1, int ntiles, hipStream_t stream,
                            int tag = TAG_GENERIC);
// EPI_LINEAR main GEMM (tiles [0, ntiles)) plus co-scheduled worker workgroups in the same launch that compute tiles
// [co_begin, co_end) of an independent EPI_PARTIAL GEMM `co` on CUs the small main grid leaves idle.
hipError_t gt_launch_skinny_co(const SkinnyArgs& main_args, int ntiles, const SkinnyArgs& co, int co_begin, int co_end,
                               int tiles_per_worker, hipStream_t stream);

// ---- lean decode-step kernels (lean_body.h): fp32, blocked operands, compile-time K; the host falls back to the
// general kernels above for any other shape or precision.
// Input half of a decode LSTM cell: z = x . W_x + partial_in (recurrent half + bias, from the workers), gates, cell update.
struct LstmXArgs {
    const float* wp;            // packed W_x [H/4 tiles][nkb][64][4]
    const float* x;             // blocked input [nkb][MT][64][4]
    const float* partial_in;    // [tiles][MT*16][16]
    float* c;                   // [M, H] cell state, in place
    float* h;                   // blocked output state
    const int32_t* row_len;     // masked-mode extension (A12) or NULL
    unsigned long long* dbg;
    int M, MT, H, t_index;
    int nkb;                    // K / 16 (the bf16 variant's 32-k blocks need not fill NW x KPW)
    const uint16_t* xh = nullptr;   // bf16 mirror of x or NULL (read by the bf16 multi-chunk body instead of x)
    uint16_t* hh = nullptr;         // bf16 mirror of h or NULL (written beside h)
};
bool gt_lstm_x_supported(int nkb);
// Both cells in one launch with an in-kernel hand-off of h1 (skinny_gemm.hip gt_lstm12_kernel): fp32, batch <= 32.
#ifndef GT_L12_NSH
#define GT_L12_NSH 8        // shards (a 128-byte line each) of the fused LSTM launches' arrival counter; GT_L12_NSH * 32 words per decode step
#endif
struct Lstm12Args {
    LstmXArgs l1, l2;           // exactly the two launches' arguments (l2.x == l1.h)
    uint32_t* arrive;           // GT_L12_NSH x 32 words, zero before the launch: per-shard arrival counters of THIS decode step
    uint32_t* err;              // host-mapped give-up word
    uint32_t expect;            // arrivals to wait for = the grid size (fault injection: one more, so the wait runs into its bound)
};
bool gt_lstm12_supported(int nkb1, int nkb2, int H1, int H2, int M, int slots);      // slots: resident workgroups = occupancy x CUs
int gt_lstm12_blocks_per_cu(int which);     // occupancy of the fused kernels: 0 = gt_lstm12_kernel, 1 / 2 = gt_lstm12_mc_kernel fp32 / bf16
hipError_t gt_launch_lstm12(const Lstm12Args& a, hipStream_t stream);
bool gt_lstm12_mc_supported(int nkb1, int nkb2, int H1, int H2, int M, int slots);     // the same at batches above 32 rows (fp32 / bf16)
int gt_lstm12_mc_grid(int H);
hipError_t gt_launch_lstm12_mc(const Lstm12Args& a, bool bf16, hipStream_t stream);
hipError_t gt_launch_lstm_x(const LstmXArgs& a, int nkb, int tag, bool bf16, hipStream_t stream);

// One time step of a Bidirectional LSTM (reference Taco2.py:39-43, 394-398) whose input halves x_t . W_x + b were hoisted
// into ONE GEMM over all time steps (columns in tile order: tile*16 + gate*4 + unit%4): z = zx + h_{t-1} . W_h, gates, cell
// update.  grid = (H/4 tiles, ceil(M/32), 2 directions).
struct BiLstmDir {
    const float* wp;        // packed W_h [H/4 tiles][H/16][64][4]
    const float* hprev;     // blocked state [H/16][MT][64][4]
    float* hnext;
    float* c;               // [M, H]
    const float* zx;        // hoisted input half of THIS time step: element (row, tile, col) at zx[row*ldz + tile*16 + col]
    float* out;             // row-major output of this time step: (row, unit) at out[row*ldo + unit]
    int t_index;            // masked-mode extension (A12): rows with t_index >= row_len[row] write h = 0 and keep c
};
struct BiLstmArgs {
    BiLstmDir d[2];
    const int32_t* row_len;
    int64_t ldz, ldo;
    int M, MT, H;
};
bool gt_bilstm_lean_supported(int nkb_h);

// Persistent BiLSTM: every time step of both directions in one launch (skinny_gemm.hip gt_bilstm_persist_kernel).
// Group g = direction + 2 * M-tile lives on XCD g.
struct BiLstmPersistArgs {
    const float* wp[2];         // packed W_h per direction [H/4 tiles][H/16][64][4] (bf16 pack [tiles][H/32][64][8] in mixed precision)
    const float* zx;            // hoisted input halves: (row, time tt, direction d, tile, col) at zx[row*ldz + tt*8H + d*4H + tile*16 + col]
    float* out;                 // (row, tt, d, unit) at out[row*ldo + tt*2H + d*H + unit]
    float* h;                   // workspace [8 groups][3 slots][H/16][64][4]: blocked state of the group's 16 rows, each word tagged in bit 30;
                                // ZEROED before the launch (tag 0 = never written)
    uint32_t* flags;            // [8] member counters, zeroed before the launch (the words right behind `h`: one zero-fill for both)
    const int32_t* row_len;     // masked-mode extension (A12) or NULL
    uint32_t* err;              // host-mapped; bit 0: a wait gave up (every member then leaves the launch)
    int64_t ldz, ldo;
    int M, MT, H, T;
    int debug_drop_member;      // -1; >= 0: fault injection, that member rank of every group exits at once (tests)
};
bool gt_bilstm_persist_supported(int H, int B, int n_cu);
hipError_t gt_launch_bilstm_persist(const BiLstmPersistArgs& a, bool bf16, hipStream_t stream);
hipError_t gt_bilstm_persist_init();       // opt in to >64 KiB dynamic LDS; call once outside stream capture
int gt_bilstm_persist_blocks_per_cu();     // occupancy of the persistent kernel (must be 1)
hipError_t gt_launch_bilstm_lean(const BiLstmArgs& a, hipStream_t stream);

// Projection [h2 | ctx] -> mel frames | stop logit | (optional) next step's prenet-0 pre-activations, with co-scheduled
// recurrent-half worker tiles of LSTM layer 2 for the next step in the same launch.
struct ProjArgs {
    const float* wp; const float* bias;
    const float* xa; const float* xb; int nkb_a;    // blocked segments: k-blocks [0, nkb_a) from xa, the rest from xb
    int M, MT, N;                                   // N: valid columns
    int n_split, n_valid2, col3;                    // [0,n_split) -> out, [n_split,n_valid2) -> out2, [col3,N) -> out3 (NULL: none)
    float* out; int64_t ldo;
    float* out2; int64_t ldo2;
    float* out3; int64_t ldo3;
    unsigned long long* dbg;                        // diagnostic stamps (8 slots) or NULL
    const uint16_t* xah = nullptr; const uint16_t* xbh = nullptr;   // bf16 mirrors of xa / xb or NULL (gt_proj_mc_kernel<true>)
    int both_m = 0;     // batches of 17..32 rows: ONE workgroup per tile multiplies both 16-row M-tiles (the tile's weights leave the
                        // memory side once) instead of one workgroup per (tile, M-tile)
};
bool gt_proj_lean_supported(int nkb_main, int nkb_co);
hipError_t gt_launch_proj_lean(const ProjArgs& m, int ntiles, const float* co_wp, const float* co_bias, const float* co_x,
                               float* co_out, int co_begin, int co_end, int tiles_per_worker, bool bf16, hipStream_t stream);

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
    const int32_t* row_len; // masked-mode extension (A12): input rows t >= row_len[b] read as zero, or NULL
    int ldw;                // row stride of w in floats (0 = N); lets N be odd (513) over a zero-padded multiple of 4
    int pool2;              // 1: the input row is max(x[t], x[t+1]) -- MaxPool1D(2, stride 1, 'same') fused into the gather
    // mixed precision (Use_Mixed_Precision): the weights TRANSPOSED in bf16, [ceil(N/256)*256][ldk] with k contiguous and
    // zero padding (ldk = ceil(taps*Cin/64)*64), or NULL = fp32 path.  Activations are rounded to bf16 on their way into
    // LDS, products accumulate in fp32 (v_mfma_f32_32x32x16_bf16), the epilogue and the output stay fp32.
    const void* wt_bf16;
    int ldk;
    // mixed precision, activations BETWEEN bf16 layers: `x` / `out` hold bf16 instead of fp32 (same [rows, channels] layout, half the
    // bytes).  The consumer rounds its input to bf16 on the way into LDS anyway, so storing the rounded value changes no result.
    // Only the bf16 kernels honour them (the host sets them for layers it knows run there); residual input and `res` stay fp32.
    int x_bf16, out_bf16;
    // Winograd F(2,5) (gemm_conv.hip): the transformed weights U[6][Cin][N] = G . w (float64 at finalize), or NULL
    const float* wino_u;    // F(2,5): [6][wino_cin][N]
    const float* wino_u4;   // F(4,5): [8][wino_cin][N], or NULL
    int wino_cin;           // rows of each U_xi: Cin rounded up to a multiple of 32 (zero rows for the padding)
    // the same U as three bf16 planes (hi + mid + lo = the float64 transform to 2^-25), [xi][plane][wino_npad columns][wino_cin], k
    // contiguous: the transform-domain GEMMs then run as split-bf16 x6 on the bf16 matrix pipe at fp32 accuracy (conv_wino_split.hip)
    const void* wino_s;     // F(2,5), or NULL
    const void* wino_s4;    // F(4,5), or NULL
    int wino_npad;          // N rounded up to the kernel's 128-column block
    const void* gemm_s;     // taps == 1: W itself as three bf16 planes [plane][wino_npad][Cin], k contiguous -> the plain split-bf16 GEMM
                            // (conv_wino_split.hip gt_gemm_split_kernel), or NULL
    int wino_x3;            // 1 = the split Winograd kernel's reduced form (two planes, products hh hm mh: ~2^-16, GSTTACO_WINO_SPLIT=3)
    int wino_min_wgs;       // 0 = the default grid-fill rule (Winograd only where >= 240 workgroups); else the caller's threshold -- the
                            // split-bf16 kernel also pays on the encoder's 4 096-row layers, whose F(2,5) grid is 128 workgroups
    float* out;             // [B*T, ldo]
    int64_t ldo;
    int B, T, Cin, N, taps, pad_before, act;
    // 2-D mode (GST reference encoder, GST.py:23-31): rows are (b, ho, wo) with T = Ho*Wo, k = (i*kw + j)*Cin + c reads
    // x[b][ho*stride + i - pad_h][wo*stride + j - pad_w][c] of an NHWC input with batch stride xb floats; taps = kh*kw.
    int conv2d, H, W, Wo, kw, stride, pad_h, pad_w;
    int64_t xb;
};

hipError_t gt_launch_conv_gemm(const ConvGemmArgs& a, hipStream_t stream);
hipError_t gt_conv5_bf16_init();           // opt in to >64 KiB dynamic LDS; call once outside stream capture
// conv_wino_split.hip: the Winograd five-tap kernel on the bf16 pipe (split-bf16 x6); mo = 4 / 2 outputs per tile
hipError_t gt_conv_wino5s_init();
hipError_t gt_launch_conv_wino5s(const ConvGemmArgs& a, int mo, hipStream_t stream);
bool gt_gemm_split_applies(const ConvGemmArgs& a);
hipError_t gt_launch_gemm_split(const ConvGemmArgs& a, hipStream_t stream);
// Highwaynet combine (reference Taco2.py:409-424): z [M, 2S] = [relu-branch | sigmoid-branch] pre-activations,
// out = relu(z_h) * sigmoid(z_t) + x * (1 - sigmoid(z_t));  S % 4 == 0
hipError_t gt_launch_highway(const float* z, const float* x, float* out, int64_t M, int S, hipStream_t stream);

// ---------------------------------------------------------------- attention.hip
struct AttnStepArgs {
    const float* q; int64_t ldq;        // [B, A] projected query
    const float* pm;                    // [B, Tv, A] processed memory (Dense_Value(memory), hoisted)
    const float* v;                     // [A] attention_v
    const float* score_bias;            // [1]
    const float* prev;  int64_t ldprev; // previous alignment rows [B, Tv] (NULL -> one-hot(0), Steps.py:201-206)
    const float* noise; int64_t ldnoise;// N(0,1) [B, Tv] or NULL -> Philox
    float* align; int64_t ldalign;      // out [B, Tv]
    float* ctx; int64_t ldctx;          // out [B, A] (row-major) or blocked buffer when ctx_mt > 0
    int ctx_mt;
    int B, Tv, A, type;                 // type: GSTTACO_ATT_*
    float sigmoid_noise;
    const uint64_t* seed_ptr; uint32_t rng_step;
    const int32_t* tok_len;             // masked-mode extension (A12): positions >= tok_len[b] do not exist, or NULL
    // LSA extension (type == GSTTACO_ATT_LSA): location conv [k,1,F] + bias [F], location dense [F,A] + bias [A],
    // additive bias [A]; state [B,Tv] = cumulative (or last) alignment, read and updated in place
    const float *loc_cw, *loc_cb, *loc_dw, *loc_db, *att_bias;
    float* lsa_state;
    int loc_k, loc_f, lsa_cumulate, lsa_smoothing;
    int rows_lds;                       // rows of pm staged in LDS per chunk
};

hipError_t gt_launch_attn_step(const AttnStepArgs& a, hipStream_t stream);
hipError_t gt_attn_init();     // opt in to >64 KiB dynamic LDS; call once outside stream capture
hipError_t gt_launch_set_seed(uint64_t* dst, uint64_t seed, hipStream_t stream);
hipError_t gt_launch_rng_fill(const uint64_t* seed_ptr, float* masks, float* noise, int steps, int B, int P0, int P1, int Tv,
                              float drop_rate, hipStream_t stream);
hipError_t gt_launch_relayout_masks(const float* src, float* dst, int steps, int B, int p0, int p1, int P0, int P1, int to_padded, hipStream_t stream);
// up to 8 device-to-device copies of 4-byte words + the call's seed in one launch (attention.hip gt_copy_segments_kernel)
struct GtCopySegs { const void* src[8]; void* dst[8]; size_t words[8]; int n; uint64_t seed; uint64_t* seed_dst; };
hipError_t gt_launch_copy_segments(const GtCopySegs& S, hipStream_t stream);
hipError_t gt_launch_embed_rows(const float* table, const int32_t* tokens, float* out, int rows, int C, hipStream_t stream);
hipError_t gt_launch_zero(float* p, size_t n_floats, hipStream_t stream);   // n rounded up to a multiple of 4 floats
size_t gt_attn_lds_bytes(int Tv, int A, int loc_f, int loc_k, int* rows_lds);

// ---------------------------------------------------------------- dec_front.hip
struct DecFrontArgs {
    const float* frame; int64_t ldframe;    // [B, mel] last emitted frame (zero rows, ld 0, at step 0)
    const float* z0;                        // [B, P0] prenet-0 PRE-activations incl. bias, produced with the previous step's
                                            // projection (NULL: compute them here from `frame`)
    const float* w0; const float* b0;       // prenet0 [mel, P0] (TF layout), [P0]
    const float* w1; const float* b1;       // prenet1 [P0, P1], [P1]
    const float* wq; const float* bq;       // attention Query [P1, A], [A]
    const float* mask0; const float* mask1; // keep-masks [B,P0] / [B,P1] or NULL -> Philox
    float drop_rate, drop_scale;
    const uint64_t* seed_ptr; uint32_t rng_step;
    const float* pm;                        // [B, Tv, A] processed memory
    const float* v; const float* score_bias;
    const float* prev; int64_t ldprev;      // NULL -> one-hot(0)
    const float* noise; int64_t ldnoise;    // NULL -> Philox
    float* align; int64_t ldalign;
    float* xa;                              // out, BLOCKED [ (P1+A)/16 ][MT][64][4]: prenet output k in [0,P1), context k in [P1,P1+A)
    int MT;
    int B, Tv, mel, P0, P1, A, type;
    float sigmoid_noise;
    const int32_t* tok_len;    // masked-mode extension (A12): memory positions >= tok_len[b] do not exist, or NULL
    unsigned long long* dbg;   // diagnostic phase stamps of block 0, or NULL
    // Worker workgroups (blockIdx.x >= B) of the same launch: recurrent halves h_{t-1}.W_h + b of the two decode LSTM
    // layers, written as pre-activation partial sums.  They only need the PREVIOUS step's state, so they run on the
    // ~224 CUs the per-utterance front end leaves idle.  n_workers == 0 disables them.
    SkinnyArgs rec[2];
    int rec_begin[2], rec_end[2];   // tile ranges [begin, end) of each layer handled by this launch's workers
    int n_workers;
    // Batches above 32 rows on the lean bodies (dec_front.hip front_worker): whole jobs [0, sched_pf) go round-robin to the pure
    // workers, jobs [sched_pf, sched_pf + sched_ne) are cut into pieces of sched_e chunks for them as well, and the remaining
    // jobs into pieces of sched_y chunks for the first `utt_jobs` utterance workgroups, which take them once their utterance is
    // done (a job = a pair of tiles over every 32-row chunk of the batch; at 128 rows in fp32 it outlasts the chain).
    // Otherwise sched_pf = all jobs, the rest 0.
    int utt_jobs, sched_pf, sched_ne, sched_e, sched_y;
    int worker_tiles;               // tiles per worker job: 1, or 0/2 = pairs sharing one pass over the activations
    LeanPartialArgs lrec[2];        // the same two GEMMs for the lean body (fp32, K = 1024); used when lean_rec != 0
    int lean_rec;                   // 0: general body, 1: lean fp32, 2: lean bf16
    int keep_hash;                  // throughput mode at dropout rate 0.5 and the reference's prenet / attention sizes: rows of
                                    // the prenet-1 / query weights that the (hashed) keep decisions zero are not requested
    int lean_front;                 // 1: the lean utterance path (front_lean.h) where its preconditions hold; 0: the general kernel
    uint16_t* xah;                  // bf16 mirror of xa (gt_blk_off_h), written beside it, or NULL
    // type == GSTTACO_ATT_LSA (dec_front_lsa.hip; the four-kernel path's AttnStepArgs fields of the same names): `prev` is then the
    // state buffer itself (ldprev = Tv), read at the start of the step and rewritten at its end
    const float* loc_pack;                  // the LSA weights as the kernel's LDS image (LsaPack below), 16-byte aligned
    float* lsa_state;                       // [B, Tv] cumulative (or previous) alignment
    int loc_k, loc_f, lsa_cumulate, lsa_smoothing;
};
// LDS image of the LSA weights on the fused front end, built once on the host (gsttaco.cpp) and copied per step with 16-byte loads:
//   dw  [LFp][LDWS]  location Dense kernel [F, A], rows zero-padded to the MFMA k granule (4)
//   cw  [LKp][LCS]   location Conv1D kernel [K, 1, F] as [K][F], taps padded to 4, filters to the MFMA n granule (16)
//   cb  [LFc]        Conv1D bias (zero beyond F)
//   ab  [A]          Dense bias + the additive attention bias
// Row strides LDWS = A + 8 and LCS = LFc + 8 floats put the four k rows of a B fragment on different banks; LFS = LFc + 4 is the row
// stride of the location-feature tile the kernel writes behind the memory tile.
struct LsaPack { int LFc, LFp, LKp, LFS, LDWS, LCS, off_cw, off_cb, off_ab, total; };
inline __host__ __device__ LsaPack gt_lsa_pack(int A, int LF, int LK) {
    LsaPack p;
    p.LFc = (LF + 15) & ~15; p.LFp = (LF + 3) & ~3; p.LKp = (LK + 3) & ~3;
    p.LFS = p.LFc + 4; p.LDWS = A + 8; p.LCS = p.LFc + 8;
    p.off_cw = p.LFp * p.LDWS; p.off_cb = p.off_cw + p.LKp * p.LCS; p.off_ab = p.off_cb + p.LFc;
    p.total = (p.off_ab + A + 3) & ~3;
    return p;
}
// (loc_f > 0: the LSA form, whose operands share the LDS with the processed-memory tile)
bool gt_dec_front_supported(int mel, int P0, int P1, int A, int Tv, int loc_f = 0, int loc_k = 0);
// dec_front_lsa.hip: the LSA instantiations of the general kernel (shape = 0..4 for A = 16, 32, 64, 128, 256)
hipError_t gt_front_lsa_init();
void gt_front_lsa_launch(int shape, bool z0, int lean, bool exact, dim3 grid, size_t lds, hipStream_t s, const DecFrontArgs& a);
hipError_t gt_dec_front_init();
hipError_t gt_launch_dec_front(const DecFrontArgs& a, hipStream_t stream);

// ---------------------------------------------------------------- persist_decode.hip
// The whole decoder loop as ONE persistent launch (batch <= 128, T_v <= 256, fp32, the reference's decoder sizes): see the file.
struct PersistDecodeArgs {
    // weights: the launch path's packs (MFMA-fragment order [tile][k-block][lane][4])
    const float *w1x, *w1h, *b1h, *w2x, *w2h, *b2h;   // lstm_x[l].wp / lstm_h[l].wp / lstm_h[l].bias
    const float *wp, *bp;                              // proj_z: projection columns | padding | fused prenet-0 columns
    int pj_tiles, n_out, n_split, z_col0;              // tiles of proj_z; mel*r + 1; mel*r; first prenet-0 column
    const float *W1, *b1, *Wq, *bq, *av, *score_bias;  // prenet-1 [P0][P1] + bias, query [P1][A] + bias (plain layouts), attention v, score bias
    const float* pm;                                   // processed memory [B][Tv][A]
    const float* noise;                                // N(0,1) [steps][B][Tv] (injected or pre-generated), or NULL when sigmoid_noise == 0
    const float* masks;                                // keep masks [steps][mask0 B*P0 | mask1 B*P1], or NULL (hashed / no dropout)
    const uint64_t* seed_ptr;
    const int32_t* tok_len;                            // masked-mode extension (A12) or NULL
    float drop_rate, drop_scale, sigmoid_noise;
    int keep_hash, att_type;
    // att_type == GSTTACO_ATT_LSA (the one-group kernel, up to 128 tokens): the location weights as the LDS image the fused front end
    // uses (LsaPack above; resident in the chain workgroups' LDS for the whole launch) -- the state lives in their LDS
    const float* loc_pack;
    int loc_f, loc_k, lsa_cumulate, lsa_smoothing;
    // state / workspace
    float* xa[2];                                      // blocked [24][MT][256], ping-pong by step parity
    float* h1[2]; float* h2[2];                        // blocked [64][MT][256]
    uint16_t* xah[2]; uint16_t* h1h[2]; uint16_t* h2h[2];  // mixed precision: the bf16 mirrors (gt_blk_off_h) of xa / h1 / h2, ping-pong by step parity
    int bf16;                                          // 1: the bf16 kernel (weights are the bf16 packs, activations travel as mirrors only)
    uint2* z0g;                                        // [B][256] {value bits, step tag}
    float* hpart;                                      // [2][32][512] (bf16 kernel: [2][64][1024]) recurrent halves of the chain workgroups' tiles
    float* stash;                                      // [256][16][512] the group kernels' chain workgroups park their tile state here during the chain
    uint32_t* ctl;                                     // gt_persist_decode_ctl_words() words, zeroed by the launcher
    uint32_t* err;                                     // host-mapped give-up word of this launch
    // outputs
    float* pre; int64_t ld_pre;                        // [B][steps*r*mel]
    float* stop;                                       // [B][steps]
    float* align; int64_t ld_align;                    // [B][steps][Tv]
    int B, MT, Tv, steps, co_tiles;                    // co_tiles: layer-2 tiles whose recurrent half the launch path sums in 8-wave order
    int G, n_chain, tvp, twopass;                      // filled in by the launcher: groups of rows, chain workgroups, T_v rounded up to 64, and
                                                       // whether the group kernels' chain workgroups sum their recurrent halves through 8 slabs
    int expect_extra;                                  // fault injection (tests): the all-to-all waits expect this many arrivals too many
    unsigned long long* dbg;                           // diagnostic stamps [3 roles][32] or NULL (GSTTACO_STAMPS=1)
};
size_t gt_persist_decode_ctl_words();
bool gt_persist_decode_supported(int mel, int r, int P0, int P1, int A, int H1, int H2, int B, int Tv, int pj_tiles, int pj_nkb, int slots, int split16, int bf16);
int gt_persist_decode_max_batch();
bool gt_persist_decode_lsa_fits(int B, int Tv, int loc_f, int loc_k);       // the LSA chain: one group, <= 128 tokens, operands beside the memory tile
hipError_t gt_persist_decode_init();                   // opt in to > 64 KiB dynamic LDS; once, outside stream capture
int gt_persist_decode_blocks_per_cu();
hipError_t gt_launch_persist_decode(const PersistDecodeArgs& a, const float* b0, int split16, hipStream_t stream);

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
hipError_t gt_gst_init();      // opt in to >64 KiB dynamic LDS; call once outside stream capture

// ---------------------------------------------------------------- audio.hip
struct AudioFrontArgs {
    const float* wav;           // [B, ld_wav] float samples in [-1, 1)
    const int32_t* wav_len;     // [B]
    double* mse;                // [B, ld_mse] workspace: trim frame mean-squares
    int32_t* bounds;            // [B, 2] workspace: trimmed (start, length) in samples
    float* mels;                // [B, cap_frames, n_mels] mels_for_gst layout (frame 0 = zeros)
    int32_t* mel_len;           // [B] frames excluding the prepended one
    const float* window;        // [n_fft] hann (periodic), zero-padded centred to n_fft
    const float2* twiddle;      // [n_fft/2] exp(-2 pi i k / n_fft)
    const float* mel_basis;     // [n_mels, n_fft/2+1]
    const int32_t* band_lo;     // [n_mels] first non-zero bin of each band
    const int32_t* band_hi;     // [n_mels] one past the last non-zero bin
    int B, ld_wav, ld_mse, cap_frames, n_fft, log2_h, hop, n_mels, trim_frame, trim_hop;
    float preemph, trim_gain, top_db, max_abs;
};
hipError_t gt_launch_audio_front(const AudioFrontArgs& a, hipStream_t stream);

struct GriffinLimArgs {
    const float* spec;          // [B, T, n_fft/2+1] normalised spectrogram (the vocoder's output layout)
    const int32_t* frames;      // [B] frames to use per utterance, or NULL (= T)
    const float* init_phase;    // [B, T, n_fft/2+1] uniform [0,1) (x 2 pi), or NULL -> Philox(seed)
    float* mag;                 // [B, T, n_fft/2+1] workspace: S ^ power
    const float* frm_old;       // [B, T, n_fft] windowed inverse-FFT frames of the previous iteration
    float* frm_new;
    float* ybuf;                // [B, ld_y] final overlap-added signal (before de-emphasis)
    float* wav;                 // [B, ld_wav] output
    int32_t* wav_len;           // [B] output: hop * (frames - 1), or NULL
    const float* window;        // [n_fft]
    const double* win_sq;       // [n_fft] window^2 in float64
    const float2* twiddle;      // [n_fft/2]
    int64_t ld_y;
    uint64_t seed;
    int B, T, n_fft, log2_h, hop, ld_wav, iters;
    float power, ref_level_db, max_abs, preemph;
};
hipError_t gt_gl_init();
hipError_t gt_launch_griffin_lim(GriffinLimArgs a, float* frm_a, float* frm_b, hipStream_t stream);

