/*
 * gsttaco.h -- C-ABI of the MI355X-native GST-Tacotron inference hot path.
 *
 * The reference (CODEJIN/GST_Tacotron, TF2/Keras) has no FFI/plugin interface: its
 * boundary is the Python class GST_Tacotron (reference Model.py:37) and
 * Hyper_Parameters.json.  This header is what a binding for that class calls instead
 * of the Keras functional model `model_Dict['Inference']` (reference Model.py:145-156):
 *
 *   gsttaco_create / gsttaco_load_weight / gsttaco_finalize_weights
 *       <- GST_Tacotron.__init__ + Model_Generate + Restore   (Model.py:38-40, 42-189, 267-276)
 *   gsttaco_inference_step
 *       <- GST_Tacotron.Inference_Step                        (Model.py:249-255)
 *   gsttaco_encode      <- Modules/Taco2.py:12-51   Encoder.call
 *   gsttaco_gst         <- Modules/GST.py:91-109    Style_Token_Layer.call
 *                          (== GST_Tacotron.Inference_GST_Step, Model.py:257-265)
 *   gsttaco_decode      <- Modules/Taco2.py:153-228 Decoder.call loop (training=False),
 *                          Decoder_Step :96-120, Prenet :262-283,
 *                          Modules/Attention/Steps.py:107-229 (BMA / SMA)
 *   gsttaco_postnet     <- Modules/Taco2.py:131-149, 230
 *   gsttaco_vocoder     <- Modules/Taco2.py:234-260 Vocoder_Taco1.call, CBHG :285-380
 *   gsttaco_mel_frontend <- Pattern_Generator.py:39-60 Mel_Generate + Audio.py:29-32,49-55,70-96 melspectrogram
 *                          + the batch layout of Feeder.py:204-225 / 229-250
 *   gsttaco_griffin_lim <- Audio.py:23-27 inv_spectrogram, :57-68 _griffin_lim, :74-75 _istft (SURVEY row N4;
 *                          called from Model.py:414-422 Export_Inference)
 *   gsttaco_mel_basis   <- Audio.py:81-83 _build_mel_basis (librosa.filters.mel); host-side getter for tests
 *
 * Conventions
 *   - every function returns 0 on success or a negative GSTTACO_E_* code; nothing throws
 *     across the ABI; gsttaco_last_error() returns a message for the last failure.
 *   - tensor arguments are DEVICE pointers (HIP) owned by the caller, dense row-major,
 *     innermost dimension contiguous, float32 unless stated; weights are HOST pointers.
 *   - all work is enqueued on the given hipStream_t (passed as void*); no implicit sync.
 *   - one ctx per device; a ctx is not thread-safe; distinct ctxs are independent.
 *   - there is NO CPU fallback: without a gfx950 device every compute call fails with
 *     GSTTACO_E_NO_DEVICE.
 */
#ifndef GSTTACO_H
#define GSTTACO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSTTACO_ABI_VERSION 12
#define GSTTACO_MAX_LAYERS 8

enum {
    GSTTACO_OK = 0,
    GSTTACO_E_INVALID = -1,     /* bad argument / unsupported configuration */
    GSTTACO_E_NO_DEVICE = -2,   /* no HIP device (the product path has no CPU fallback) */
    GSTTACO_E_HIP = -3,         /* a HIP runtime call failed */
    GSTTACO_E_WEIGHTS = -4,     /* missing / mis-shaped weight, or weights not finalized */
    GSTTACO_E_CAPACITY = -5     /* batch / tokens / frames exceed the capacity given at create */
};

/* BMA / SMA: reference Taco2.py:66-75.  LSA: EXTENSION -- step-wise restatement of LocationSensitiveAttention
 * (reference Modules/Attention/Layers.py:289-444), which the reference decoder cannot select. */
enum { GSTTACO_ATT_BMA = 0, GSTTACO_ATT_SMA = 1, GSTTACO_ATT_LSA = 2 };

/* Mirrors the hot-path keys of Hyper_Parameters.json (reference file of that name). */
typedef struct gsttaco_config {
    int32_t abi_version;        /* GSTTACO_ABI_VERSION */
    int32_t device;             /* HIP device ordinal */
    /* Sound / loop */
    int32_t mel_dim;            /* Sound.Mel_Dim */
    int32_t step_reduction;     /* Step_Reduction */
    int32_t max_step;           /* Max_Step (decoder runs Max_Step // Step_Reduction iterations, Taco2.py:213) */
    /* Tacotron2.Encoder */
    int32_t vocab;              /* len(Token_Index_Dict) */
    int32_t emb;                /* Embedding.Size */
    int32_t n_enc_conv;
    int32_t enc_filters[GSTTACO_MAX_LAYERS];
    int32_t enc_kernels[GSTTACO_MAX_LAYERS];
    int32_t enc_rnn;            /* RNN.Size (per direction) */
    /* Tacotron2.Decoder */
    int32_t n_prenet;           /* must be 2 */
    int32_t prenet[GSTTACO_MAX_LAYERS];
    float   prenet_rate;        /* Prenet.Dropout_Rate -- live at inference (Taco2.py:283) */
    int32_t n_dec_rnn;          /* must be 2 */
    int32_t dec_rnn[GSTTACO_MAX_LAYERS];
    int32_t att_type;           /* GSTTACO_ATT_* */
    int32_t att_size;           /* Attention.Size */
    float   sigmoid_noise;      /* SMA 2.0 (Steps.py:212), BMA 0.0 (Steps.py:58) */
    int32_t loc_filters;        /* LSA only: Attention.Conv.Filters      (Layers.py:314-319) */
    int32_t loc_kernel;         /* LSA only: Attention.Conv.Kernel_Size */
    int32_t lsa_cumulate;       /* LSA only: cumulate_weights (Layers.py:302, default 1) */
    int32_t lsa_smoothing;      /* LSA only: smoothing normalisation instead of softmax (Layers.py:300, 426-444) */
    int32_t n_post;             /* len(Decoder.Conv.Filters)+1 */
    int32_t post_filters[GSTTACO_MAX_LAYERS];
    int32_t post_kernels[GSTTACO_MAX_LAYERS];
    int32_t post_tanh;          /* number of leading postnet layers followed by tanh (Taco2.py:145) */
    /* GST */
    int32_t gst_use;
    int32_t n_ref_conv;
    int32_t ref_filters[GSTTACO_MAX_LAYERS];
    int32_t ref_kernels[GSTTACO_MAX_LAYERS];
    int32_t ref_strides[GSTTACO_MAX_LAYERS];
    int32_t ref_rnn;            /* Reference_Encoder.RNN.Size */
    int32_t ref_dense;          /* Reference_Encoder.Dense.Size */
    int32_t n_tokens;           /* Style_Token.Size */
    int32_t token_emb;          /* Style_Token.Embedding.Size */
    int32_t heads;              /* Style_Token.Attention.Head */
    int32_t gst_att;            /* Style_Token.Attention.Size */
    /* Vocoder_Taco1 (CBHG mel -> linear spectrogram, reference Taco2.py:234-260, 285-424); optional */
    int32_t voc_use;            /* 0: no vocoder weights / entry point */
    int32_t spec_dim;           /* Sound.Spectrogram_Dim */
    int32_t bank_count;         /* CBHG.Conv_Bank.Stack_Count (kernel sizes 1..count) */
    int32_t bank_filters;       /* CBHG.Conv_Bank.Filters */
    int32_t n_voc_proj;
    int32_t voc_proj_filters[GSTTACO_MAX_LAYERS];
    int32_t voc_proj_kernels[GSTTACO_MAX_LAYERS];
    int32_t highway_count;
    int32_t highway_size;
    int32_t voc_rnn;            /* CBHG.RNN.Size (per direction) */
    /* Sound: audio front end (wav -> mels_for_gst) and back end (spectrogram -> wav, row N4); optional */
    int32_t sample_rate;        /* Sound.Sample_Rate */
    int32_t frame_length;       /* Sound.Frame_Length (STFT window, <= n_fft = 2 * (Spectrogram_Dim - 1)) */
    int32_t frame_shift;        /* Sound.Frame_Shift (hop) */
    float   max_abs_mel;        /* Sound.Max_Abs_Mel; 0 = the [0,1] normalisation (Audio.py:92-93) */
    int32_t max_wav_samples;    /* capacity of the audio entry points per utterance; 0 = disabled */
    /* Use_Mixed_Precision (reference Model.py:31-35 selects Keras mixed_float16; here: bf16 GEMM operands, fp32 accumulation,
     * fp32 state / epilogues / outputs -- BASELINE configs[4]).  0 = everything fp32 (the parity path). */
    int32_t mixed_precision;
    /* capacity: workspace is sized once, at finalize */
    int32_t max_batch;
    int32_t max_tokens;
    int32_t max_ref_frames;     /* frames of mels_for_gst INCLUDING the prepended zero frame */
} gsttaco_config;

typedef struct gsttaco_ctx gsttaco_ctx;

int gsttaco_abi_version(void);

/* Validates cfg and creates a context.  Does not touch the GPU (so the host logic is
 * testable without one); the device is first used by gsttaco_finalize_weights. */
int gsttaco_create(const gsttaco_config* cfg, gsttaco_ctx** out);
void gsttaco_destroy(gsttaco_ctx* ctx);
const char* gsttaco_last_error(const gsttaco_ctx* ctx);   /* ctx may be NULL: last create error */

/* Weight manifest (names/shapes in the TF variable layouts). */
int gsttaco_num_weights(const gsttaco_ctx* ctx);
int gsttaco_weight_info(const gsttaco_ctx* ctx, int index, const char** name, int64_t shape[4], int* ndim);
/* Copies one tensor (HOST float32 pointer) into the context. */
int gsttaco_load_weight(gsttaco_ctx* ctx, const char* name, const float* host_data,
                        const int64_t* shape, int ndim);
/* BN folding, MFMA-fragment repack, upload to HBM, workspace allocation. */
int gsttaco_finalize_weights(gsttaco_ctx* ctx);

/* Masked mode (EXTENSION): the reference has no padding masks -- `token_lengths` is accepted by
 * Inference_Step and ignored (Model.py:249-253).  token_lengths == NULL reproduces that.  With a [B] int32
 * device array the padded positions t >= token_lengths[b] are treated as non-existent (encoder convs see zeros there,
 * the BiLSTM and the attention cover [0, length) only), so every utterance of a ragged batch equals that utterance
 * run alone; encoder rows / alignment columns beyond the length are written as 0. */

/* tokens [B,Tv] int32  ->  enc [B,Tv,2*enc_rnn] */
int gsttaco_encode(gsttaco_ctx* ctx, const int32_t* tokens, const int32_t* token_lengths, int B, int Tv, float* enc,
                   void* stream);

/* mels_for_gst [B,Tref1,mel] (frame 0 = the prepended zero frame, dropped inside as GST.py:98 does),
 * mel_lengths [B] int32 (excluding that frame)  ->  gst [B,gst_att] */
int gsttaco_gst(gsttaco_ctx* ctx, const float* mels_for_gst, const int32_t* mel_lengths,
                int B, int Tref1, float* gst, void* stream);

/* enc [B,Tv,2*enc_rnn], gst [B,gst_att] (NULL when GST is off)
 * prenet_mask: keep-masks [steps,2,B,prenet] float32 (1 keep / 0 drop), or NULL = generated on the device from `seed`:
 *              at the reference's rate 0.5 the keep bit of (step, layer, utterance row, column) is bit (column & 31) of a
 *              32-bit COUNTER HASH of (seed, step, layer, row, column >> 5) -- three murmur3 finalisers, device_utils.h
 *              gt_keep_word; NOT Philox: the scalar unit derives a wave's 16 decisions in ~25 instructions, which is what
 *              lets the front kernel skip the weight rows that meet an exact zero -- and at any other rate
 *              Philox4x32-10 (u01 > rate).  The reference draws tf.nn.dropout's unseeded stream (Taco2.py:283): only
 *              the distribution can match, checked by tests/test_gpu_parity.py::test_hashed_keep_decisions_look_random
 * attn_noise : N(0,1) samples [steps,B,Tv], or NULL = Philox4x32-10 + Box-Muller from `seed`
 *              (gsttaco_debug_randomness reads back what a call used)
 * steps      : 0 = Max_Step // Step_Reduction, else 1..that
 * outputs    : pre_mel [B,steps*r,mel], stop [B,steps], align [B,steps,Tv] */
int gsttaco_decode(gsttaco_ctx* ctx, const float* enc, const float* gst, const int32_t* token_lengths,
                   const float* prenet_mask, const float* attn_noise, uint64_t seed,
                   int B, int Tv, int steps, float* pre_mel, float* stop, float* align, void* stream);

/* pre_mel [B,T,mel] -> mel [B,T,mel] (5 x Conv1D+BN, tanh on the first post_tanh layers, + residual) */
int gsttaco_postnet(gsttaco_ctx* ctx, const float* pre_mel, int B, int T, float* mel, void* stream);

/* mel [B,T,mel] (the post-net mel) -> spectrogram [B,T,spec_dim]: Vocoder_Taco1 = CBHG (conv bank k=1..N + BN + ReLU,
 * max-pool 2/1, two projection convs, residual, highway stack, BiLSTM) + Dense (reference Taco2.py:234-260, 285-424). */
int gsttaco_vocoder(gsttaco_ctx* ctx, const float* mel, int B, int T, float* spectrogram, void* stream);

/* wav -> mels_for_gst without leaving the GPU and without needing weights (works before finalize).
 * Needs cfg.max_wav_samples > 0.  Reference: Mel_Generate(path, top_db, range_Ignore=True) per utterance
 * (Pattern_Generator.py:39-60; top_db is 60 for one reference wav and 15 for several, Feeder.py:204-209, and 60 in
 * Get_Inference_GST_Pattern, Feeder.py:232), stacked as Feeder does: frame 0 of every utterance is zero, frames
 * 1..mel_lengths[b] hold the mel, the rest is zero padding.
 * wav          : [B, ld_wav] float32 samples in [-1,1) at Sound.Sample_Rate (librosa.load's output; resampling is the
 *                caller's job), wav_lengths [B] valid samples (17 <= length <= max_wav_samples)
 * mels_for_gst : [B, cap_frames, mel_dim] with cap_frames >= 2 + ld_wav / frame_shift
 * mel_lengths  : [B] int32, frames EXCLUDING the prepended one; 0 when the trimmed signal is shorter than n_fft/2+1
 *                samples (librosa.stft raises there) */
int gsttaco_mel_frontend(gsttaco_ctx* ctx, const float* wav, const int32_t* wav_lengths, int B, int ld_wav, float top_db,
                         float* mels_for_gst, int32_t* mel_lengths, int cap_frames, void* stream);

/* spectrogram -> waveform: Audio.inv_spectrogram (reference Audio.py:23-27) = symmetric de-normalisation (max_abs_mel,
 * or the [0,1] one when it is 0), + ref_level_db, dB -> amplitude, ^power, `iters` Griffin-Lim iterations
 * (Audio.py:57-68; librosa stft / istft, hann, centred), inverse pre-emphasis 0.97.  Needs cfg.max_wav_samples > 0.
 * spectrogram : [B, T, spec_dim], the layout gsttaco_vocoder / gsttaco_inference_step write (time-major; the reference
 *               transposes it before the call, Model.py:415)
 * frames      : [B] int32 frames to use per utterance (Model.py:415: max(1, stop index) * Step_Reduction) or NULL = T
 * init_phase  : [B, T, spec_dim] uniform [0,1) numbers (the reference draws np.random.rand unseeded, Audio.py:61) or
 *               NULL = Philox4x32-10 from `seed`
 * wav         : [B, ld_wav] float32, ld_wav >= frame_shift * (T - 1); samples beyond an utterance's length are zero
 * wav_lengths : [B] int32 out (may be NULL): frame_shift * (frames - 1); 0 when that is <= n_fft/2 (librosa raises) */
int gsttaco_griffin_lim(gsttaco_ctx* ctx, const float* spectrogram, const int32_t* frames, int B, int T, int iters,
                        float power, float ref_level_db, const float* init_phase, uint64_t seed,
                        float* wav, int32_t* wav_lengths, int ld_wav, void* stream);

/* CRC-32C of a host buffer continued from `crc` (0 to start): the checksum of TensorFlow checkpoint bundles, used by
 * gst_tacotron_amd/tf_checkpoint.py for the reference's tf.train.Checkpoint files (Model.py:186-189, 267-276).  Host only. */
uint32_t gsttaco_crc32c(const void* data, size_t n, uint32_t crc);

/* host_out [mel_dim, spec_dim] float32 <- the Slaney mel filterbank the front end uses (librosa.filters.mel defaults) */
int gsttaco_mel_basis(gsttaco_ctx* ctx, float* host_out);

/* The whole Inference_Step (Model.py:249-255): encoder, GST, decode loop, postnet and -- when `spectrogram` is not
 * NULL -- the CBHG vocoder, replayed from one cached hipGraph per (B,Tv,Tref1,steps) shape.
 * mels_for_gst / mel_lengths are ignored (may be NULL) when GST is off; pre_mel and spectrogram may be NULL. */
int gsttaco_inference_step(gsttaco_ctx* ctx, const int32_t* tokens, const int32_t* token_lengths,
                           const float* mels_for_gst, const int32_t* mel_lengths,
                           const float* prenet_mask, const float* attn_noise, uint64_t seed,
                           int B, int Tv, int Tref1, int steps,
                           float* mel, float* stop, float* align, float* pre_mel, float* spectrogram, void* stream);

/* hipGraph cache policy.  Every entry point replays one cached graph executable per (entry, B, Tv, Tref1, steps, flags) key.
 * The cache is LRU-bounded to `max_cached` executables (default 16 -- an Inference_Step replays two or three: encoder segment, GST + decode + postnet, vocoder; GSTTACO_GRAPH_CACHE; 0 = no graphs, everything is
 * enqueued eagerly on the caller's stream); the least recently used one is destroyed when a new shape is captured.
 * `capture_after` = n >= 1: a key is captured at its n-th use and enqueued eagerly before that (default 1;
 * GSTTACO_GRAPH_CAPTURE_AFTER).  Callers whose shapes vary from batch to batch -- the reference's Feeder pads to the
 * batch maximum (Feeder.py:175-180) -- should use 2, or bucket shapes with masked mode, so that a shape that never
 * repeats never pays a ~2 000-node capture.  Results are identical either way (a GPU test compares them bitwise). */
int gsttaco_set_graph_policy(gsttaco_ctx* ctx, int max_cached, int capture_after);
int gsttaco_graph_cache_size(const gsttaco_ctx* ctx);

/* Measurement support (bench.py): per-kernel timing of the last gsttaco_inference_step replay.
 * When enabled, HIP event-record nodes bracket the four kernels of every `every`-th decode step inside the graph. */
int gsttaco_set_profiling(gsttaco_ctx* ctx, int every);
/* After the stream has been synchronised by the caller: average duration (ms) of the bracketed
 * launches of decode-LSTM layer `layer` (0/1) and how many were bracketed. */
int gsttaco_get_profile(gsttaco_ctx* ctx, int layer, float* avg_ms, int* count);
/* Diagnostic (GSTTACO_STAMPS=1): 96 words of phase stamps (100 MHz ticks) at the middle decode step of the last replay.  Launch
 * path: the first 3 x 16 = workgroup 0 of the fused front kernel and of the two decode LSTM kernels; persistent decode launch:
 * 3 x 32 = its chain workgroup 0, projection workgroup 32 and plain workgroup 255 (tools/stamps.py, tools/stamps_persist.py).
 * Synchronises the device. */
int gsttaco_debug_stamps(gsttaco_ctx* ctx, unsigned long long* host_out96);
/* Three launches hand data between their workgroups INSIDE the kernel and therefore need those workgroups resident together: the
 * persistent BiLSTM launch (one launch for all time steps of the encoder's / vocoder's Bidirectional LSTM, reference
 * Taco2.py:39-43, 394-398; the 32 workgroups of each of its groups), the persistent decode launch (the whole decoder loop,
 * Taco2.py:153-228, in one launch of 256 workgroups) and the fused decode-LSTM launch (both LSTMCells of a decoder step,
 * Taco2.py:77-85,111, in one launch; all of its workgroups).  The library arranges that for everything it controls:
 * exactly one persistent workgroup per compute unit, and the fused launches' whole grid against occupancy x compute units, are
 * checked at finalize; the persistent launches of ALL contexts of the process are chained on the GPU, so two of them never split
 * an XCD; the fused / persistent decode launches are taken only while the process has ONE live context, and another context's
 * segments start behind those still in flight (one recorded event per device).  What the library cannot see -- another process on
 * the GPU, a CU mask -- is caught by BOUNDED waits: a wait that gives up raises a word in host-mapped memory, the whole launch
 * drains at once, and
 *   - gsttaco_synchronize(ctx, stream) synchronises the stream and returns GSTTACO_E_HIP if that happened since the last check:
 *     the outputs of the calls since then are invalid, repeat them.  It is the ONLY place that clears the condition: calls
 *     enqueued behind the one that gave up do not erase it (they notice the word when they are enqueued, keep it in a sticky
 *     per-context flag, and already run the next launch form down -- and are correct: each launch form has a give-up word of its
 *     own, so a give-up of the persistent decode launch does not make the fused LSTM launches enqueued behind it leave early);
 *   - the next compute call that notices the word switches the context to the launch-per-step / two-launch form (bitwise the
 *     same results in fp32; under Use_Mixed_Precision the per-step BiLSTM kernel sums in a different order: within the mixed
 *     tolerance), succeeds, and leaves a "warning: ..." text in gsttaco_last_error.
 * Nothing stays poisoned and nothing hangs.  gsttaco_debug_handoff_error synchronises the device and returns what is pending,
 * i.e. raised and not yet reported by gsttaco_synchronize (bit 0: fused decode-LSTM launch, bit 8: persistent BiLSTM, bit 16:
 * persistent decode launch; 0 = clear). */
int gsttaco_synchronize(gsttaco_ctx* ctx, void* stream);
int gsttaco_debug_handoff_error(gsttaco_ctx* ctx, uint32_t* host_out);
/* Test support: out[0] = persistent BiLSTM launches this context has enqueued (eagerly or into a captured graph), out[1] = 1
 * while the context uses the persistent launch, 0 once it has fallen back to one launch per time step; out[2] / out[3] the same
 * for the persistent DECODE launch (the whole decoder loop of Taco2.py:153-228 as one launch: batch <= 128 -- above 32 rows as
 * groups of 32 through one set of resident weights --, T_v <= 256, fp32, decoder sizes up to the reference's (smaller ones are
 * zero-padded to them at finalize: exact, GSTTACO_PAD_DECODER), one live context;
 * GSTTACO_PERSIST_DECODE=0, GSTTACO_PERSIST_ROWS=<max batch> or a give-up: launches per step, bitwise the same). */
int gsttaco_debug_counters(const gsttaco_ctx* ctx, uint64_t out[4]);
/* Test support (fault injection).  bits 0..6 / 7 / 8..15: raise the fused launch's / the persistent decode launch's / the persistent
 * BiLSTM's give-up word as a kernel would.  bits 16..: n > 0 makes member n - 1 of every group of the NEXT persistent launches exit at once and the next fused
 * launches expect one arrival too many, so that their waits really run into the bound. */
int gsttaco_debug_raise_handoff_error(gsttaco_ctx* ctx, uint32_t bits);
/* Test support: the prenet keep-masks [steps][mask0 B*P0 | mask1 B*P1] (0/1) and SMA noise [steps][B][Tv] the LAST
 * gsttaco_inference_step / gsttaco_decode of that shape used -- generated from the seed in throughput mode, or the
 * injected tensors -- copied to HOST buffers (either may be NULL).  Synchronises the device. */
int gsttaco_debug_randomness(gsttaco_ctx* ctx, float* host_masks, float* host_noise, int steps, int B, int Tv);
/* Which variant of the decode step a (Tv)-token batch runs on (finalized context).  plan[0]: 1 = fused per-utterance
 * front kernel (prenet + query + attention), 0 = the four-kernel front end (LSA, or a shape the fused kernel does not
 * cover); plan[1]: 1 = prenet layer 0's pre-activations ride in the previous step's projection launch (a composed
 * weight matrix: under Use_Mixed_Precision that matrix is what gets rounded to bf16, which the parity oracle must know);
 * plan[2]: 1 = lean compile-time-K kernels for the LSTM / projection launches, 0 = general skinny GEMM. */
int gsttaco_decode_plan(const gsttaco_ctx* ctx, int Tv, int32_t plan[3]);
/* Algorithmic bytes one launch of decode-LSTM layer `layer` moves at batch B (weights + activations). */
int64_t gsttaco_lstm_launch_bytes(const gsttaco_ctx* ctx, int layer, int B);

#ifdef __cplusplus
}
#endif
#endif /* GSTTACO_H */
