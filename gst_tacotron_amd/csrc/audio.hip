// Audio front end and back end (row N4) on the GPU.
//
// Front end = reference Pattern_Generator.Mel_Generate (Pattern_Generator.py:39-60) + Audio.melspectrogram
// (Audio.py:29-32, 49-55, 70-96) + the batch layout of Feeder.Get_Inference_Pattern (Feeder.py:204-225):
//     wav -> pre-emphasis -> librosa.effects.trim(top_db, frame 32 / hop 16) x 0.99 -> inverse pre-emphasis
//         -> [pre-emphasis -> STFT(hann, centred, reflect) -> |.| -> Slaney mel -> 20 log10 max(1e-5, .) -> normalise]
// The inverse pre-emphasis (an IIR started at the trimmed signal's first sample) followed by the pre-emphasis inside
// _magnitude (a FIR started at the same sample) is the identity in exact arithmetic (the reference does both in
// float64), so the kernels transform 0.99 * trimmed pre-emphasised signal directly: no sequential pass at all.
//   gt_preemph_rms_kernel   frame mean-squares of the pre-emphasised signal, float64 in scipy.lfilter's operation
//                           order so that the trim decision (a threshold on a ratio) cannot flip on rounding
//   gt_trim_bounds_kernel   per utterance: max, first / last frame above -top_db -> sample bounds, mel frame count
//   gt_stft_mel_kernel      one workgroup per output frame: gather (reflect padding) * window -> n_fft-point real FFT
//                           as an n_fft/2 complex radix-2 FFT in LDS -> magnitudes -> mel bands -> dB -> normalise,
//                           written straight into the mels_for_gst batch layout (zero frame 0, zero padding)
// Back end = Audio.inv_spectrogram (Audio.py:23-27, 57-68, 74-75): see the second half of this file.
//
// All of this is a few MFLOP per utterance; the kernels are written for coalesced accesses and zero host round trips,
// not for a roofline.
#include <algorithm>

#include "device_utils.h"
#include "kernels.h"

// ---------------------------------------------------------------------------------------------- shared FFT helper
// In-place radix-2 DIT FFT of H complex points held in LDS (input already in bit-reversed order).
// tw[k] = exp(-2 pi i k / (2H)), k in [0, H): the real-FFT size is N = 2H.  sign = -1 forward, +1 inverse (unscaled).
__device__ __forceinline__ void gt_fft_lds(float2* z, int H, const float2* __restrict__ tw, float sign) {
    for (int half = 1; half < H; half <<= 1) {
        const int tstep = H / half;                     // W_{2 half}^j = W_{2H}^{j * H / half}
        for (int t = threadIdx.x; t < (H >> 1); t += blockDim.x) {
            const int j = t & (half - 1);
            const int i0 = ((t - j) << 1) + j, i1 = i0 + half;
            float2 w = tw[j * tstep];
            w.y *= -sign;                               // table holds the forward twiddle (negative angle)
            const float2 a = z[i0], b = z[i1];
            const float2 bw = make_float2(b.x * w.x - b.y * w.y, b.x * w.y + b.y * w.x);
            z[i0] = make_float2(a.x + bw.x, a.y + bw.y);
            z[i1] = make_float2(a.x - bw.x, a.y - bw.y);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ int gt_bitrev(int v, int bits) { return (int)(__brev((unsigned)v) >> (32 - bits)); }

__device__ __forceinline__ int gt_reflect(int u, int n) {      // np.pad(mode='reflect') index, valid for |pad| < n
    if (u < 0) u = -u;
    if (u >= n) u = 2 * (n - 1) - u;
    return u;
}

// pre-emphasised sample n of an utterance in float64, scipy.signal.lfilter([1,-c],[1]) operation order
__device__ __forceinline__ double gt_preemph_d(const float* x, int n, double negc) {
    const double prev = n > 0 ? (double)x[n - 1] : 0.0;
    return __dadd_rn((double)x[n], __dmul_rn(negc, prev));
}

// ---------------------------------------------------------------------------------------------- N2: front end
__global__ __launch_bounds__(256) void gt_preemph_rms_kernel(AudioFrontArgs P) {
    const int b = blockIdx.y;
    const int len = min(P.wav_len[b], P.ld_wav);
    const int nfr = len >= P.trim_frame / 2 + 1 ? 1 + len / P.trim_hop : 0;
    const float* x = P.wav + (size_t)b * P.ld_wav;
    double* out = P.mse + (size_t)b * P.ld_mse;
    for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < nfr; f += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int i = 0; i < P.trim_frame; ++i) {
            const int u = gt_reflect(f * P.trim_hop + i - P.trim_frame / 2, len);
            const double p = gt_preemph_d(x, u, -(double)P.preemph);
            s = __dadd_rn(s, __dmul_rn(fabs(p), fabs(p)));
        }
        const double r = sqrt(s / (double)P.trim_frame);      // librosa.feature.rms, then effects.trim squares it again
        out[f] = __dmul_rn(r, r);
    }
}

__global__ __launch_bounds__(256) void gt_trim_bounds_kernel(AudioFrontArgs P) {
    __shared__ double smax[256];
    __shared__ int sfirst[256], slast[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int len = min(P.wav_len[b], P.ld_wav);
    const int nfr = len >= P.trim_frame / 2 + 1 ? 1 + len / P.trim_hop : 0;
    const double* mse = P.mse + (size_t)b * P.ld_mse;
    double m = 0.0;
    for (int f = tid; f < nfr; f += 256) m = fmax(m, mse[f]);
    smax[tid] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) smax[tid] = fmax(smax[tid], smax[tid + s]);
        __syncthreads();
    }
    const double ref_db = 10.0 * log10(fmax(1e-10, smax[0]));           // power_to_db(ref=np.max, amin=1e-10, top_db=None)
    int first = INT_MAX, last = -1;
    for (int f = tid; f < nfr; f += 256) {
        const double db = 10.0 * log10(fmax(1e-10, mse[f])) - ref_db;
        if (db > -(double)P.top_db) { first = min(first, f); last = max(last, f); }
    }
    sfirst[tid] = first; slast[tid] = last;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { sfirst[tid] = min(sfirst[tid], sfirst[tid + s]); slast[tid] = max(slast[tid], slast[tid + s]); }
        __syncthreads();
    }
    if (tid == 0) {
        int start = 0, end = 0;
        if (slast[0] >= 0) { start = sfirst[0] * P.trim_hop; end = min(len, (slast[0] + 1) * P.trim_hop); }
        const int tlen = end - start;
        // librosa.stft pads n_fft/2 by reflection: a shorter signal raises in the reference -> 0 frames here
        int nmel = tlen > P.n_fft / 2 ? 1 + tlen / P.hop : 0;
        if (nmel > P.cap_frames - 1) nmel = P.cap_frames - 1;
        P.bounds[2 * b] = start;
        P.bounds[2 * b + 1] = tlen;
        P.mel_len[b] = nmel;
    }
}

__global__ __launch_bounds__(256) void gt_stft_mel_kernel(AudioFrontArgs P) {
    extern __shared__ __attribute__((aligned(16))) float2 zsm[];      // [H] complex, then mag[H+1] floats
    const int N = P.n_fft, H = N >> 1, bits = P.log2_h;
    float* mag = reinterpret_cast<float*>(zsm + H);
    const int b = blockIdx.y, f = blockIdx.x, tid = threadIdx.x;
    float* out = P.mels + ((size_t)b * P.cap_frames + f) * P.n_mels;
    const int nmel = P.mel_len[b];
    if (f == 0 || f - 1 >= nmel) {                                    // prepended zero frame / zero padding
        for (int m = tid; m < P.n_mels; m += blockDim.x) out[m] = 0.f;
        return;
    }
    const int j = f - 1;
    const int start = P.bounds[2 * b], tlen = P.bounds[2 * b + 1];
    const float* x = P.wav + (size_t)b * P.ld_wav + start;
    for (int n = tid; n < H; n += blockDim.x) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 2 * n + e;
            const int u = gt_reflect(j * P.hop + i - H, tlen);
            const float prev = (start + u) > 0 ? x[u - 1] : 0.f;
            v[e] = (x[u] - P.preemph * prev) * P.trim_gain * P.window[i];
        }
        zsm[gt_bitrev(n, bits)] = make_float2(v[0], v[1]);
    }
    __syncthreads();
    gt_fft_lds(zsm, H, P.twiddle, -1.f);
    // untangle the packed real FFT: X[k] = E[k] + W_N^k O[k], E = (Z[k] + conj Z[H-k]) / 2, O = (Z[k] - conj Z[H-k]) / 2i
    for (int k = tid; k <= H; k += blockDim.x) {
        const float2 a = zsm[k & (H - 1)], c = zsm[(H - k) & (H - 1)];
        const float er = 0.5f * (a.x + c.x), ei = 0.5f * (a.y - c.y);
        const float orr = 0.5f * (a.y + c.y), oi = -0.5f * (a.x - c.x);
        float2 w = k < H ? P.twiddle[k] : make_float2(-1.f, 0.f);
        const float xr = er + (orr * w.x - oi * w.y), xi = ei + (orr * w.y + oi * w.x);
        mag[k] = sqrtf(xr * xr + xi * xi);
    }
    __syncthreads();
    for (int m = tid; m < P.n_mels; m += blockDim.x) {
        const float* wrow = P.mel_basis + (size_t)m * (H + 1);
        float acc = 0.f;
        for (int k = P.band_lo[m]; k < P.band_hi[m]; ++k) acc += wrow[k] * mag[k];
        const float db = 20.f * log10f(fmaxf(1e-5f, acc));                                  // Audio.py:86-87
        float v;
        if (P.max_abs > 0.f) v = fminf(fmaxf(2.f * P.max_abs * ((db + 100.f) / 100.f) - P.max_abs, -P.max_abs), P.max_abs);   // :95-96
        else v = fminf(fmaxf((db + 100.f) / 100.f, 0.f), 1.f);                              // :92-93
        out[m] = v;
    }
}

hipError_t gt_launch_audio_front(const AudioFrontArgs& a, hipStream_t s) {
    const int max_rms = a.ld_wav / a.trim_hop + 1;
    hipLaunchKernelGGL(gt_preemph_rms_kernel, dim3((max_rms + 255) / 256, a.B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(gt_trim_bounds_kernel, dim3(a.B), dim3(256), 0, s, a);
    const int H = a.n_fft / 2;
    const size_t lds = (size_t)H * sizeof(float2) + (size_t)(H + 1) * sizeof(float);
    hipLaunchKernelGGL(gt_stft_mel_kernel, dim3(a.cap_frames, a.B), dim3(256), lds, s, a);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- N4: back end
// Audio.inv_spectrogram (reference Audio.py:23-27): denormalise -> dB -> amplitude ^ power -> Griffin-Lim (:57-68) ->
// inverse pre-emphasis.  Griffin-Lim iteration i:  y = istft(S * angles);  angles = exp(1j * angle(stft(y))).
//   gt_gl_prepare_kernel   S^power per bin (zero beyond the utterance's frame count)
//   gt_gl_frames_kernel    one workgroup per STFT frame.  <INIT>: angles = exp(2 pi i u), u injected or Philox.
//                          otherwise the frame of stft(y) is rebuilt on the fly from the previous iteration's windowed
//                          inverse-FFT frames (overlap-add of the <= n_fft/hop frames covering each sample, divided by
//                          the window sum-square, centred trim, reflect padding -- librosa.istft + librosa.stft fused),
//                          forward real FFT, X/|X|, times S, inverse real FFT, window -> this iteration's frame.
//                          One launch per iteration, ping-pong frame buffers, no signal buffer in between.
//   gt_gl_ola_kernel       final overlap-add -> y
//   gt_gl_deemph_kernel    inverse pre-emphasis y[n] = x[n] + c y[n-1] (float64): each thread owns a chunk and warms its
//                          state up over the 2048 samples before it (c^2048 ~ 1e-27: exact to double precision)
__global__ __launch_bounds__(256) void gt_gl_prepare_kernel(GriffinLimArgs P) {
    const int nb = P.n_fft / 2 + 1;
    const int64_t total = (int64_t)P.B * P.T * nb;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)((i / nb) % P.T), b = (int)(i / ((int64_t)nb * P.T));
        const int Tb = P.frames ? min(P.frames[b], P.T) : P.T;
        float v = P.spec[i], db;
        if (P.max_abs > 0.f) db = (fminf(fmaxf(v, -P.max_abs), P.max_abs) + P.max_abs) / (2.f * P.max_abs) * 100.f - 100.f;   // Audio.py:101-102
        else db = fminf(fmaxf(v, 0.f), 1.f) * 100.f - 100.f;                                                             // :98-99
        // (10 ^ ((db + ref) * 0.05)) ^ power                                                                              :26-27, :89-90
        P.mag[i] = t < Tb ? expf(P.power * 0.11512925464970229f * (db + P.ref_level_db)) : 0.f;
    }
}

__device__ __forceinline__ int gt_gl_frames_of(const GriffinLimArgs& P, int b) {
    const int Tb = P.frames ? min(P.frames[b], P.T) : P.T;
    return P.hop * (Tb - 1) > P.n_fft / 2 ? Tb : 0;        // librosa.stft's reflect padding needs len(y) > n_fft/2
}

// sample p of the un-trimmed overlap-added signal: sum of the covering frames / window sum-square (librosa.istft)
__device__ __forceinline__ float gt_gl_ola(const GriffinLimArgs& P, const float* frm, int p, int Tb) {
    const int N = P.n_fft;
    const int t_hi = min(Tb - 1, p / P.hop);
    const int t_lo = p >= N ? (p - N) / P.hop + 1 : 0;
    float acc = 0.f, wss = 0.f;
    for (int tt = t_lo; tt <= t_hi; ++tt) {
        const int off = p - tt * P.hop;
        acc += frm[(size_t)tt * N + off];
        wss = (float)((double)wss + P.win_sq[off]);      // float32 accumulator, float64 addend (filters.window_sumsquare)
    }
    return wss > 1.17549435e-38f ? acc / wss : acc;
}

template <bool INIT>
__global__ __launch_bounds__(256) void gt_gl_frames_kernel(GriffinLimArgs P) {
    extern __shared__ __attribute__((aligned(16))) float2 zsm[];      // [H] FFT buffer, then xs[H+1]
    const int N = P.n_fft, H = N >> 1, bits = P.log2_h, nb = H + 1;
    float2* xs = zsm + H;
    const int b = blockIdx.y, t = blockIdx.x, tid = threadIdx.x;
    const int Tb = gt_gl_frames_of(P, b);
    float* dst = P.frm_new + ((size_t)b * P.T + t) * N;
    if (t >= Tb) {
        for (int i = tid; i < N; i += blockDim.x) dst[i] = 0.f;
        return;
    }
    const float* mag = P.mag + ((size_t)b * P.T + t) * nb;
    if (INIT) {
        for (int k = tid; k <= H; k += blockDim.x) {
            float u;
            if (P.init_phase) u = P.init_phase[((size_t)b * P.T + t) * nb + k];
            else {
                const Philox4 r = gt_philox(P.seed, (uint32_t)(((size_t)b * P.T + t) * nb + k), 0u, 0u, 0x4000u);
                u = (float)(r.x >> 8) * (1.0f / 16777216.0f);          // [0, 1)
            }
            float sn, cs;
            sincosf(6.283185307179586f * u, &sn, &cs);
            const float m = mag[k];
            xs[k] = make_float2(m * cs, (k == 0 || k == H) ? 0.f : m * sn);   // irfft ignores Im X[0], Im X[N/2]
        }
    } else {
        const int L = P.hop * (Tb - 1);
        const float* frm = P.frm_old + (size_t)b * P.T * N;
        for (int n = tid; n < H; n += blockDim.x) {
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int i = 2 * n + e;
                const int u = gt_reflect(t * P.hop + i - H, L);
                v[e] = gt_gl_ola(P, frm, u + H, Tb) * P.window[i];
            }
            zsm[gt_bitrev(n, bits)] = make_float2(v[0], v[1]);
        }
        __syncthreads();
        gt_fft_lds(zsm, H, P.twiddle, -1.f);
        for (int k = tid; k <= H; k += blockDim.x) {
            const float2 a = zsm[k & (H - 1)], c = zsm[(H - k) & (H - 1)];
            const float er = 0.5f * (a.x + c.x), ei = 0.5f * (a.y - c.y);
            const float orr = 0.5f * (a.y + c.y), oi = -0.5f * (a.x - c.x);
            const float2 w = k < H ? P.twiddle[k] : make_float2(-1.f, 0.f);
            const float xr = er + (orr * w.x - oi * w.y), xi = ei + (orr * w.y + oi * w.x);
            const float r = sqrtf(xr * xr + xi * xi);
            const float m = mag[k];
            // exp(1j * angle(X)) = X / |X|; angle(0) = 0
            float2 unit = r > 0.f ? make_float2(xr / r, xi / r) : make_float2(1.f, 0.f);
            xs[k] = make_float2(m * unit.x, (k == 0 || k == H) ? 0.f : m * unit.y);
        }
    }
    __syncthreads();
    // inverse real FFT through the half-size complex one: Z[k] = E[k] + i O[k],
    // E = (X[k] + conj X[H-k]) / 2, O = (X[k] - conj X[H-k]) / 2 * conj(W_N^k)
    for (int k = tid; k < H; k += blockDim.x) {
        const float2 a = xs[k], c = xs[H - k];
        const float er = 0.5f * (a.x + c.x), ei = 0.5f * (a.y - c.y);
        const float dr = 0.5f * (a.x - c.x), di = 0.5f * (a.y + c.y);
        const float2 w = P.twiddle[k];                                  // conj(w) = (w.x, -w.y)
        const float orr = dr * w.x + di * w.y, oi = di * w.x - dr * w.y;
        zsm[gt_bitrev(k, bits)] = make_float2(er - oi, ei + orr);
    }
    __syncthreads();
    gt_fft_lds(zsm, H, P.twiddle, +1.f);
    const float sc = 1.f / (float)H;
    for (int n = tid; n < H; n += blockDim.x) {
        const float2 z = zsm[n];
        dst[2 * n] = z.x * sc * P.window[2 * n];
        dst[2 * n + 1] = z.y * sc * P.window[2 * n + 1];
    }
}

__global__ __launch_bounds__(256) void gt_gl_ola_kernel(GriffinLimArgs P) {
    const int b = blockIdx.y;
    const int Tb = gt_gl_frames_of(P, b);
    const int L = Tb > 0 ? P.hop * (Tb - 1) : 0;
    const float* frm = P.frm_old + (size_t)b * P.T * P.n_fft;
    float* y = P.ybuf + (size_t)b * P.ld_y;
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < L; u += gridDim.x * blockDim.x)
        y[u] = gt_gl_ola(P, frm, u + P.n_fft / 2, Tb);
}

#define GL_CHUNK 128
#define GL_WARM 2048
__global__ __launch_bounds__(256) void gt_gl_deemph_kernel(GriffinLimArgs P) {
    const int b = blockIdx.y;
    const int Tb = gt_gl_frames_of(P, b);
    const int L = Tb > 0 ? P.hop * (Tb - 1) : 0;
    const float* y = P.ybuf + (size_t)b * P.ld_y;
    float* out = P.wav + (size_t)b * P.ld_wav;
    if (blockIdx.x == 0 && threadIdx.x == 0 && P.wav_len) P.wav_len[b] = L;
    const int nchunks = (P.ld_wav + GL_CHUNK - 1) / GL_CHUNK;
    for (int ch = blockIdx.x * blockDim.x + threadIdx.x; ch < nchunks; ch += gridDim.x * blockDim.x) {
        const int n0 = ch * GL_CHUNK;
        double s = 0.0;
        for (int n = max(0, n0 - GL_WARM); n < n0 && n < L; ++n) s = (double)y[n] + (double)P.preemph * s;
        for (int n = n0; n < min(n0 + GL_CHUNK, P.ld_wav); ++n) {
            if (n < L) { s = (double)y[n] + (double)P.preemph * s; out[n] = (float)s; }
            else out[n] = 0.f;
        }
    }
}

hipError_t gt_gl_init() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gt_gl_frames_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gt_gl_frames_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
}

hipError_t gt_launch_griffin_lim(GriffinLimArgs a, float* frm_a, float* frm_b, hipStream_t s) {
    const int H = a.n_fft / 2;
    const size_t lds = (size_t)(2 * H + 1) * sizeof(float2);
    const int64_t total = (int64_t)a.B * a.T * (H + 1);
    hipLaunchKernelGGL(gt_gl_prepare_kernel, dim3((unsigned)std::min<int64_t>((total + 255) / 256, 65535)), dim3(256), 0, s, a);
    float* cur = frm_a;
    float* nxt = frm_b;
    a.frm_old = nullptr; a.frm_new = cur;
    hipLaunchKernelGGL(gt_gl_frames_kernel<true>, dim3(a.T, a.B), dim3(256), lds, s, a);
    for (int it = 0; it < a.iters; ++it) {
        a.frm_old = cur; a.frm_new = nxt;
        hipLaunchKernelGGL(gt_gl_frames_kernel<false>, dim3(a.T, a.B), dim3(256), lds, s, a);
        std::swap(cur, nxt);
    }
    a.frm_old = cur;
    a.ybuf = nxt;                                       // the other frame buffer is free: [B, T*n_fft] >= [B, hop*(T-1)]
    a.ld_y = (int64_t)a.T * a.n_fft;
    const int Lmax = a.hop * (a.T - 1);
    hipLaunchKernelGGL(gt_gl_ola_kernel, dim3(std::max(1, (Lmax + 255) / 256), a.B), dim3(256), 0, s, a);
    const int nchunks = (a.ld_wav + GL_CHUNK - 1) / GL_CHUNK;
    hipLaunchKernelGGL(gt_gl_deemph_kernel, dim3((nchunks + 255) / 256, a.B), dim3(256), 0, s, a);
    return hipGetLastError();
}
