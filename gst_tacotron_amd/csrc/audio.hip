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
