// Ceiling of the fp32 matrix pipe as the Winograd kernels use it: W waves per SIMD, C independent accumulator chains per wave of
// v_mfma_f32_32x32x2_f32 (or 16x16x4), nothing else in the loop.  Prints TFLOP/s over the whole chip and cycles per MFMA per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/mfma_rate tools/mfma_rate.hip && tools/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
template <int C, int NT>
__global__ __launch_bounds__(NT) void k32(float* out, int iters, unsigned long long* cyc) {
    f32x16 acc[C];
    for (int c = 0; c < C; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < C; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int C, int NT>
__global__ __launch_bounds__(NT) void k16(float* out, int iters, unsigned long long* cyc) {
    f32x4 acc[C];
    for (int c = 0; c < C; ++c) for (int e = 0; e < 4; ++e) acc[c][e] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < C; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < C; ++c) for (int e = 0; e < 4; ++e) s += acc[c][e];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// ... and with V independent VALU fmas (and optionally one LDS read) issued between consecutive MFMAs of ONE wave per SIMD: does the wave's
// other work run in the shadow of its own MFMAs?
template <int V, int LDSR, int NT>
__global__ __launch_bounds__(NT) void kmix(float* out, int iters, unsigned long long* cyc) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += NT) lds[i] = i * 1e-4f;
    __syncthreads();
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f;
    float w[16];
    for (int i = 0; i < 16; ++i) w[i] = a + i;
    int ix = threadIdx.x & 1023;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
#pragma unroll
            for (int v = 0; v < V; ++v) w[v & 15] = __builtin_fmaf(w[v & 15], 1.0001f, 0.5f);
            if (LDSR) { b += lds[ix]; ix = (ix + 64) & 1023; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    for (int i = 0; i < 16; ++i) s += w[i];
    out[blockIdx.x * NT + threadIdx.x] = s + b;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// ... and 16x16x4 MFMAs whose B operands come from LDS (two ds_read_b128 per 8 MFMAs, requested one step ahead) and, with XG, whose A
// operand comes from memory (one 16-byte load per lane per 8 MFMAs, 8 steps ahead): the M-split decode body's inner loop (msplit_body.h).
template <int XG, int NT>
__global__ __launch_bounds__(NT) void kfeed(float* out, const float* __restrict__ xg, int iters, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float lds[32768];            // 128 KB
    for (int i = threadIdx.x; i < 32768; i += NT) lds[i] = i * 1e-5f;
    __syncthreads();
    f32x4 acc[2][8];
    for (int c = 0; c < 16; ++c) for (int e = 0; e < 4; ++e) acc[c >> 3][c & 7][e] = 0.f;
    const int lane = threadIdx.x & 63;
    const float4* wl = reinterpret_cast<const float4*>(lds) + lane;
    const float4* xp = reinterpret_cast<const float4*>(xg) + (size_t)(blockIdx.x * (NT / 64) + (threadIdx.x >> 6)) * 64 * 64 + lane;
    float4 x[8];
    for (int i = 0; i < 8; ++i) x[i] = XG ? xp[i * 64] : make_float4(lane * 1e-3f, 0.5f, 0.25f, 0.125f);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        float4 b[2][2];
        b[0][0] = wl[0]; b[0][1] = wl[64];
#pragma unroll
        for (int kb = 0; kb < 64; ++kb) {
            if (kb + 1 < 64) { b[(kb + 1) & 1][0] = wl[((kb + 1) * 2) * 64]; b[(kb + 1) & 1][1] = wl[((kb + 1) * 2 + 1) * 64]; }
            const float4 xv = x[kb % 8];
            for (int t = 0; t < 2; ++t) acc[t][kb % 8] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.x, b[kb & 1][t].x, acc[t][kb % 8], 0, 0, 0);
            for (int t = 0; t < 2; ++t) acc[t][kb % 8] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.y, b[kb & 1][t].y, acc[t][kb % 8], 0, 0, 0);
            for (int t = 0; t < 2; ++t) acc[t][kb % 8] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.z, b[kb & 1][t].z, acc[t][kb % 8], 0, 0, 0);
            for (int t = 0; t < 2; ++t) acc[t][kb % 8] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv.w, b[kb & 1][t].w, acc[t][kb % 8], 0, 0, 0);
            if (XG && kb + 8 < 64) x[kb % 8] = xp[(kb + 8) * 64];
            __builtin_amdgcn_sched_barrier(0);
        }
        if (XG) for (int i = 0; i < 8; ++i) x[i] = xp[i * 64];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < 16; ++c) for (int e = 0; e < 4; ++e) s += acc[c >> 3][c & 7][e];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <typename K>
static void runfeed(const char* name, K kern, int nt, float* out, const float* xg, unsigned long long* cyc) {
    const int iters = 40;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(256), dim3(nt), 0, 0, out, xg, 2, cyc);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(256), dim3(nt), 0, 0, out, xg, iters, cyc);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double mfma_per_simd = (double)iters * 512 * (nt / 256.0);
    printf("%-44s %7.1f us  %6.1f TFLOP/s  %6.1f s_memtime ticks per MFMA and SIMD  (%.0f ticks/us)\n", name, ms * 1e3,
           256.0 * (nt / 64) * iters * 512 * 2048.0 / (ms * 1e-3) * 1e-12, (double)c / mfma_per_simd, (double)c / (ms * 1e3));
}

// ... and the bf16 pipe as gt_conv5_bf16_kernel uses it: 32x32x16 bf16 MFMAs, a wave's 2 x 4 accumulator tiles, operands read from LDS
// (6 ds_read_b128 per 8 MFMAs), data = pseudo-random bf16 (RND) or a constant: what the pipe sustains under its own power draw.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <int RND, int NT>
__global__ __launch_bounds__(NT) void kbf16(float* out, int iters, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[40960];            // 80 KB
    for (int i = threadIdx.x; i < 40960; i += NT) {
        unsigned int h = (unsigned int)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        lds[i] = RND ? (unsigned short)(0x3C00u + (h & 0x83FFu)) : (unsigned short)0x3F80u;     // +-[0.5, 2) or 1.0
    }
    __syncthreads();
    f32x16 acc[2][4];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned short* Ab = lds + ((wave >> 1) * 64 + (lane & 31)) * 72 + (lane >> 5) * 8;
    const unsigned short* Bb = lds + 20480 + ((wave & 1) * 128 + (lane & 31)) * 72 + (lane >> 5) * 8;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            bf16x8_t av[2], bv[4];
            for (int i = 0; i < 2; ++i) av[i] = *reinterpret_cast<const bf16x8_t*>(Ab + ((i * 32 + (it & 3)) * 72 + ks * 16) % 18000);
            for (int j = 0; j < 4; ++j) bv[j] = *reinterpret_cast<const bf16x8_t*>(Bb + (j * 32 * 72 + ks * 16) % 18000);
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <typename K>
static void runbf(const char* name, K kern, int nt, float* out, unsigned long long* cyc) {
    const int iters = 4000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(256), dim3(nt), 0, 0, out, 10, cyc);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(256), dim3(nt), 0, 0, out, iters, cyc);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-44s %7.1f us  %6.1f TFLOP/s  %6.1f ticks per MFMA and wave  (%.0f ticks/us)\n", name, ms * 1e3,
           256.0 * (nt / 64) * iters * 32 * 32768.0 / (ms * 1e-3) * 1e-12, (double)c / (iters * 32.0), (double)c / (ms * 1e3));
}

// ... the fp32 pipe on RANDOM operands from LDS, both shapes: 16x16x4 reads one A and one B register per 2048 FLOP, 32x32x2 per 4096 FLOP.
template <int BIG, int NT, int WIDE>
__global__ __launch_bounds__(NT) void kf32(float* out, int iters, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float lds[16384];            // 64 KB
    for (int i = threadIdx.x; i < 16384; i += NT) {
        unsigned int h = (unsigned int)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        // WIDE = 0: +-[0.5, 1) (random sign and mantissa, one exponent); 1: the exponent random over 2^-8 .. 2^-1 as well
        lds[i] = WIDE ? __builtin_bit_cast(float, (0x3B800000u + (((h >> 23) & 7u) << 23)) | (h & 0x807FFFFFu))
                      : __builtin_bit_cast(float, 0x3F000000u | (h & 0x807FFFFFu));
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4* ap = reinterpret_cast<const float4*>(lds) + lane + wave * 64;
    const float4* bp = reinterpret_cast<const float4*>(lds) + 2048 + lane;
    f32x16 accB[4];
    f32x4 accS[16];
    for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) accB[c][e] = 0.f;
    for (int c = 0; c < 16; ++c) for (int e = 0; e < 4; ++e) accS[c][e] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int kb = 0; kb < 16; ++kb) {
            const float4 a = ap[((kb + it) & 15) * 64 % 1024], b = bp[(kb * 64 + (it & 3) * 16) % 1024];
            if (BIG) {
                accB[kb & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, accB[kb & 3], 0, 0, 0);
                accB[kb & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, accB[kb & 3], 0, 0, 0);
            } else {
                accS[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, accS[kb], 0, 0, 0);
                accS[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, accS[kb], 0, 0, 0);
                accS[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, accS[kb], 0, 0, 0);
                accS[kb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, accS[kb], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int e = 0; e < 16; ++e) s += accB[c][e];
    for (int c = 0; c < 16; ++c) for (int e = 0; e < 4; ++e) s += accS[c][e];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <typename K>
static void runf32(const char* name, K kern, int nt, double flop_per_iter_wave, float* out, unsigned long long* cyc) {
    const int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(256), dim3(nt), 0, 0, out, 10, cyc);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(256), dim3(nt), 0, 0, out, iters, cyc);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-44s %7.1f us  %6.1f TFLOP/s  (%.0f shader clocks per us)\n", name, ms * 1e3, 256.0 * (nt / 64) * iters * flop_per_iter_wave / (ms * 1e-3) * 1e-12, (double)c / (ms * 1e3));
}
template <typename K>
static void run(const char* name, K kern, int nt, int chains, double flop_per_mfma, float* out, unsigned long long* cyc) {
    const int iters = 4000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(256), dim3(nt), 0, 0, out, 100, cyc);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(256), dim3(nt), 0, 0, out, iters, cyc);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    const double mfma_per_simd = (double)iters * 8 * chains * (nt / 256.0);
    printf("%-44s %7.1f us  %6.1f TFLOP/s  %6.1f s_memtime ticks per MFMA and SIMD  (%.0f ticks/us)\n", name, ms * 1e3,
           256.0 * (nt / 64) * iters * 8 * chains * flop_per_mfma / (ms * 1e-3) * 1e-12, (double)c / mfma_per_simd, (double)c / (ms * 1e3));
}
int main() {
    float* out; unsigned long long* cyc; CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 8));
    run("32x32x2  1 wave/SIMD, 1 chain", k32<1, 256>, 256, 1, 4096, out, cyc);
    run("32x32x2  1 wave/SIMD, 2 chains", k32<2, 256>, 256, 2, 4096, out, cyc);
    run("32x32x2  1 wave/SIMD, 4 chains", k32<4, 256>, 256, 4, 4096, out, cyc);
    run("32x32x2  1 wave/SIMD, 16 chains (256 acc)", k32<16, 256>, 256, 16, 4096, out, cyc);
    run("32x32x2  2 waves/SIMD, 1 chain", k32<1, 512>, 512, 1, 4096, out, cyc);
    run("32x32x2  2 waves/SIMD, 2 chains", k32<2, 512>, 512, 2, 4096, out, cyc);
    run("32x32x2  2 waves/SIMD, 8 chains", k32<8, 512>, 512, 8, 4096, out, cyc);
    run("16x16x4  1 wave/SIMD, 1 chain", k16<1, 256>, 256, 1, 2048, out, cyc);
    run("16x16x4  1 wave/SIMD, 4 chains", k16<4, 256>, 256, 4, 2048, out, cyc);
    run("16x16x4  2 waves/SIMD, 4 chains", k16<4, 512>, 512, 4, 2048, out, cyc);
    run("32x32x2 1 wave/SIMD + 4 fma per MFMA", kmix<4, 0, 256>, 256, 1, 4096, out, cyc);
    run("32x32x2 1 wave/SIMD + 8 fma per MFMA", kmix<8, 0, 256>, 256, 1, 4096, out, cyc);
    run("32x32x2 1 wave/SIMD + 12 fma per MFMA", kmix<12, 0, 256>, 256, 1, 4096, out, cyc);
    run("32x32x2 1 wave/SIMD + 16 fma per MFMA", kmix<16, 0, 256>, 256, 1, 4096, out, cyc);
    run("32x32x2 1 wave/SIMD + 24 fma per MFMA", kmix<24, 0, 256>, 256, 1, 4096, out, cyc);
    run("32x32x2 1 wave/SIMD + 4 fma + LDS read", kmix<4, 1, 256>, 256, 1, 4096, out, cyc);
    run("32x32x2 2 waves/SIMD + 8 fma per MFMA", kmix<8, 0, 512>, 512, 1, 4096, out, cyc);
    run("32x32x2 2 waves/SIMD + 16 fma per MFMA", kmix<16, 0, 512>, 512, 1, 4096, out, cyc);
    float* xg; CK(hipMalloc(&xg, (size_t)256 * 8 * 64 * 64 * 16));
    CK(hipMemset(xg, 0, (size_t)256 * 8 * 64 * 64 * 16));
    runfeed("16x16x4 B from LDS, 1 wave/SIMD", kfeed<0, 256>, 256, out, xg, cyc);
    runfeed("16x16x4 B from LDS, 2 waves/SIMD", kfeed<0, 512>, 512, out, xg, cyc);
    runfeed("16x16x4 B from LDS + A from memory, 2 waves/SIMD", kfeed<1, 512>, 512, out, xg, cyc);
    runbf("bf16 32x32x16 from LDS, constant data, 8 waves", kbf16<0, 512>, 512, out, cyc);
    runbf("bf16 32x32x16 from LDS, random data, 8 waves", kbf16<1, 512>, 512, out, cyc);
    runbf("bf16 32x32x16 from LDS, random data, 4 waves", kbf16<1, 256>, 256, out, cyc);
    runf32("fp32 16x16x4, random sign+mantissa, 8 waves", kf32<0, 512, 0>, 512, 64 * 2048.0, out, cyc);
    runf32("fp32 32x32x2, random sign+mantissa, 8 waves", kf32<1, 512, 0>, 512, 32 * 4096.0, out, cyc);
    runf32("fp32 16x16x4, + random exponent, 8 waves", kf32<0, 512, 1>, 512, 64 * 2048.0, out, cyc);
    runf32("fp32 32x32x2, + random exponent, 8 waves", kf32<1, 512, 1>, 512, 32 * 4096.0, out, cyc);
    return 0;
}
