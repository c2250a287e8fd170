// Split-bf16 ("bf16 x 3") prototype asked for by the round-5 review: can the fp32 GEMMs of this path move onto the bf16 matrix pipe
// (v_mfma_f32_16x16x32_bf16: 16 cycles per 16 384 FLOP and SIMD) at fp32 accuracy?  On gfx950 v_mfma_f32_16x16x4_f32 takes 32 cycles for
// 2 048 FLOP: per 32-deep k-block of a 16 x 16 tile fp32 costs 8 x 32 = 256 cycles, six bf16 products 6 x 16 = 96 (2.67 x), three 48 (5.3 x).
//   value = hi + mid + lo, each plane a bf16 (hi = rne(v), mid = rne(v - hi), lo = rne(v - hi - mid): 8 + 8 + 8 significand bits)
//   x6: hh, hm, mh, mm, hl, lh   (everything down to 2^-24 relative)        x3: hh, hm, mh (2^-16)
//   x5: two-plane weights (the bytes of an fp32 weight: what a register-resident tile could hold) x three-plane activations
// Two questions, two parts:
//   RATE      per-CU loops with every operand plane re-read from LDS (ds_read_b128), 8 waves, MI x NJ tiles of 16 x 16 per wave, random
//             bf16 bit patterns; fp32-EQUIVALENT TFLOP/s (2 M N K / time) against the 16x16x4 fp32 loop of tools/mfma_rate.hip's kind, and
//             the cost of splitting an fp32 A fragment in the consumer (VALU) instead of reading planes its producer wrote.
//   ACCURACY  K = 2 048 dot products (4 096 of them per case), operands N(0,1) and wide-exponent: error against float64 in units of
//             2^-24 sum|a b|, for the fp32 MFMA chain, x6 (one accumulator, and low-order products in an accumulator of their own), x5, x3.
// Gate: rate >= 2.2 x the 16x16x4 rate AND x6 as accurate as the fp32 MFMA chain it replaces (within 2 units of it).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/split_bf16 tools/split_bf16.hip && tools/split_bf16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ------------------------------------------------------------------------------------------------------------------ RATE
// LDS image: A planes [3][MI*16 rows * 8 waves/..] -- every wave reads its own MI row tiles and NJ column tiles of a shared image;
// rows are 64 bf16 (one 2 x 32-k step) + 8 pad = 72 halves = 144 B apart (ds_read_b128 conflict-free across the 16 rows x 4 k-slices).
template <int NP, int MI, int NJ, int NT>      // NP products per k-block (3 / 5 / 6)
__global__ __launch_bounds__(NT) void ksplit(float* out, int iters, unsigned long long* cyc) {
    constexpr int ROWS = 64, COLS = 64, LD = 72;
    __shared__ __attribute__((aligned(16))) unsigned short lds[3 * (ROWS + COLS) * LD];      // 55 KB
    for (int i = threadIdx.x; i < 3 * (ROWS + COLS) * LD; i += NT) {
        unsigned int h = (unsigned int)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        lds[i] = (unsigned short)(0x3C00u + (h & 0x83FFu));     // +-[0.5, 2)
    }
    __syncthreads();
    f32x4 acc[MI][NJ];
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned short* Ab = lds + (((wave & 1) * 16 + (lane & 15)) * LD) + (lane >> 4) * 8;
    const unsigned short* Bb = lds + 3 * ROWS * LD + ((((wave >> 1) & 1) * 16 + (lane & 15)) * LD) + (lane >> 4) * 8;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t a[3][MI], b[3][NJ];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < MI; ++i) a[p][i] = *reinterpret_cast<const bf16x8_t*>(Ab + (p * ROWS + (i * 16 + (it & 1) * 8) % 32) * LD + ks * 32);
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[p][j] = *reinterpret_cast<const bf16x8_t*>(Bb + (p * COLS + (j * 16) % 32) * LD + ks * 32);
            }
            // product list in decreasing magnitude: (a plane, b plane)
            constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
            constexpr int PA5[5] = {0, 1, 0, 2, 1}, PB5[5] = {0, 0, 1, 0, 1};       // b = two-plane weights
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int pa = NP == 5 ? PA5[q] : PA[q], pb = NP == 5 ? PB5[q] : PB[q];
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[pa][i], b[pb][j], acc[i][j], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// the three planes of eight fp32 values (one lane's A fragment of a 32-k block), as the consumer would have to make them
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8_t& hi, bf16x8_t& mid, bf16x8_t& lo) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)v[e];
        const float r1 = v[e] - (float)h;
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - (float)m;
        hi[e] = h; mid[e] = m; lo[e] = (__bf16)r2;
    }
}
// the same loop with the A planes made IN the consumer from fp32 fragments read from LDS (B planes pre-split, as offline weights are)
template <int NP, int MI, int NJ, int NT>
__global__ __launch_bounds__(NT) void ksplit_cons(float* out, int iters, unsigned long long* cyc) {
    constexpr int ROWS = 64, COLS = 64, LD = 72, LDF = 68;
    __shared__ __attribute__((aligned(16))) unsigned short lds[3 * COLS * LD];
    __shared__ __attribute__((aligned(16))) float ldf[ROWS * LDF];
    for (int i = threadIdx.x; i < 3 * COLS * LD; i += NT) {
        unsigned int h = (unsigned int)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        lds[i] = (unsigned short)(0x3C00u + (h & 0x83FFu));
    }
    for (int i = threadIdx.x; i < ROWS * LDF; i += NT) {
        unsigned int h = (unsigned int)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        ldf[i] = __builtin_bit_cast(float, 0x3F000000u | (h & 0x807FFFFFu));
    }
    __syncthreads();
    f32x4 acc[MI][NJ];
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* Af = ldf + (((wave & 1) * 16 + (lane & 15)) * LDF) + (lane >> 4) * 8;
    const unsigned short* Bb = lds + ((((wave >> 1) & 1) * 16 + (lane & 15)) * LD) + (lane >> 4) * 8;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t a[3][MI], b[3][NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                float v[8];
                const float4 v0 = *reinterpret_cast<const float4*>(Af + ((i * 16 + (it & 1) * 8) % 32) * LDF + ks * 32);
                const float4 v1 = *reinterpret_cast<const float4*>(Af + ((i * 16 + (it & 1) * 8) % 32) * LDF + ks * 32 + 4);
                v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
                split8(v, a[0][i], a[1][i], a[2][i]);
            }
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[p][j] = *reinterpret_cast<const bf16x8_t*>(Bb + (p * COLS + (j * 16) % 32) * LD + ks * 32);
            constexpr int PA[6] = {0, 0, 1, 1, 0, 2}, PB[6] = {0, 1, 0, 1, 2, 0};
#pragma unroll
            for (int q = 0; q < NP; ++q)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PA[q]][i], b[PB[q]][j], acc[i][j], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

// the fp32 pipe on the same tile shape: 16x16x4, A and B fragments of 4 k each per ds_read_b128... one float per lane and MFMA: a lane
// reads 16 bytes = its element of four consecutive k-steps (the blocked layout the product kernels use)
template <int MI, int NJ, int NT>
__global__ __launch_bounds__(NT) void kfp32(float* out, int iters, unsigned long long* cyc) {
    __shared__ __attribute__((aligned(16))) float lds[8192];            // 32 KB
    for (int i = threadIdx.x; i < 8192; i += NT) {
        unsigned int h = (unsigned int)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        lds[i] = __builtin_bit_cast(float, 0x3F000000u | (h & 0x807FFFFFu));
    }
    __syncthreads();
    f32x4 acc[MI][NJ];
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0, 0, 0, 0};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float4* ap = reinterpret_cast<const float4*>(lds) + lane + (wave & 1) * 64;
    const float4* bp = reinterpret_cast<const float4*>(lds) + 1024 + lane + ((wave >> 1) & 1) * 64;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {           // 4 x 16 k = the 64 k of one ksplit iteration
            float4 a[MI], b[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = ap[((i * 2 + ks + (it & 1)) * 128) % 896];
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = bp[((j * 2 + ks) * 128) % 896];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
                }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < MI; ++i) for (int j = 0; j < NJ; ++j) for (int e = 0; e < 4; ++e) s += acc[i][j][e];
    out[blockIdx.x * NT + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <typename K>
static double run_rate(const char* name, K kern, int nt, int mi, int nj, int iters, float* out, unsigned long long* cyc, double base) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(256), dim3(nt), 0, 0, out, iters / 20, cyc);
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(256), dim3(nt), 0, 0, out, iters, cyc);
    CK(hipEventRecord(e1, 0)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
    // fp32-equivalent work: every iteration multiplies MI x NJ tiles of 16 x 16 over 64 k, per wave
    const double tf = 256.0 * (nt / 64) * (double)iters * mi * nj * 16 * 16 * 64 * 2.0 / (ms * 1e-3) * 1e-12;
    printf("%-64s %8.1f us  %7.1f fp32-equivalent TFLOP/s  %6.0f clk/us", name, ms * 1e3, tf, (double)c / (ms * 1e3));
    if (base > 0) printf("  = %.2f x fp32 MFMA", tf / base);
    printf("\n");
    return tf;
}

// ------------------------------------------------------------------------------------------------------------------ ACCURACY
// one wave = one 16 x 16 output tile over K; a: [16][K], b: [K][16] fp32 in memory; mode: 0 fp32 MFMA, 1 x6 one accumulator,
// 2 x6 with the four low-order products in a second accumulator (added at the end), 3 x3, 4 x5 (b as two planes)
__global__ __launch_bounds__(64) void kacc(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, int K, int mode) {
    const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
    const float* A = a + (size_t)blockIdx.x * 16 * K;
    const float* B = b + (size_t)blockIdx.x * 16 * K;
    f32x4 acc = {0, 0, 0, 0}, low = {0, 0, 0, 0};
    if (mode == 0) {
        for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(size_t)r * K + k + q], B[(size_t)(k + q) * 16 + r], acc, 0, 0, 0);
    } else {
        for (int k = 0; k < K; k += 32) {
            float av[8], bv[8];
            for (int e = 0; e < 8; ++e) { av[e] = A[(size_t)r * K + k + q * 8 + e]; bv[e] = B[(size_t)(k + q * 8 + e) * 16 + r]; }
            bf16x8_t ah, am, al, bh, bm, bl;
            split8(av, ah, am, al);
            split8(bv, bh, bm, bl);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
            if (mode == 1) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
            } else if (mode == 2) {
                low = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, low, 0, 0, 0);
                low = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, low, 0, 0, 0);
                low = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, low, 0, 0, 0);
                low = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, low, 0, 0, 0);
                low = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, low, 0, 0, 0);
            } else if (mode == 3) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
            } else {            // x5: b = hi + mid only (two planes = the bytes of an fp32 weight)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, acc, 0, 0, 0);
            }
        }
        for (int e = 0; e < 4; ++e) acc[e] += low[e];
    }
    for (int e = 0; e < 4; ++e) out[(size_t)blockIdx.x * 256 + (q * 4 + e) * 16 + r] = acc[e];
}

static void accuracy_case(const char* name, int K, int wide, unsigned seed) {
    const int NB = 16;      // 16 tiles x 256 dot products
    std::mt19937 rng(seed);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::uniform_int_distribution<int> ex(-8, 0);
    std::vector<float> a((size_t)NB * 16 * K), b((size_t)NB * 16 * K);
    for (auto& v : a) v = nd(rng) * (wide ? std::ldexp(1.f, ex(rng)) : 1.f);
    for (auto& v : b) v = nd(rng) * (wide ? std::ldexp(1.f, ex(rng)) : 1.f) * 0.05f;
    float *da, *db, *dout;
    CK(hipMalloc(&da, a.size() * 4)); CK(hipMalloc(&db, b.size() * 4)); CK(hipMalloc(&dout, (size_t)NB * 256 * 4));
    CK(hipMemcpy(da, a.data(), a.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    // float64 reference, the fp32 sequential fma chain, and the scale an fp32 sum's error is measured in
    std::vector<double> ref((size_t)NB * 256), mag((size_t)NB * 256);
    std::vector<float> seq((size_t)NB * 256);
    for (int t = 0; t < NB; ++t)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double s = 0, m = 0; float f = 0.f;
                for (int k = 0; k < K; ++k) {
                    const float x = a[((size_t)t * 16 + i) * K + k], y = b[((size_t)t * K + k) * 16 + j];
                    s += (double)x * (double)y; m += std::fabs((double)x * (double)y); f = std::fmaf(x, y, f);
                }
                ref[(size_t)t * 256 + i * 16 + j] = s; mag[(size_t)t * 256 + i * 16 + j] = m; seq[(size_t)t * 256 + i * 16 + j] = f;
            }
    // error in units of 2^-24 sum|a b| -- the scale an fp32 summation's rounding errors live on.  (Units of the RESULT's ulp say nothing
    // here: a random-sign sum is ~sqrt(K) smaller than the sum of its terms' magnitudes, and arbitrarily smaller where it cancels.)
    auto report = [&](const char* what, const std::vector<float>& got) {
        double max_u = 0, rms_u = 0;
        for (size_t i = 0; i < got.size(); ++i) {
            const double u = std::fabs((double)got[i] - ref[i]) / (mag[i] * std::ldexp(1.0, -24));
            max_u = std::max(max_u, u); rms_u += u * u;
        }
        printf("  %-58s max %6.2f  rms %5.2f   (x 2^-24 sum|a b|)\n", what, max_u, std::sqrt(rms_u / got.size()));
    };
    printf("%s (K = %d, %d dot products; error against the float64 sum)\n", name, K, NB * 256);
    report("fp32 fma chain on the host (k ascending)", seq);
    const char* names[5] = {"v_mfma_f32_16x16x4_f32 chain", "split-bf16 x6, one accumulator", "split-bf16 x6, low-order products in their own accumulator",
                            "split-bf16 x3 (hh, hm, mh)", "split-bf16 x5 (two-plane b, three-plane a)"};
    std::vector<float> got((size_t)NB * 256);
    for (int mode = 0; mode < 5; ++mode) {
        hipLaunchKernelGGL(kacc, dim3(NB), dim3(64), 0, 0, da, db, dout, K, mode);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost));
        report(names[mode], got);
    }
    CK(hipFree(da)); CK(hipFree(db)); CK(hipFree(dout));
}

int main() {
    float* out; unsigned long long* cyc; CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 8));
    printf("== RATE: 256 workgroups x 8 waves, operands re-read from LDS every k-block, random bit patterns\n");
    const double b22 = run_rate("fp32 16x16x4, 2 x 2 tiles per wave", kfp32<2, 2, 512>, 512, 2, 2, 20000, out, cyc, 0);
    const double b24 = run_rate("fp32 16x16x4, 2 x 4 tiles per wave", kfp32<2, 4, 512>, 512, 2, 4, 10000, out, cyc, 0);
    const double b44 = run_rate("fp32 16x16x4, 4 x 4 tiles per wave", kfp32<4, 4, 512>, 512, 4, 4, 5000, out, cyc, 0);
    const double base = std::max(b22, std::max(b24, b44));
    run_rate("split-bf16 x6, 2 x 2 tiles per wave (planes from LDS)", ksplit<6, 2, 2, 512>, 512, 2, 2, 40000, out, cyc, base);
    run_rate("split-bf16 x6, 2 x 4 tiles per wave", ksplit<6, 2, 4, 512>, 512, 2, 4, 20000, out, cyc, base);
    run_rate("split-bf16 x6, 4 x 4 tiles per wave", ksplit<6, 4, 4, 512>, 512, 4, 4, 10000, out, cyc, base);
    run_rate("split-bf16 x6, 4 x 4 tiles per wave, 4 waves", ksplit<6, 4, 4, 256>, 256, 4, 4, 10000, out, cyc, base);
    run_rate("split-bf16 x5 (two-plane weights), 4 x 4 tiles per wave", ksplit<5, 4, 4, 512>, 512, 4, 4, 10000, out, cyc, base);
    run_rate("split-bf16 x3, 2 x 4 tiles per wave", ksplit<3, 2, 4, 512>, 512, 2, 4, 20000, out, cyc, base);
    run_rate("split-bf16 x3, 4 x 4 tiles per wave", ksplit<3, 4, 4, 512>, 512, 4, 4, 10000, out, cyc, base);
    run_rate("split-bf16 x6, 2 x 4, A split in the CONSUMER from fp32", ksplit_cons<6, 2, 4, 512>, 512, 2, 4, 10000, out, cyc, base);
    run_rate("split-bf16 x6, 4 x 4, A split in the CONSUMER from fp32", ksplit_cons<6, 4, 4, 512>, 512, 4, 4, 5000, out, cyc, base);
    run_rate("split-bf16 x6, 1 x 4 (a skinny GEMM's wave), consumer split", ksplit_cons<6, 1, 4, 512>, 512, 1, 4, 20000, out, cyc, base);
    printf("== ACCURACY\n");
    accuracy_case("N(0,1) x N(0,0.05)", 2048, 0, 1);
    accuracy_case("wide exponents (2^-8 .. 2^0 on both operands)", 2048, 1, 2);
    accuracy_case("N(0,1) x N(0,0.05), K = 512", 512, 0, 3);
    return 0;
}
