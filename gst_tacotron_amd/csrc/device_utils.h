// Device-side helpers shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define GT_WAVE 64

// Cross-lane moves inside a 16-lane row on the VALU (DPP) -- no LDS round trip.  hipcc compiles every __shfl_* to ds_bpermute_b32
// (~100 cycles through the LDS crossbar, waited for with lgkmcnt): six dependent ones summed a hand-off's counter shards inside every
// poll, two rounds of three sat in every LSTM epilogue and three in the attention score (EXPERIMENTS round 5, item 8).
template <int CTRL>
__device__ __forceinline__ float gt_dpp(float x) {      // lanes without a source keep their own value
    const int v = __builtin_bit_cast(int, x);
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ uint32_t gt_dpp_u32(uint32_t x) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)x, (int)x, CTRL, 0xF, 0xF, false);
}
// (integer form of gt_row_sum below: arrival-counter shards)
template <int L>
__device__ __forceinline__ uint32_t gt_row_sum_u32(uint32_t s) {
    static_assert(L == 4 || L == 8 || L == 16, "a power of two inside a 16-lane row");
    s += gt_dpp_u32<0xB1>(s);
    s += gt_dpp_u32<0x4E>(s);
    if (L >= 8) s += gt_dpp_u32<0x141>(s);
    if (L >= 16) s += gt_dpp_u32<0x140>(s);
    return s;
}
// over all 64 lanes, every lane holding the result: the four in-row steps as DPP moves, the two cross-row steps through the LDS crossbar
// (two ds_bpermute round trips instead of six)
__device__ __forceinline__ float gt_wave_sum(float s);
__device__ __forceinline__ float gt_wave_max(float m);
// inclusive prefix sum over the 64 lanes of a FULLY ACTIVE wave: Hillis-Steele inside each 16-lane row on DPP row_shr moves (a lane
// without a source adds 0), then the totals of the rows in front, read as scalars (v_readlane) -- no LDS-crossbar round trip (as six
// __shfl_up steps it was six, twice per BMA alignment).  Row r adds ((T0 + T1) + T2) up to its predecessor, in that order.
__device__ __forceinline__ float gt_wave_incl_scan(float x, int lane) {
    auto shr = [](float v, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    x += shr(x, std::integral_constant<int, 0x111>{});
    x += shr(x, std::integral_constant<int, 0x112>{});
    x += shr(x, std::integral_constant<int, 0x114>{});
    x += shr(x, std::integral_constant<int, 0x118>{});
    const float t0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 15));
    const float t1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 31));
    const float t2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), 47));
    const int r = lane >> 4;
    const float before = r == 0 ? 0.f : (r == 1 ? t0 : (r == 2 ? t0 + t1 : (t0 + t1) + t2));
    return r == 0 ? x : x + before;
}
// the value of lane i + N of the same 16-lane row (row_shl:N) = __shfl_down(x, N, 16)
template <int N>
__device__ __forceinline__ float gt_row_down(float x) {
    static_assert(N >= 1 && N <= 15, "row_shl:1..15");
    return gt_dpp<0x100 + N>(x);
}
// sum over aligned groups of L = 4 / 8 / 16 lanes, every lane of the group holding the total: the same additions as the xor butterfly
// `for (d = 1; d < L; d <<= 1) s += __shfl_xor(s, d)` -- after the two quad steps the four lanes of a quad hold identical bits, so
// adding the mirror lane's value (row_half_mirror, row_mirror) is adding the xor partner's
template <int L>
__device__ __forceinline__ float gt_row_sum(float s) {
    static_assert(L == 4 || L == 8 || L == 16, "a power of two inside a 16-lane row");
    s += gt_dpp<0xB1>(s);                   // quad_perm [1,0,3,2]
    s += gt_dpp<0x4E>(s);                   // quad_perm [2,3,0,1]
    if (L >= 8) s += gt_dpp<0x141>(s);      // row_half_mirror
    if (L >= 16) s += gt_dpp<0x140>(s);     // row_mirror
    return s;
}
__device__ __forceinline__ float gt_wave_sum(float s) {
    s = gt_row_sum<16>(s);
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    return s;
}
__device__ __forceinline__ float gt_wave_max(float m) {
    m = fmaxf(m, gt_dpp<0xB1>(m));
    m = fmaxf(m, gt_dpp<0x4E>(m));
    m = fmaxf(m, gt_dpp<0x141>(m));
    m = fmaxf(m, gt_dpp<0x140>(m));
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    return m;
}

// tanh(x) = 1 - 2/(exp(2x)+1) on one v_exp_f32 + one v_rcp_f32 (5 VALU ops).  No clamp is needed: exp(2x) -> inf
// gives rcp -> 0 -> +1, exp(2x) -> 0 gives -1, never NaN.  abs error <= ~2e-7 (covered by the parity tests).
__device__ __forceinline__ float gt_tanh(float x) {
    const float e = __builtin_amdgcn_exp2f(x * 2.885390081777927f);      // exp(2x) = 2^(2x*log2(e))
    // (an explicit fma: whether `1 - 2 r` contracts is otherwise the optimiser's choice per call site, and kernels whose results
    // are compared bitwise -- the decode launch forms -- must round alike)
    return __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// two at a time on the packed fp32 VALU ops (v_pk_mul / v_pk_add / v_pk_fma): the attention score pass is 16 K tanh per
// utterance and issue-bound, and only the two transcendentals per element cannot be paired
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gt_tanh2(f32x2 x) {
    const f32x2 y = x * 2.885390081777927f;
    f32x2 e;
    e.x = __builtin_amdgcn_exp2f(y.x);
    e.y = __builtin_amdgcn_exp2f(y.y);
    e = e + 1.0f;
    f32x2 r;
    r.x = __builtin_amdgcn_rcpf(e.x);
    r.y = __builtin_amdgcn_rcpf(e.y);
    return __builtin_elementwise_fma(f32x2{-2.0f, -2.0f}, r, f32x2{1.0f, 1.0f});
}

__device__ __forceinline__ float gt_sigmoid(float x) {
    const float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);    // exp(-x)
    return __builtin_amdgcn_rcpf(1.0f + e);
}

// Philox4x32-10 counter RNG: throughput-mode randomness (prenet dropout, SMA noise) is generated
// on the device; parity runs inject the tensors instead.
struct Philox4 { uint32_t x, y, z, w; };

__device__ __forceinline__ Philox4 gt_philox(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}

__device__ __forceinline__ float gt_u01(uint32_t r) {        // (0,1]
    return ((float)(r >> 8) + 1.0f) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float gt_normal(uint32_t r0, uint32_t r1) {   // Box-Muller
    float u1 = gt_u01(r0), u2 = gt_u01(r1);
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.28318530718f * u2);
}

// Prenet dropout keep decisions in throughput mode (the reference draws them unseeded and always on, Taco2.py:283).
// At the reference's rate 0.5 a decision is ONE bit, and 32 of them -- columns [32 w, 32 w + 32) of one row of one layer at
// one step -- are one word of a counter hash of the seed: cheap enough for the front kernel to know, a few scalar
// instructions after it starts, which rows of the next Dense's weights will multiply an exact zero, BEFORE it requests
// them (dec_front.hip: the dropped half of its 384 KB weight pull is never loaded).  Other rates: Philox, one call per
// decision, no such shortcut.  Every consumer (front kernel, general prenet kernels, gt_rng_fill_kernel) goes through
// gt_drop_keep so they all see the same masks.
__device__ __forceinline__ uint32_t gt_mix32(uint32_t h) {      // murmur3 finaliser
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ uint32_t gt_keep_word(uint64_t seed, uint32_t step, uint32_t layer, uint32_t row, uint32_t word) {
    uint32_t h = gt_mix32((uint32_t)seed ^ (step * 0x9E3779B1u + layer));
    h = gt_mix32(h ^ (uint32_t)(seed >> 32) ^ (row * 0x85EBCA77u));
    return gt_mix32(h ^ (word * 0xC2B2AE3Du + 0x27D4EB2Fu));
}
__device__ __forceinline__ bool gt_keep_is_hashed(float rate) { return rate == 0.5f; }
// keep (1) / drop (0) of column `col` of row `row` of prenet layer `layer` (0 / 1) at decode step `step`
__device__ __forceinline__ float gt_drop_keep(uint64_t seed, uint32_t step, uint32_t layer, uint32_t row, uint32_t col, uint32_t ncols,
                                             float rate);

// stream ids for the Philox counter's 4th word
#define GT_RNG_PRENET0 0x1000u
#define GT_RNG_NOISE   0x2000u

__device__ __forceinline__ float gt_drop_keep(uint64_t seed, uint32_t step, uint32_t layer, uint32_t row, uint32_t col, uint32_t ncols,
                                             float rate) {
    if (gt_keep_is_hashed(rate)) return (float)((gt_keep_word(seed, step, layer, row, col >> 5) >> (col & 31)) & 1u);
    return (gt_u01(gt_philox(seed, row * ncols + col, step, 0u, GT_RNG_PRENET0 + layer).x) > rate) ? 1.f : 0.f;
}

// ---- raw buffer loads (unconditional, bounds-checked by the descriptor; see front_lean.h for why the decode loop uses them)
#define GT_OOB 0x80000000u      // voffset beyond any descriptor's num_records: the load returns 0 and accesses nothing

__device__ __forceinline__ __amdgpu_buffer_rsrc_t gt_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float gt_bload1(__amdgpu_buffer_rsrc_t rs, uint32_t voff) {
    const unsigned int t = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, 0, 0);
    return __builtin_bit_cast(float, t);
}
__device__ __forceinline__ float4 gt_bload4(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
    // (NOT __builtin_bit_cast(float, t[i]): on a vector ELEMENT clang 20 reads element 0 for every i)
    const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 0);
    float4 r;
    __builtin_memcpy(&r, &t, 16);
    return r;
}

// 16-byte load with the sc1 (agent-scope, L2-bypassing) cache policy, tracked by the compiler's wait counting (cpol bit 4 = sc1
// on gfx940+).  For activations another XCD has just written: measured with tools/persist_phase.hip, a 256-workgroup launch in
// which every workgroup reads the same 128 KB the previous launch wrote is 0.6 us shorter with sc1 loads than with plain ones
// (6.29 vs 6.88 us) -- 32 CUs of an XCD missing the same fresh lines serialise in that XCD's L2; bypassing it, each request goes
// to the Infinity Cache on its own.
__device__ __forceinline__ float4 gt_bload4_sc1(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
    const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, (int)soff, 16);
    float4 r;
    __builtin_memcpy(&r, &t, 16);
    return r;
}

// fp32 -> bf16 bits, round to nearest even: what the bf16 GEMM bodies do to an activation on its way into the MFMA
__device__ __forceinline__ uint16_t gt_bf16_bits(const float v) {
    const __bf16 h = (__bf16)v;
    uint16_t u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}

// diagnostic: phase stamp (constant 100 MHz counter) written by thread 0 of block 0 when dbg != NULL
#define GT_STAMP(dbg, slot)                                                                    \
    do {                                                                                       \
        if ((dbg) && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) \
            (dbg)[slot] = __builtin_amdgcn_s_memrealtime();                                    \
    } while (0)
