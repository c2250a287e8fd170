// Shared by the Winograd Conv1D kernels (gemm_conv.hip: fp32 matrix pipe; conv_wino_split.hip: split-bf16 x6 on the bf16 pipe):
// the Cook-Toom transform rows at compile time, the raw tap gather and the input transform.
#pragma once
#include "device_utils.h"
#include "kernels.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MO>
struct Wino {
    static constexpr int ALPHA = MO + 4;
    static constexpr float bt(int xi, int i) {
        if (MO == 2) {
            constexpr float t[6][6] = {{0.25f, 0.f, -1.25f, 0.f, 1.f, 0.f},   {0.f, -0.25f, -0.25f, 1.f, 1.f, 0.f},
                                       {0.f, 0.25f, -0.25f, -1.f, 1.f, 0.f},  {0.f, -0.5f, -1.f, 0.5f, 1.f, 0.f},
                                       {0.f, 0.5f, -1.f, -0.5f, 1.f, 0.f},    {0.f, 0.25f, 0.f, -1.25f, 0.f, 1.f}};
            return t[xi % 6][i % 6];
        }
        constexpr float t[8][8] = {{-1.f, 0.f, 5.25f, 0.f, -5.25f, 0.f, 1.f, 0.f},     {0.f, 1.f, 1.f, -4.25f, -4.25f, 1.f, 1.f, 0.f},
                                   {0.f, -1.f, 1.f, 4.25f, -4.25f, -1.f, 1.f, 0.f},    {0.f, 2.f, 4.f, -2.5f, -5.f, 0.5f, 1.f, 0.f},
                                   {0.f, -2.f, 4.f, 2.5f, -5.f, -0.5f, 1.f, 0.f},      {0.f, 0.5f, 0.25f, -2.5f, -1.25f, 2.f, 1.f, 0.f},
                                   {0.f, -0.5f, 0.25f, 2.5f, -1.25f, -2.f, 1.f, 0.f},  {0.f, -1.f, 0.f, 5.25f, 0.f, -5.25f, 0.f, 1.f}};
        return t[xi % 8][i % 8];
    }
    static constexpr float at(int o, int xi) {
        if (MO == 2) {
            constexpr float t[2][6] = {{1.f, 1.f, 1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, 0.5f, -0.5f, 1.f}};
            return t[o % 2][xi % 6];
        }
        constexpr float t[4][8] = {{1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 0.f},          {0.f, 1.f, -1.f, 0.5f, -0.5f, 2.f, -2.f, 0.f},
                                   {0.f, 1.f, 1.f, 0.25f, 0.25f, 4.f, 4.f, 0.f},      {0.f, 1.f, -1.f, 0.125f, -0.125f, 8.f, -8.f, 1.f}};
        return t[o % 4][xi % 8];
    }
};

#define GT_WINO_OOB 0x80000000u
#define WT 512             // threads: 8 waves as 2 (tile rows) x 4 (columns), each a 32 x 32 MFMA tile

// One slice (32 input channels), REQUESTED: this thread's A element (tile row f >> 3, channel quad f & 7) of every one of the
// ALPHA transform-domain GEMMs is a combination of the same ALPHA input rows ("taps"), which are loaded raw into d[] ONCE per
// slice.  Every gather load is an unconditional buffer load: a row outside [0, len) gets an out-of-range offset and reads as
// zero (SAME padding / masked mode; also a slice past the last one) -- no branches, so the number of loads in flight is known
// exactly at every later point and the waits the compiler inserts are counted, not vmcnt(0).
template <int MO>
__device__ __forceinline__ void wino_issue_taps(const ConvGemmArgs& A, __amdgpu_buffer_rsrc_t rs_x, const uint32_t voff, const int first, const int len,
                                                const int c0, const bool live, float4 (&d)[Wino<MO>::ALPHA]) {
#pragma unroll
    for (int tap = 0; tap < Wino<MO>::ALPHA; ++tap) {
        const int ts = first + tap;
        const uint32_t vo = (live && ts >= 0 && ts < len) ? voff + (uint32_t)(tap * A.Cin * 4) : GT_WINO_OOB;
        const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)vo, c0 * 4, 0);
        __builtin_memcpy(&d[tap], &t, 16);
    }
}
// ... and this thread's two pieces of the B slice U_XI[c0 .. c0 + 32): k rows tid >> 5 and 16 + (tid >> 5), quad tid & 31.  Buffer
// loads with the (XI, slice) part of the address in the SCALAR offset: the two per-thread offsets are the same for every
// step, so nothing per step lives in vector registers (as 64-bit pointers, hoisted out of the unrolled loop, the 16 steps'
// addresses cost 32 registers -- spilled, and every scratch reload waits for vmcnt(0), i.e. for all the prefetches).
__device__ __forceinline__ void wino_issue_b(const ConvGemmArgs& A, __amdgpu_buffer_rsrc_t rs_u, const uint32_t vb0, const uint32_t vb1, const int xi,
                                             const int c0, float4& rb0, float4& rb1) {
    const int so = (xi * A.wino_cin + c0) * A.N * 4;
    const auto t0 = __builtin_amdgcn_raw_buffer_load_b128(rs_u, (int)vb0, so, 0);
    const auto t1 = __builtin_amdgcn_raw_buffer_load_b128(rs_u, (int)vb1, so, 0);
    __builtin_memcpy(&rb0, &t0, 16);
    __builtin_memcpy(&rb1, &t1, 16);
}
// ... and TRANSFORMED once it has arrived: V_XI = sum_tap BT[XI][tap] d[tap]
template <int MO, int XI>
__device__ __forceinline__ float4 wino_xform(const float4 (&d)[Wino<MO>::ALPHA]) {
    f32x2 lo = {0.f, 0.f}, hi = {0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < Wino<MO>::ALPHA; ++tap) {
        const float cf = Wino<MO>::bt(XI, tap);
        // (explicit fma: left to the compiler, which products are contracted differs from one instantiation to the next; PACKED fma:
        // the fp32 matrix instructions run on the vector ALU's own multipliers -- tools/mfma_rate.hip: a wave's VALU instructions
        // add to its MFMA time instead of hiding under it -- so the transform's instruction count is paid in full)
        if (cf != 0.f) { lo = __builtin_elementwise_fma((f32x2){cf, cf}, (f32x2){d[tap].x, d[tap].y}, lo); hi = __builtin_elementwise_fma((f32x2){cf, cf}, (f32x2){d[tap].z, d[tap].w}, hi); }
    }
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}

