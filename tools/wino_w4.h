// The Winograd F(4,5) / F(2,5) convolution (gst_tacotron_amd/csrc/gemm_conv.hip, gt_conv_wino5_kernel) in the form round 2's and round 3's
// reviews asked for: ONE workgroup of 4 waves per CU, each wave alone on its SIMD with the whole 512-register file.  Measured
// (tools/wino_bench.hip, profiles/r04_wino_w4.txt): bitwise the 8-wave kernel's results and 5-8 % SLOWER, so it is a tool, not the
// product's kernel.  Included by tools/wino_bench.hip AFTER gemm_conv.hip (it uses that file's Wino<>, ConvGemmArgs, BK, GT_WINO_OOB).
#pragma once
// The same algorithm at ONE workgroup of 4 waves per CU, each wave alone on its SIMD with the whole 512-register file: a wave owns 64 tiles
// x 32 columns = TWO 32 x 32 accumulator tiles per transform-domain GEMM (2 x ALPHA x 16 accumulator registers -- 256 for F(4,5), the
// AGPR half of the file), so
//   (a) every B operand word read from LDS feeds two MFMAs, and the two tiles' MFMA chains are independent and alternate: one wave keeps the
//       matrix pipe issuing back to back without a second wave on the SIMD;
//   (b) TWO transform-domain GEMMs are staged per barrier (half the barriers, four waves at each instead of eight);
//   (c) a thread gathers the raw rows of two ADJACENT tiles, which overlap: 12 rows instead of 2 x 8 for F(4,5) (8 instead of 12 for
//       F(2,5)) -- a quarter less through the L1, which is what bound the 8-wave kernel (EXPERIMENTS item 10).  Tiles are paired inside an
//       utterance (the tile count per utterance is rounded up to even; the extra tile reads zeros and stores nothing);
//   (d) the B slices go from global memory STRAIGHT into LDS (buffer_load_dwordx4 ... lds: no registers, no ds_write) -- as inline asm,
//       because the compiler orders every later LDS read behind a DMA it knows about (vmcnt(0) in front of the MFMAs' operand reads of
//       the OTHER stage); the asm's own waits are counted: the only younger requests at the barrier are the next slice's taps.
// LDS: 2 stages x 2 GEMMs x (A 32 x 65 + B 32 x 128) floats = 96.5 KB, dynamic.  Same workgroup tile, grid shape, XCD mapping and summation
// order per output as gt_conv_wino5_kernel: results are bitwise the same.
#define WT4 256
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
__device__ __forceinline__ void gt_lds_dma16(__amdgpu_buffer_rsrc_t rs, const uint32_t lds_addr, const uint32_t voff, const uint32_t soff) {
    // M0 = the wave's LDS base; lane l's 16 bytes land at base + 16 l.  (M0 is a scratch register for the compiler: it sets it right before
    // each of its own uses.)
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_addr), "v"(voff), "s"(rs), "s"(soff) : "memory", "m0");
}
#pragma clang diagnostic pop

// raw rows first .. first + NT - 1 of this thread's tile pair, channel quad c0 / 4 + (tid & 7): unconditional loads (see wino_issue_taps)
template <int NT, int R0, int R1>
__device__ __forceinline__ void wino_issue_rows(const ConvGemmArgs& A, __amdgpu_buffer_rsrc_t rs_x, const uint32_t voff, const int first, const int len,
                                                const int c0, const bool live, float4 (&d)[NT]) {
#pragma unroll
    for (int tap = R0; tap < R1; ++tap) {
        const int ts = first + tap;
        const uint32_t vo = (live && ts >= 0 && ts < len) ? voff + (uint32_t)(tap * A.Cin * 4) : GT_WINO_OOB;
        const auto t = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)vo, c0 * 4, 0);
        __builtin_memcpy(&d[tap], &t, 16);
    }
}
template <int MO, int XI, int OFF, int NT>
__device__ __forceinline__ float4 wino_xform_at(const float4 (&d)[NT]) {
    f32x2 lo = {0.f, 0.f}, hi = {0.f, 0.f};
#pragma unroll
    for (int tap = 0; tap < Wino<MO>::ALPHA; ++tap) {
        const float cf = Wino<MO>::bt(XI, tap);
        if (cf != 0.f) {
            lo = __builtin_elementwise_fma((f32x2){cf, cf}, (f32x2){d[OFF + tap].x, d[OFF + tap].y}, lo);
            hi = __builtin_elementwise_fma((f32x2){cf, cf}, (f32x2){d[OFF + tap].z, d[OFF + tap].w}, hi);
        }
    }
    return make_float4(lo.x, lo.y, hi.x, hi.y);
}

template <int MO>
__global__ __launch_bounds__(WT4) void gt_conv_wino5_w4_kernel(ConvGemmArgs A, const float* __restrict__ U) {
    constexpr int AL = Wino<MO>::ALPHA, NP = AL / 2, NT = AL + MO, RPS = NT / (NP - 1);     // RPS: rows requested per pair step
    constexpr int BMP = 64, BN = 128, LDA = BMP + 1, ASZ = BK * LDA, BSZ = BK * BN;
    extern __shared__ __attribute__((aligned(16))) float w4_lds[];
    float* const Bs = w4_lds;                            // [stage][gemm of the pair][BK][BN]  (ds_read_b32 banks per 32-lane half: no padding needed)
    float* const As = w4_lds + 4 * BSZ;                  // [stage][gemm of the pair][BK][LDA]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Pu = ((A.T + MO - 1) / MO + 1) & ~1;       // tiles per utterance, even
    const int Ptot = A.B * Pu;
    const int ncb = (A.N + BN - 1) / BN;
    const int wi = blockIdx.x >> 3;
    const int rb = (wi / ncb) * 8 + (blockIdx.x & 7), cb = wi % ncb;      // (XCD-aware mapping: see gt_conv_wino5_kernel)
    if (rb * BMP >= Ptot) return;
    const int p0 = rb * BMP, n0 = cb * BN;
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A.x), 0, (int)((size_t)A.B * A.T * A.Cin * 4), 0x00020000);
    // this thread's A elements: tiles p0 + 2 q and p0 + 2 q + 1 (q = tid >> 3), kept in LDS rows q and 32 + q; channel quad tid & 7
    int first, len;
    uint32_t voff;
    {
        const int p = p0 + 2 * (tid >> 3);
        const bool ok = p < Ptot;
        const int pp = ok ? p : 0;
        const int b = pp / Pu;
        first = MO * (pp - b * Pu) - 2;
        len = ok ? (A.row_len ? min(A.T, A.row_len[b]) : A.T) : 0;
        voff = (uint32_t)(((int64_t)b * A.T + first) * A.Cin + (tid & 7) * 4) * 4u;
    }
    f32x16 M[AL][2];
#pragma unroll
    for (int xi = 0; xi < AL; ++xi)
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) M[xi][r][e] = 0.f;
    const int kh = lane >> 5, l31 = lane & 31;
    const int nsl = A.wino_cin / BK;                      // even, >= 4 (gt_conv_wino5_applies)
    const auto rs_u = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(U), 0, (int)((size_t)AL * A.wino_cin * A.N * 4), 0x00020000);
    // B: one DMA instruction moves two k rows (lanes 0-31 / 32-63) x 128 columns; a wave moves rows 8 wave .. 8 wave + 7 of each GEMM's slice
    const uint32_t vbq = (uint32_t)((kh * A.N + min(n0 + l31 * 4, A.N - 4)) * 4);                  // (columns past N are never stored)
    const uint32_t lds_b = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)Bs;
    auto dma_b = [&](const int sg, const int xi, const int c0) {
#ifdef GT_W4_NO_DMA
        return;
#endif
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row0 = 8 * wave + 2 * i;
            gt_lds_dma16(rs_u, lds_b + (uint32_t)((sg * BSZ + row0 * BN) * 4), vbq, (uint32_t)(((xi * A.wino_cin + c0 + row0) * A.N) * 4));
        }
    };
    // Per-thread LDS indices, made opaque at the top of every pair step: everything else in an LDS address is a compile-time constant
    // that fits the instructions' 16-bit offset field (B region 64 KB, A region 33 KB from its own base).  Left visible, the ~100
    // (stage, GEMM, k row) addresses are hoisted out of the loop into registers -- and spilled.
    float w4_sink = 0.f;                                 // (ablations only)
    int ix_ra = kh * LDA + l31, ix_rb = kh * BN + wave * 32 + l31, ix_wa = (tid & 7) * 4 * LDA + (tid >> 3);
    auto store_a = [&](const int sg, const int e, const float4 ra) {
#ifdef GT_W4_NO_XSTORE
        w4_sink += ra.x + ra.y + ra.z + ra.w;
        return;
#endif
        float* const a = As + ix_wa;
        a[sg * ASZ + 0 * LDA + 32 * e] = ra.x;
        a[sg * ASZ + 1 * LDA + 32 * e] = ra.y;
        a[sg * ASZ + 2 * LDA + 32 * e] = ra.z;
        a[sg * ASZ + 3 * LDA + 32 * e] = ra.w;
    };
    // A quarter (4 k-pairs) of BOTH GEMMs of the pair on this wave's two tiles: 24 operand words, 16 MFMAs on FOUR accumulators in turn (a
    // dependent 32x32x2 MFMA can issue ~190 cycles after its predecessor, three times the pipe's 64: two alternating chains per wave
    // measured 97 cycles per MFMA, tools/wino_bench).
#define W4_READ(O, ST, Q) do {                                                                                     \
        _Pragma("unroll") for (int g_ = 0; g_ < 2; ++g_)                                                           \
            _Pragma("unroll") for (int kp = 0; kp < 4; ++kp) {                                                     \
                const int k2 = ((Q) * 4 + kp) * 2;                                                                 \
                O[g_ * 12 + kp] = As[ix_ra + ((ST) * 2 + g_) * ASZ + k2 * LDA];                                    \
                O[g_ * 12 + 4 + kp] = As[ix_ra + ((ST) * 2 + g_) * ASZ + k2 * LDA + 32];                           \
                O[g_ * 12 + 8 + kp] = Bs[ix_rb + ((ST) * 2 + g_) * BSZ + k2 * BN];                                 \
            }                                                                                                      \
    } while (0)
#ifdef GT_W4_NO_MFMA            // (ablation switches of tools/wino_bench.hip; the product compiles the plain forms)
#define W4_MMA(O, J) do { _Pragma("unroll") for (int kp = 0; kp < 24; ++kp) w4_sink += O[kp]; } while (0)
#else
#define W4_MMA(O, J) do {                                                                                          \
        _Pragma("unroll") for (int kp = 0; kp < 4; ++kp) {                                                         \
            M[2 * (J)][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(O[kp], O[8 + kp], M[2 * (J)][0], 0, 0, 0);        \
            M[2 * (J)][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(O[4 + kp], O[8 + kp], M[2 * (J)][1], 0, 0, 0);    \
            M[2 * (J) + 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(O[12 + kp], O[20 + kp], M[2 * (J) + 1][0], 0, 0, 0); \
            M[2 * (J) + 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(O[16 + kp], O[20 + kp], M[2 * (J) + 1][1], 0, 0, 0); \
        }                                                                                                          \
    } while (0)
#endif
    // Pair step (J of slice s_), operands in LDS stage CUR: DMA the B slices of the NEXT pair into the other stage, then request RPS of the
    // next slice's rows (all but the slice's last step: spread, because all CUs asking for a whole slice's rows at the same moment is a
    // 12 MB burst that took ~2 us to serve -- one pair step in four stalled on it); then four REGIONS, each = the operand reads of the next
    // region + the 16 MFMAs of this one + one quarter of the next pair's A operands (one GEMM, one of the two tiles) transformed and
    // stored.  The scheduler interleaves inside a region (the VALU / LDS work lands between the MFMAs) but not across (left free over the
    // whole step it hoists until it spills: with one wave per SIMD a scratch reload is an exposed vmcnt(0)).  Last: wait for the DMA
    // (the only younger requests are this step's RPS row loads: requests return in order, so a row load has until the end of the NEXT
    // step); barrier.  Stages are compile-time (SP = slice parity in the unrolled pair).
    float4 dE[NT], dO[NT];
    float o0[24], o1[24];
#ifdef GT_WINO_STAMPS
    const bool stamp_on = blockIdx.x == 0 && tid == 0;
    int nstamp = 0;
#define W4_STAMP_STEP() do { if (stamp_on && nstamp < 1023) gt_wino_stamp[nstamp++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define W4_STAMP_STEP() do { } while (0)
#endif
#ifdef GT_W4_CLUSTER           // (tools/wino_bench.hip: MFMAs of a region back to back, then the transform, not interleaved)
#define W4_SEP() __builtin_amdgcn_sched_barrier(0)
#else
#define W4_SEP() do { } while (0)
#endif
#if defined(GT_WINO_STAMPS) && defined(GT_W4_FINE)   // (tools/wino_bench.hip: stamps inside the pair steps of workgroup 0, wave 0)
#define W4_FINE(k) do { __builtin_amdgcn_sched_barrier(0); if (stamp_on && nstamp < 60) gt_wino_stamp[512 + 8 * nstamp + (k)] = __builtin_amdgcn_s_memtime(); \
                        __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define W4_FINE(k) do { } while (0)
#endif
#define W4_PAIR(J, SP, DCUR, DNXT, s_)                                                                             \
    {                                                                                                             \
        constexpr int CUR = (((SP) * NP + (J)) & 1), NXT = CUR ^ 1, J1 = ((J) + 1) % NP;                           \
        const int c1_ = min((s_) + ((J) + 1 >= NP ? 1 : 0), nsl - 1) * BK;                                         \
        W4_FINE(0);                                                                                                \
        asm volatile("" : "+v"(ix_ra), "+v"(ix_rb), "+v"(ix_wa));                                                  \
        W4_READ(o0, CUR, 0);                                                                                       \
        W4_FINE(1);                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        dma_b(NXT * 2, 2 * J1, c1_);                                                                               \
        dma_b(NXT * 2 + 1, 2 * J1 + 1, c1_);                                                                       \
        if constexpr ((J) + 1 < NP) wino_issue_rows<NT, RPS * (J), RPS * (J) + RPS>(A, rs_x, voff, first, len, min((s_) + 1, nsl - 1) * BK, (s_) + 1 < nsl, DNXT); \
        W4_READ(o1, CUR, 1);                                                                                       \
        W4_MMA(o0, J);                                                                                             \
        W4_SEP();                                                                                                  \
        if constexpr ((J) + 1 < NP) store_a(NXT * 2, 0, wino_xform_at<MO, 2 * J1, 0, NT>(DCUR));                   \
        else store_a(NXT * 2, 0, wino_xform_at<MO, 0, 0, NT>(DNXT));                                               \
        W4_FINE(2);                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        W4_READ(o0, CUR, 2);                                                                                       \
        W4_MMA(o1, J);                                                                                             \
        W4_SEP();                                                                                                  \
        if constexpr ((J) + 1 < NP) store_a(NXT * 2, 1, wino_xform_at<MO, 2 * J1, MO, NT>(DCUR));                  \
        else store_a(NXT * 2, 1, wino_xform_at<MO, 0, MO, NT>(DNXT));                                              \
        W4_FINE(3);                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        W4_READ(o1, CUR, 3);                                                                                       \
        W4_MMA(o0, J);                                                                                             \
        W4_SEP();                                                                                                  \
        if constexpr ((J) + 1 < NP) store_a(NXT * 2 + 1, 0, wino_xform_at<MO, 2 * J1 + 1, 0, NT>(DCUR));           \
        else store_a(NXT * 2 + 1, 0, wino_xform_at<MO, 1, 0, NT>(DNXT));                                           \
        W4_FINE(4);                                                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        W4_MMA(o1, J);                                                                                             \
        W4_SEP();                                                                                                  \
        if constexpr ((J) + 1 < NP) store_a(NXT * 2 + 1, 1, wino_xform_at<MO, 2 * J1 + 1, MO, NT>(DCUR));          \
        else store_a(NXT * 2 + 1, 1, wino_xform_at<MO, 1, MO, NT>(DNXT));                                          \
        W4_FINE(5);                                                                                                   \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
        W4_FINE(6);                                                                                                   \
        __syncthreads();                                                                                          \
        W4_STAMP_STEP();                                                                                          \
    }
#define W4_SLICE(SP, DCUR, DNXT, s_)                                                                               \
    W4_PAIR(0, SP, DCUR, DNXT, s_) W4_PAIR(1, SP, DCUR, DNXT, s_) W4_PAIR(2, SP, DCUR, DNXT, s_)                   \
    if constexpr (NP == 4) { W4_PAIR(3 % NP, SP, DCUR, DNXT, s_) }
    // prologue: rows of slice 0, B of pair 0, operands of pair 0 into stage 0
    dma_b(0, 0, 0);
    dma_b(1, 1, 0);
    wino_issue_rows<NT, 0, NT>(A, rs_x, voff, first, len, 0, true, dE);
    store_a(0, 0, wino_xform_at<MO, 0, 0, NT>(dE)); store_a(0, 1, wino_xform_at<MO, 0, MO, NT>(dE));
    store_a(1, 0, wino_xform_at<MO, 1, 0, NT>(dE)); store_a(1, 1, wino_xform_at<MO, 1, MO, NT>(dE));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < nsl; s += 2) {
        W4_SLICE(0, dE, dO, s)
        W4_SLICE(1, dO, dE, s + 1)
    }
#undef W4_SLICE
#undef W4_PAIR
#undef W4_MMA
#undef W4_READ
#undef W4_STAMP_STEP
#undef W4_SEP
#undef W4_FINE

    if (w4_sink == 12345.678f) A.out[tid] = w4_sink;     // (never; keeps the ablations' operands alive)
    // epilogue (output transform), both tiles; C/D layout as in gt_conv_wino5_kernel; accumulator tile r, row j <-> tile p0 + 2 j + r
    const int n = n0 + wave * 32 + l31;
    if (n < A.N) {
        const float sc = A.scale ? A.scale[n] : 1.f;
        const float sh = A.shift ? A.shift[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int p = p0 + 2 * ((e & 3) + 8 * (e >> 2) + 4 * kh) + r;
                if (p >= Ptot) continue;
                const int b = p / Pu, t0 = MO * (p - b * Pu);
#pragma unroll
                for (int o = 0; o < MO; ++o) {
                    const int t = t0 + o;
                    if (t >= A.T) continue;
                    const int64_t m = (int64_t)b * A.T + t;
                    float y = 0.f;
#pragma unroll
                    for (int xi = 0; xi < AL; ++xi)
                        if (Wino<MO>::at(o, xi) != 0.f) y += Wino<MO>::at(o, xi) * M[xi][r][e];
                    float v = y * sc + sh;
                    if (A.rowbias) v += A.rowbias[(int64_t)b * A.N + n];
                    if (A.act == ACT_RELU) v = fmaxf(v, 0.f);
                    else if (A.act == ACT_TANH) v = gt_tanh(v);
                    if (A.res) v += A.res[m * A.ldo + n];
                    A.out[m * A.ldo + n] = v;
                }
            }
    }
}
constexpr size_t w4_lds_bytes() { return (size_t)4 * (BK * (64 + 1) + BK * 128) * 4; }
// tiles of the 4-wave kernel's grid: per utterance an even number
static inline int w4_tiles(const ConvGemmArgs& a, int mo) { return a.B * ((((a.T + mo - 1) / mo) + 1) & ~1); }

hipError_t gt_conv_wino5_w4_init() {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gt_conv_wino5_w4_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)w4_lds_bytes());
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gt_conv_wino5_w4_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)w4_lds_bytes());
}

