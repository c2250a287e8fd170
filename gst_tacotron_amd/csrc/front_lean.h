// Lean utterance path of the fused decoder front end (dec_front.hip), for the shapes the decode loop runs from step 1 on:
// prenet-0 pre-activations already there (Z0), GEMV plans that divide evenly, dropout keep decisions either injected or
// hashed from the seed, sigmoid noise injected / pre-generated, T_v <= 1024.  Same arithmetic, same summation order as the
// general kernel (reference Taco2.py:262-283, Steps.py:122-229); what changes is how the loads are issued.
//
// Why it exists (ISA of the general kernel, round 2): the compiler's s_waitcnt bookkeeping is STATIC.  A load inside a
// branch -- the weight rows skipped for dropped prenet outputs, `if (P.mask0) ...`, `if (t < Tv) ...` -- makes the number of
// loads in flight unknown at every later point, so every wait behind it degrades to `vmcnt(0)`: the small operands'
// LDS writes waited for all 16 weight rows, and a register copy the allocator placed behind the second conditional weight
// load made the waves that kept that row wait a full memory latency with 2 of their 16 requests issued (their workgroup's
// first barrier then waited ~1 us for them).  Here EVERY vector load is an unconditional buffer load: what must not be
// read (a skipped weight row, a row past T_v, an absent mask) gets an out-of-range offset or an empty descriptor, for which
// the hardware returns zero without touching memory.  Loads are counted exactly, waits are `vmcnt(N)`, nothing is copied.
#pragma once

#define GT_LSTAMP(slot)                                                                         \
    do {                                                                                        \
        if (P.dbg && stamp_wg && threadIdx.x == 0) P.dbg[slot] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)

// Preconditions (checked on the host, front_lean_ok): P.z0 != NULL; FT % (P1/4) == 0, P0 == 16 * FT / (P1/4);
// FT % (A/4) == 0, P1 == 8 * FT / (A/4); P0, P1, A, T_v <= FT; drop_rate == 0 or (both masks given) or keep_hash;
// sigmoid_noise == 0 or noise given.
// `b`: utterance (row) of this workgroup.
template <int L, int NP>
__device__ __forceinline__ void gt_front_lean(const DecFrontArgs& P, float* smem, const int b, const bool stamp_wg) {
    constexpr int A = 4 * L * NP;
    constexpr int ROWS = FT / L;
    constexpr int LD = A + 4;
    constexpr int CPARTS = FT / A;
    const int tid = threadIdx.x, lane = tid & 63;
    GT_LSTAMP(12);
    const int P0 = P.P0, P1 = P.P1, TvFull = P.Tv;
    const bool drop = P.drop_rate > 0.f;
    const bool hashed = drop && P.keep_hash != 0;
    // scalar loads through the constant address space (a plain load of data an earlier kernel wrote compiles to a vector
    // load + readfirstlane, waited for on the spot)
    uint64_t kseed = 0;
    if (hashed) kseed = *(const __attribute__((address_space(4))) uint64_t*)P.seed_ptr;
    const float sbias = *(const __attribute__((address_space(4))) float*)P.score_bias;
    int Tv = TvFull;
    if (P.tok_len) Tv = max(1, min(TvFull, (int)*(const __attribute__((address_space(4))) int32_t*)(P.tok_len + b)));

    // LDS carve (floats), the general kernel's: xs | y0 | y1 | q | v | score | prev | align | red | biases, keep-scales,
    // noise | partial[4 FT] | tile[ROWS][LD]
    int mx = P.mel > P0 ? P.mel : P0; if (P1 > mx) mx = P1;
    float* xs = smem;
    float* y0 = xs + ((mx + 3) & ~3);
    float* y1 = y0 + P0;
    float* qs = y1 + P1;
    float* vs = qs + A;
    float* sc = vs + A;
    float* pv = sc + ((TvFull + 3) & ~3);
    float* al = pv + ((TvFull + 3) & ~3);
    float* red = al + ((TvFull + 3) & ~3);
    float* sb0 = red + FT;
    float* sb1 = sb0 + P0;
    float* sbq = sb1 + P1;
    float* sk0 = sbq + A;
    float* sk1 = sk0 + P0;
    float* snz = sk1 + P1;
    float* partial = snz + ((TvFull + 3) & ~3);
    float* tile = partial + 4 * FT;

    // ---- descriptors (wave-uniform: kernel arguments and blockIdx only); absent operands get an empty one
    const auto rs_z0 = gt_rsrc(P.z0 + (size_t)b * P0, (uint32_t)P0 * 4u);
    const auto rs_v = gt_rsrc(P.v, (uint32_t)A * 4u);
    const auto rs_pv = gt_rsrc(P.prev ? P.prev + (size_t)b * P.ldprev : P.v, P.prev ? (uint32_t)Tv * 4u : 0u);
    const auto rs_b1 = gt_rsrc(P.b1, (uint32_t)P1 * 4u);
    const auto rs_bq = gt_rsrc(P.bq, (uint32_t)A * 4u);
    const bool inj = drop && !hashed;        // keep decisions injected (or pre-generated) as 0/1 tensors
    const auto rs_m0 = gt_rsrc(inj ? P.mask0 + (size_t)b * P0 : P.v, inj ? (uint32_t)P0 * 4u : 0u);
    const auto rs_m1 = gt_rsrc(inj ? P.mask1 + (size_t)b * P1 : P.v, inj ? (uint32_t)P1 * 4u : 0u);
    const bool noisy = P.sigmoid_noise > 0.f;
    const auto rs_nz = gt_rsrc(noisy ? P.noise + (size_t)b * P.ldnoise : P.v, noisy ? (uint32_t)Tv * 4u : 0u);
    const auto rs_w1 = gt_rsrc(P.w1, (uint32_t)(P0 * P1) * 4u);
    const auto rs_wq = gt_rsrc(P.wq, (uint32_t)(P1 * A) * 4u);
    const auto rs_pm = gt_rsrc(P.pm + (size_t)b * TvFull * A, (uint32_t)(Tv * A) * 4u);     // rows >= T_v read as zero

    // ---- small operands first (loads return in issue order); indices past an operand's end are out of range = 0
    const uint32_t t4 = (uint32_t)tid * 4u;
    const float in_x = gt_bload1(rs_z0, t4);
    const float in_v = gt_bload1(rs_v, t4);
    float in_p = gt_bload1(rs_pv, t4);
    const float t_b1 = gt_bload1(rs_b1, t4);
    const float t_bq = gt_bload1(rs_bq, t4);
    float t_k0 = gt_bload1(rs_m0, t4);
    float t_k1 = gt_bload1(rs_m1, t4);
    const float t_nz = gt_bload1(rs_nz, t4);
    GT_LSTAMP(13);

    // Throughput mode at rate 0.5 (keep_hash): wave w owns prenet-1 rows 16w..16w+15 (mask 0) and query rows 16w..16w+15
    // (mask 1, 8 per lane half); the rows its keep decisions zero are not requested
    uint32_t rb1 = 0xFFFFu, rbq = 0xFFu;
    if (hashed) {
        const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        rb1 = (gt_keep_word(kseed, P.rng_step, 0u, (uint32_t)b, wave >> 1) >> ((wave & 1u) * 16u)) & 0xFFFFu;
        const uint32_t q16 = (gt_keep_word(kseed, P.rng_step, 1u, (uint32_t)b, wave >> 1) >> ((wave & 1u) * 16u)) & 0xFFFFu;
        rbq = (q16 >> ((lane >> 5) * 8)) & 0xFFu;
    }
    GT_LSTAMP(14);
    // ---- all of prenet 1's weights: lane = 4 consecutive output columns, 16 consecutive k rows
    const GemvPlan g1 = make_plan(P0, P1, tid);
    const GemvPlan g2 = make_plan(P1, A, tid);
    const uint32_t off1 = (uint32_t)((g1.kp * 16 * P1 + g1.cg * 4) * 4);
    const int row = tid / L, li = tid % L;
    float4 rq[8], v0[NP];
    float4 r1[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r1[i] = gt_bload4(rs_w1, ((rb1 >> i) & 1u) ? off1 : GT_OOB, (uint32_t)(i * P1 * 4));
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    GT_LSTAMP(15);

    // ---- keep-scales (hash: a few VALU ops) and the small operands into LDS while the weights are in flight
    if (!P.prev) in_p = tid == 0 ? 1.f : 0.f;
    if (hashed) {
        t_k0 = gt_drop_keep(kseed, P.rng_step, 0u, (uint32_t)b, (uint32_t)min(tid, P0 - 1), (uint32_t)P0, P.drop_rate);
        t_k1 = gt_drop_keep(kseed, P.rng_step, 1u, (uint32_t)b, (uint32_t)min(tid, P1 - 1), (uint32_t)P1, P.drop_rate);
    }
    if (drop) { t_k0 *= P.drop_scale; t_k1 *= P.drop_scale; }
    else { t_k0 = 1.f; t_k1 = 1.f; }
    if (tid < P1) { sb1[tid] = t_b1; sk1[tid] = t_k1; }
    if (tid < A) { sbq[tid] = t_bq; vs[tid] = in_v; }
    if (tid < Tv) { snz[tid] = P.sigmoid_noise * t_nz; pv[tid] = in_p; }
    if (tid < P0) y0[tid] = fmaxf(in_x, 0.f) * t_k0;
    GT_LSTAMP(0);
    __syncthreads();
    GT_LSTAMP(1);
    GT_LSTAMP(2);

    // ---- prenet layer 1.  The per-CU load pipe (64 B/clk) is what the chain waits for, so every request goes out as early
    // as its landing registers exist: the query weights are requested as soon as the FIRST half of W1 has been consumed
    // (they take its registers, and queue right behind W1's second half), the processed-memory rows after the second.
    {
        float x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = y0[g1.kp * 16 + i];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            gt_fma4(acc, x[i], r1[i]);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        {
            const uint32_t offq = (uint32_t)((g2.kp * 8 * A + g2.cg * 4) * 4);
#pragma unroll
            for (int i = 0; i < 8; ++i) rq[i] = gt_bload4(rs_wq, ((rbq >> i) & 1u) ? offq : GT_OOB, (uint32_t)(i * A * 4));
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 8; i < 16; ++i) {
            gt_fma4(acc, x[i], r1[i]);
        }
        gemv_store(g1, P1, acc, partial);
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    {
        const uint32_t offr = (uint32_t)((row * A + 4 * li) * 4);
#pragma unroll
        for (int j = 0; j < NP; ++j) v0[j] = gt_bload4(rs_pm, offr, (uint32_t)(16 * L * j));
    }
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (tid < P1) {
        const float v = fmaxf(reduce_partial(partial, g1.kparts, P1, tid) + sb1[tid], 0.f) * sk1[tid];
        y1[tid] = v;
        P.xa[gt_blk_off(b, tid, P.MT)] = v;          // LSTM-1 input (blocked), k in [0, P1)
        if (P.xah) P.xah[gt_blk_off_h(b, tid, P.MT)] = gt_bf16_bits(v);
    }
    __syncthreads();
    GT_LSTAMP(3);
    // ---- query projection
    {
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = y1[g2.kp * 8 + i];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            gt_fma4(acc, x[i], rq[i]);
        }
        gemv_store(g2, A, acc, partial);
    }
    __syncthreads();
    if (tid < A) qs[tid] = reduce_partial(partial, g2.kparts, A, tid) + sbq[tid];
#pragma unroll
    for (int j = 0; j < NP; ++j) *reinterpret_cast<float4*>(tile + row * LD + 4 * (li + L * j)) = v0[j];
    __syncthreads();
    GT_LSTAMP(4);

    // ---- scores: L lanes per memory row read their NP 16-byte pieces of the LDS tile
    const int nchunks = (Tv + ROWS - 1) / ROWS;
    auto load_rows = [&](float4 (&v)[NP], int c) {
        const uint32_t o = (uint32_t)(((c * ROWS + row) * A + 4 * li) * 4);
#pragma unroll
        for (int j = 0; j < NP; ++j) v[j] = gt_bload4(rs_pm, o, (uint32_t)(16 * L * j));
    };
    auto store_rows = [&](const float4 (&v)[NP]) {
#pragma unroll
        for (int j = 0; j < NP; ++j) *reinterpret_cast<float4*>(tile + row * LD + 4 * (li + L * j)) = v[j];
    };
    for (int c = 0; c < nchunks; ++c) {
        if (c > 0) {                                   // Tv > ROWS: stream further chunks through the one tile
            float4 v[NP];
            load_rows(v, c);
            __syncthreads();
            store_rows(v);
            __syncthreads();
        }
        const int t = c * ROWS + row;
        f32x2 s2 = {0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const int a0 = 4 * (li + L * j);
            const float4 m4 = *reinterpret_cast<const float4*>(tile + row * LD + a0);
            const float4 q4 = *reinterpret_cast<const float4*>(qs + a0);
            const float4 w4 = *reinterpret_cast<const float4*>(vs + a0);
            s2 = __builtin_elementwise_fma(f32x2{w4.x, w4.y}, gt_tanh2(f32x2{q4.x, q4.y} + f32x2{m4.x, m4.y}), s2);
            s2 = __builtin_elementwise_fma(f32x2{w4.z, w4.w}, gt_tanh2(f32x2{q4.z, q4.w} + f32x2{m4.z, m4.w}), s2);
        }
        float s = s2.x + s2.y;
        s = gt_row_sum<L>(s);
        if (li == 0 && t < Tv) sc[t] = s + sbias;
    }
    __syncthreads();
    GT_LSTAMP(5);
    // ---- noise + sigmoid + alignment
    if (P.type == GSTTACO_ATT_SMA) {
        if (tid < Tv) {
            const int t = tid;
            float v = pv[t] * gt_sigmoid(sc[t] + (noisy ? snz[t] : 0.f));
            // (explicit fma: `a*b + c*d` may contract around either product, and this kernel, the general one and the persistent decode
            // kernel must round alike -- their outputs are compared bitwise)
            if (t > 0) v = __builtin_fmaf(pv[t - 1], 1.f - gt_sigmoid(sc[t - 1] + (noisy ? snz[t - 1] : 0.f)), v);
            al[t] = v;
        }
    } else {
        if (tid < Tv) {
            float s = sc[tid];
            if (noisy) s += snz[tid];
            sc[tid] = gt_sigmoid(s);
        }
        __syncthreads();
    }
    if (P.type != GSTTACO_ATT_SMA && tid < 64) {
        const int per = (Tv + 63) / 64;
        const int t0 = lane * per, t1 = min(Tv, t0 + per);
        float run = 0.f;
        for (int t = t0; t < t1; ++t) run += logf(fminf(fmaxf(1.f - sc[t], 1.17549435e-38f), 1.f));
        float base = front_wave_incl_scan(run, lane) - run;
        for (int t = t0; t < t1; ++t) {
            const float lg = logf(fminf(fmaxf(1.f - sc[t], 1.17549435e-38f), 1.f));
            al[t] = expf(base);
            base += lg;
        }
        run = 0.f;
        for (int t = t0; t < t1; ++t) run += pv[t] / fminf(fmaxf(al[t], 1e-10f), 1.f);
        base = front_wave_incl_scan(run, lane) - run;
        for (int t = t0; t < t1; ++t) {
            base += pv[t] / fminf(fmaxf(al[t], 1e-10f), 1.f);
            al[t] = sc[t] * al[t] * base;
        }
    }
    __syncthreads();
    GT_LSTAMP(6);
    if (tid < TvFull) P.align[(size_t)b * P.ldalign + tid] = tid < Tv ? al[tid] : 0.f;

    // ---- context: ctx[a] = sum_t al[t] * pm[t][a]; lane = channel a (conflict-free column reads of the tile),
    //      CPARTS row groups reduced through LDS
    const int ca = tid % A, cp = tid / A;
    float cacc = 0.f;
    for (int c = nchunks - 1; c >= 0; --c) {           // the tile still holds the LAST chunk of the score pass
        if (c != nchunks - 1) {
            float4 v[NP];
            load_rows(v, c);
            __syncthreads();
            store_rows(v);
            __syncthreads();
        }
        const int nr = min(ROWS, Tv - c * ROWS);
        const float* alc = al + c * ROWS;
        float p0 = 0.f, p1 = 0.f;
        int t = cp;
        for (; t + CPARTS < nr; t += 2 * CPARTS) {
            p0 = __builtin_fmaf(alc[t], tile[t * LD + ca], p0);
            p1 = __builtin_fmaf(alc[t + CPARTS], tile[(t + CPARTS) * LD + ca], p1);
        }
        if (t < nr) p0 = __builtin_fmaf(alc[t], tile[t * LD + ca], p0);
        cacc += p0 + p1;
    }
    red[cp * A + ca] = cacc;
    __syncthreads();
    if (tid < A) {
        float v[CPARTS];
#pragma unroll
        for (int w = 0; w < CPARTS; ++w) v[w] = red[w * A + tid];
        float z = 0.f;
#pragma unroll
        for (int w = 0; w < CPARTS; ++w) z += v[w];
        P.xa[gt_blk_off(b, P1 + tid, P.MT)] = z;     // context, k in [P1, P1+A)
        if (P.xah) P.xah[gt_blk_off_h(b, P1 + tid, P.MT)] = gt_bf16_bits(z);
    }
    GT_LSTAMP(7);
}
