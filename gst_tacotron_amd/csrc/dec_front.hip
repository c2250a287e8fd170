// Fused decoder-step front end, one 1024-thread workgroup per utterance:
//   prenet Dense+ReLU+Dropout x2 (reference Modules/Taco2.py:262-283, dropout live at inference :283)
//   attention Query Dense                        (Modules/Attention/Steps.py:122)
//   score / SMA|BMA alignment / context          (Steps.py:126-166, 168-199, 215-229)
// on the hoisted processed memory (Steps.py:123 is loop-invariant).
//
// Why one kernel: on MI355X a dependent kernel boundary costs ~2 us and a tiny kernel ~3-5 us of pure
// latency; the four separate launches of v1 (prenet0, prenet1, query, attention) cost ~31 us per decoder
// step.  Everything here is per-utterance, so one workgroup can run the whole chain with only
// workgroup barriers in between.  The price is that each workgroup pulls the prenet / query weights and its
// processed-memory rows (448 KB fp32) itself, at a CU's ~130 GB/s L1 fill rate -- a third of the chain.  So:
//   * requests go out in NEED order (loads return in issue order): small operands, all of prenet 1's weights, and --
//     once prenet 1 has consumed them and freed their registers -- the query weights and the memory rows;
//   * prenet 0's pre-activations arrive from the previous step's projection launch (Z0; both layers are linear);
//   * in throughput mode the dropout keep decisions are a counter hash of the seed (device_utils.h gt_keep_word), known a
//     few scalar instructions after the kernel starts: weight rows that will meet an exact zero are never requested;
//   * the prologue is branch-free and, where the GEMV plans divide evenly (EXACT), its loads are unpredicated
//     SGPR-base + one-VGPR-offset loads (a predicated load costs ~30 instructions of exec-mask bookkeeping).
// The other ~224 CUs of the launch run worker workgroups: recurrent halves of the LSTM gate GEMMs (see DecFrontArgs).
//
// GEMV mapping: lane = 4 consecutive output columns (one coalesced 16-byte load per k row), the K range
// is split over the remaining lanes and reduced through LDS.  Attention mapping: the utterance's processed memory is
// staged once into a padded LDS tile, read row-wise for the scores (L lanes per row, shuffle reduction) and
// column-wise for the context.
#include "front_body.h"

static size_t front_lds_bytes(const DecFrontArgs& a) {
    int mx = a.mel > a.P0 ? a.mel : a.P0; if (a.P1 > mx) mx = a.P1;
    const size_t tv4 = (a.Tv + 3) & ~3;
    const int L = a.A == 16 ? 4 : 8;
    size_t fl = ((mx + 3) & ~3) + a.P0 + a.P1 + 2 * (size_t)a.A + 3 * tv4 + FT + 4 * (size_t)FT;
    fl += 2 * (size_t)a.P0 + 2 * (size_t)a.P1 + a.A + tv4;      // staged biases, keep-scales, noise
    fl += (size_t)(FT / L) * (a.A + 4);             // processed-memory tile
    if (a.loc_f > 0) {                              // LSA: location features of the tile's rows + the weight image (kernels.h LsaPack)
        const LsaPack lp = gt_lsa_pack(a.A, a.loc_f, a.loc_k);
        fl += (size_t)(FT / L) * lp.LFS + lp.total;
    }
    size_t worker = SkinnyLds<FT / 64>::kFloats;
    if ((size_t)SkinnyMultiLds<FT / 64, WT>::kFloats > worker) worker = SkinnyMultiLds<FT / 64, WT>::kFloats;
    if (fl < worker) fl = worker;
    return fl * sizeof(float);
}

bool gt_dec_front_supported(int mel, int P0, int P1, int A, int Tv, int loc_f, int loc_k) {
    if (!(A == 16 || A == 32 || A == 64 || A == 128 || A == 256) || mel > FT) return false;
    auto ok = [](int K, int N, int maxr) {
        if (N % 4 || N / 4 > FT) return false;
        int kparts = FT / (N / 4);
        if (kparts > K) kparts = K;
        return (K + kparts - 1) / kparts <= maxr;
    };
    // register blocks: prenet0 and query one 8-row block per lane, prenet1 two
    if (!ok(mel, P0, 8) || !ok(P0, P1, FMAXR) || !ok(P1, A, 8)) return false;
    DecFrontArgs t{};
    t.mel = mel; t.P0 = P0; t.P1 = P1; t.A = A; t.Tv = Tv; t.loc_f = loc_f; t.loc_k = loc_k;
    return front_lds_bytes(t) <= 150 * 1024;
}

// prenet-1 and query plans divide evenly: 16 resp. 8 rows for every lane (gemv_load EXACT)
static bool front_exact(const DecFrontArgs& a) {
    auto rows = [](int K, int N) {
        if (N % 4 || N / 4 > FT || FT % (N / 4)) return -1;
        const int kparts = FT / (N / 4);
        return (kparts <= K && K % kparts == 0) ? K / kparts : -1;
    };
    return rows(a.P0, a.P1) == 16 && rows(a.P1, a.A) == 8;
}

// the lean utterance path's preconditions (front_lean.h)
static bool front_lean_ok(const DecFrontArgs& a) {
    if (!a.lean_front || !a.z0 || !front_exact(a) || a.type == GSTTACO_ATT_LSA) return false;
    if (a.P0 > FT || a.P1 > FT || a.A > FT || a.Tv > FT) return false;
    if (a.drop_rate > 0.f && !(a.mask0 && a.mask1) && !a.keep_hash) return false;
    if (a.sigmoid_noise > 0.f && !a.noise) return false;
    return true;
}

template <int L, int NP, bool Z0, int LEAN>
static void front_launch2(const DecFrontArgs& a, hipStream_t s) {
    const dim3 grid(a.B + a.n_workers), block(FT);
    const size_t lds = front_lds_bytes(a);
    if (a.type == GSTTACO_ATT_LSA) {
        gt_front_lsa_launch(NP == 1 ? (L == 4 ? 0 : 1) : (NP == 2 ? 2 : (NP == 4 ? 3 : 4)), Z0, LEAN, Z0 && front_exact(a), grid, lds, s, a);
        return;
    }
    // (without Z0 -- step 0, or prenet-0 fusion off -- the old request order keeps more rows in flight and the
    // unpredicated variant spills: predicated loads there)
    if (Z0 && front_lean_ok(a)) hipLaunchKernelGGL((gt_dec_front_lean_kernel<L, NP, LEAN>), grid, block, lds, s, a);
    else if (Z0 && front_exact(a)) hipLaunchKernelGGL((gt_dec_front_kernel<L, NP, Z0, LEAN, true>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((gt_dec_front_kernel<L, NP, Z0, LEAN, false>), grid, block, lds, s, a);
}

template <int L, int NP>
static hipError_t front_launch(const DecFrontArgs& a, hipStream_t s) {
    if (a.z0) {
        if (a.lean_rec == 2) front_launch2<L, NP, true, 2>(a, s);
        else if (a.lean_rec == 1) front_launch2<L, NP, true, 1>(a, s);
        else front_launch2<L, NP, true, 0>(a, s);
    } else {
        if (a.lean_rec == 2) front_launch2<L, NP, false, 2>(a, s);
        else if (a.lean_rec == 1) front_launch2<L, NP, false, 1>(a, s);
        else front_launch2<L, NP, false, 0>(a, s);
    }
    return hipGetLastError();
}

hipError_t gt_launch_dec_front(const DecFrontArgs& a, hipStream_t s) {
    switch (a.A) {
        case 16: return front_launch<4, 1>(a, s);
        case 32: return front_launch<8, 1>(a, s);
        case 64: return front_launch<8, 2>(a, s);
        case 128: return front_launch<8, 4>(a, s);
        case 256: return front_launch<8, 8>(a, s);
    }
    return hipErrorInvalidValue;
}

hipError_t gt_dec_front_init() {
    hipError_t e;
#define FRONT_ATTR1(L, NP, Z, LN, EX)                                                                 \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gt_dec_front_kernel<L, NP, Z, LN, EX>),   \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);            \
    if (e != hipSuccess) return e;
#define FRONT_ATTR2(L, NP, EX) FRONT_ATTR1(L, NP, false, 0, EX) FRONT_ATTR1(L, NP, false, 1, EX) FRONT_ATTR1(L, NP, false, 2, EX) FRONT_ATTR1(L, NP, true, 0, EX) FRONT_ATTR1(L, NP, true, 1, EX) FRONT_ATTR1(L, NP, true, 2, EX)
#define FRONT_ATTRL(L, NP, LN)                                                                       \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gt_dec_front_lean_kernel<L, NP, LN>),      \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);            \
    if (e != hipSuccess) return e;
#define FRONT_ATTR(L, NP) FRONT_ATTR2(L, NP, false) FRONT_ATTR2(L, NP, true) FRONT_ATTRL(L, NP, 0) FRONT_ATTRL(L, NP, 1) FRONT_ATTRL(L, NP, 2)
    FRONT_ATTR(4, 1) FRONT_ATTR(8, 1) FRONT_ATTR(8, 2) FRONT_ATTR(8, 4) FRONT_ATTR(8, 8)
    if ((e = gt_front_lsa_init()) != hipSuccess) return e;
#undef FRONT_ATTR
#undef FRONT_ATTRL
#undef FRONT_ATTR1
#undef FRONT_ATTR2
    return hipSuccess;
}
