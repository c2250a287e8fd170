// The step-wise location-sensitive attention extension (reference Modules/Attention/Layers.py:289-444 restated per
// decoder step) on the fused front end: the LSA = true instantiations of the general utterance kernel (front_body.h
// gt_dec_front_kernel, where the design note is).  A translation unit of its own so that the two sets of instantiations compile
// side by side.  Which path an LSA model takes: fp32, <= 32 utterances and <= 128 tokens (gt_persist_decode_lsa_fits) run the
// persistent decode launch, whose one-group chain carries the same two MFMA products (persist_decode.hip pd_chain_rest<.., LSA>,
// bitwise this kernel); everything else -- more rows or tokens, mixed precision, several live contexts -- runs the launch path with
// this kernel as the first launch of every step (the lean utterance path does not know LSA).
#include "front_body.h"

// without the prenet-0 pre-activations (step 0, or prenet-0 fusion off) the predicated variant, as for BMA / SMA (dec_front.hip front_launch2)
template <int L, int NP, int LEAN>
static void lsa_launch3(bool z0, bool exact, dim3 grid, size_t lds, hipStream_t s, const DecFrontArgs& a) {
    if (z0 && exact) hipLaunchKernelGGL((gt_dec_front_kernel<L, NP, true, LEAN, true, true>), grid, dim3(FT), lds, s, a);
    else if (z0) hipLaunchKernelGGL((gt_dec_front_kernel<L, NP, true, LEAN, false, true>), grid, dim3(FT), lds, s, a);
    else hipLaunchKernelGGL((gt_dec_front_kernel<L, NP, false, LEAN, false, true>), grid, dim3(FT), lds, s, a);
}

template <int L, int NP>
static void lsa_launch2(bool z0, int lean, bool exact, dim3 grid, size_t lds, hipStream_t s, const DecFrontArgs& a) {
    if (lean == 2) lsa_launch3<L, NP, 2>(z0, exact, grid, lds, s, a);
    else if (lean == 1) lsa_launch3<L, NP, 1>(z0, exact, grid, lds, s, a);
    else lsa_launch3<L, NP, 0>(z0, exact, grid, lds, s, a);
}

void gt_front_lsa_launch(int shape, bool z0, int lean, bool exact, dim3 grid, size_t lds, hipStream_t s, const DecFrontArgs& a) {
    switch (shape) {
        case 0: lsa_launch2<4, 1>(z0, lean, exact, grid, lds, s, a); break;
        case 1: lsa_launch2<8, 1>(z0, lean, exact, grid, lds, s, a); break;
        case 2: lsa_launch2<8, 2>(z0, lean, exact, grid, lds, s, a); break;
        case 3: lsa_launch2<8, 4>(z0, lean, exact, grid, lds, s, a); break;
        default: lsa_launch2<8, 8>(z0, lean, exact, grid, lds, s, a); break;
    }
}

hipError_t gt_front_lsa_init() {
    hipError_t e;
#define LSA_ATTR1(L, NP, Z, LN, EX)                                                                       \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(gt_dec_front_kernel<L, NP, Z, LN, EX, true>),  \
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                      \
    if (e != hipSuccess) return e;
#define LSA_ATTR2(L, NP, LN) LSA_ATTR1(L, NP, true, LN, true) LSA_ATTR1(L, NP, true, LN, false) LSA_ATTR1(L, NP, false, LN, false)
#define LSA_ATTR(L, NP) LSA_ATTR2(L, NP, 0) LSA_ATTR2(L, NP, 1) LSA_ATTR2(L, NP, 2)
    LSA_ATTR(4, 1) LSA_ATTR(8, 1) LSA_ATTR(8, 2) LSA_ATTR(8, 4) LSA_ATTR(8, 8)
#undef LSA_ATTR
#undef LSA_ATTR2
#undef LSA_ATTR1
    return hipSuccess;
}
