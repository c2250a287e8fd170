// Skinny (batch-rows) fp32 GEMM on MFMA with fused epilogues -- the decode-loop workhorse.
//
//   Z[M, 16*ntiles] = concat(seg0, seg1, seg2)[M, K] . W[K, 16*ntiles] + bias      (M = batch rows)
//
// Replaces, per decoder step (reference Modules/Taco2.py:96-120):
//   Prenet Dense+ReLU+Dropout (Taco2.py:262-283)            -> EPI_RELU_DROP
//   attention Query Dense (Steps.py:122)                    -> EPI_LINEAR
//   LSTMCell x2 via StackedRNNCells (Taco2.py:77-85,111)    -> EPI_LSTM (gates fused, Appendix A.6)
//   Projection Dense + split mel/stop (Taco2.py:112-118)    -> EPI_LINEAR with a split column
// and per encoder time step the two directions of the BiLSTM (Taco2.py:39-43) -> EPI_LSTM, grid.z = 2.
//
// Design (gfx950): one workgroup owns 16 output columns (for an LSTM: 4 hidden units x 4 gates, so the
// gate non-linearity and the cell update never leave the CU) and ALL batch rows (<=32 per blockIdx.y);
// K is split over the workgroup's waves and reduced through LDS.  Weights are repacked at finalize into
// MFMA-fragment order [tile][k-block of 16][lane][4] so a wave streams them with one 1-KiB coalesced
// global_load_dwordx4 per k-block and never touches LDS for them (weights are used once per step --
// cdna_hip_programming.md 'GEMV / M<=16 decode weights: load straight to VGPRs').  The 4 values a lane
// loads are the B operands of 4 consecutive v_mfma_f32_16x16x4_f32; the matching A operands (activations)
// are 4 consecutive k of one batch row = one 16-byte load.  fp32 in / fp32 accumulate (exact fp32 FMA
// chain), since the parity bar is 1e-3 through a 1000-frame recurrence.
#include "skinny_body.h"

// TAG only separates kernel symbols per call site (decode LSTM layer 1 / 2, encoder BiLSTM) so that
// rocprofv3 --stats reports them on separate lines.
template <int EPI, int NW, int TAG>
__global__ __launch_bounds__(NW * 64) void gt_skinny_kernel(SkinnyArgs a0, SkinnyArgs a1) {
    const SkinnyArgs& A = (blockIdx.z == 0) ? a0 : a1;
    __shared__ __attribute__((aligned(16))) float lds[SkinnyLds<NW>::kFloats];
    // weights of the big decode LSTM GEMMs are read by exactly one CU once per step: stream them non-temporally
    constexpr bool NT = (TAG == TAG_DEC_LSTM1 || TAG == TAG_DEC_LSTM2);
    gt_skinny_body<EPI, NW, NT>(A, blockIdx.x, blockIdx.y, lds);
}

// Projection (EPI_LINEAR, 11 workgroups at 161 columns) co-scheduled with recurrent-half partial GEMM tiles.
template <int NW>
__global__ __launch_bounds__(NW * 64) void gt_skinny_co_kernel(SkinnyArgs main_args, SkinnyArgs co, int n_main, int co_begin) {
    __shared__ __attribute__((aligned(16))) float lds[SkinnyLds<NW>::kFloats];
    const int mchunks = (main_args.M + 31) / 32;
    if ((int)blockIdx.x < n_main) {
        for (int mc = 0; mc < mchunks; ++mc) {
            gt_skinny_body<EPI_LINEAR, NW, false>(main_args, blockIdx.x, mc, lds);
            __syncthreads();
        }
    } else {
        const int tile = co_begin + (int)blockIdx.x - n_main;
        for (int mc = 0; mc < mchunks; ++mc) {
            gt_skinny_body<EPI_PARTIAL, NW, true>(co, tile, mc, lds);
            __syncthreads();
        }
    }
}

hipError_t gt_launch_skinny_co(const SkinnyArgs& main_args, int ntiles, const SkinnyArgs& co, int co_begin, int co_end,
                               hipStream_t stream) {
    const int nco = co_end > co_begin ? co_end - co_begin : 0;
    hipLaunchKernelGGL((gt_skinny_co_kernel<8>), dim3(ntiles + nco), dim3(512), 0, stream, main_args, co, ntiles, co_begin);
    return hipGetLastError();
}

template <int EPI, int TAG>
static hipError_t launch_epi(const SkinnyArgs& a0, const SkinnyArgs* a1, int ntiles, hipStream_t stream) {
    dim3 grid(ntiles, (a0.M + 31) / 32, a1 ? 2 : 1);
    const SkinnyArgs& b = a1 ? *a1 : a0;
    // K-split width: 8 waves when the K loop is long enough to feed them
    if (a0.nkb >= 32) {
        hipLaunchKernelGGL((gt_skinny_kernel<EPI, 8, TAG>), grid, dim3(512), 0, stream, a0, b);
    } else {
        hipLaunchKernelGGL((gt_skinny_kernel<EPI, 4, TAG>), grid, dim3(256), 0, stream, a0, b);
    }
    return hipGetLastError();
}

hipError_t gt_launch_skinny(int epi, const SkinnyArgs& a0, const SkinnyArgs* a1, int ntiles, hipStream_t stream, int tag) {
    switch (epi) {
        case EPI_LINEAR: return launch_epi<EPI_LINEAR, 0>(a0, a1, ntiles, stream);
        case EPI_RELU_DROP: return launch_epi<EPI_RELU_DROP, 0>(a0, a1, ntiles, stream);
        case EPI_LSTM:
            switch (tag) {
                case TAG_DEC_LSTM1: return launch_epi<EPI_LSTM, TAG_DEC_LSTM1>(a0, a1, ntiles, stream);
                case TAG_DEC_LSTM2: return launch_epi<EPI_LSTM, TAG_DEC_LSTM2>(a0, a1, ntiles, stream);
                default: return launch_epi<EPI_LSTM, TAG_ENC_BILSTM>(a0, a1, ntiles, stream);
            }
    }
    return hipErrorInvalidValue;
}
