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
#include "device_utils.h"
#include "kernels.h"

template <int EPI, int NW>
__global__ __launch_bounds__(NW * 64) void gt_skinny_kernel(SkinnyArgs a0, SkinnyArgs a1) {
    const SkinnyArgs& A = (blockIdx.z == 0) ? a0 : a1;
    __shared__ float part[NW][32][17];
    __shared__ float zs[32][17];

    const int tile = blockIdx.x;
    const int m0 = blockIdx.y * 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int M = A.M;
    const int row0 = min(m0 + r, M - 1);
    const int row1 = min(m0 + 16 + r, M - 1);

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};

    const float4* wp = reinterpret_cast<const float4*>(A.wp) + (size_t)tile * A.nkb * 64 + lane;
    int kb_base = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int snkb = A.seg[s].nkb;
        if (snkb > 0) {
            const float* p0 = A.seg[s].ptr + (size_t)row0 * A.seg[s].ld + 4 * q;
            const float* p1 = A.seg[s].ptr + (size_t)row1 * A.seg[s].ld + 4 * q;
            int start = (wave - kb_base) % NW;
            if (start < 0) start += NW;
            const float4* wps = wp + (size_t)kb_base * 64;
#pragma unroll 4
            for (int kb = start; kb < snkb; kb += NW) {
                const float4 b = wps[(size_t)kb * 64];
                const float4 x0 = *reinterpret_cast<const float4*>(p0 + kb * 16);
                const float4 x1 = *reinterpret_cast<const float4*>(p1 + kb * 16);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.x, b.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.x, b.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.y, b.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.y, b.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.z, b.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.z, b.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0.w, b.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1.w, b.w, acc1, 0, 0, 0);
            }
        }
        kb_base += snkb;
    }

    // C/D layout of 16x16x4: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        part[wave][q * 4 + j][r] = acc0[j];
        part[wave][16 + q * 4 + j][r] = acc1[j];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 512; e += NW * 64) {
        const int row = e >> 4, col = e & 15;
        float z = A.bias[tile * 16 + col];
#pragma unroll
        for (int w = 0; w < NW; ++w) z += part[w][row][col];
        zs[row][col] = z;
    }
    __syncthreads();

    if (EPI == EPI_LSTM) {
        // tile-local column g*4+u  <->  gate g (i,f,c~,o) of hidden unit tile*4+u
        const int e = threadIdx.x;
        if (e < 128) {
            const int row = e >> 2, u = e & 3;
            const int grow = m0 + row;
            const int unit = tile * 4 + u;
            if (grow < M && unit < A.N) {
                const float gi = gt_sigmoid(zs[row][u]);
                const float gf = gt_sigmoid(zs[row][4 + u]);
                const float gg = gt_tanh(zs[row][8 + u]);
                const float go = gt_sigmoid(zs[row][12 + u]);
                float* cp = A.c + (size_t)grow * A.N + unit;
                const float c2 = gf * (*cp) + gi * gg;
                *cp = c2;
                A.h[(size_t)grow * A.ldh + unit] = go * gt_tanh(c2);
            }
        }
    } else {
        for (int e = threadIdx.x; e < 512; e += NW * 64) {
            const int row = e >> 4, col = e & 15;
            const int grow = m0 + row, gcol = tile * 16 + col;
            if (grow < M && gcol < A.N) {
                float v = zs[row][col];
                if (EPI == EPI_RELU_DROP) {
                    v = fmaxf(v, 0.f);
                    if (A.drop_rate > 0.f) {
                        float keep;
                        if (A.mask) {
                            keep = A.mask[(size_t)grow * A.ldm + gcol];
                        } else {
                            Philox4 p = gt_philox(*A.seed_ptr, (uint32_t)(grow * A.N + gcol), A.rng_step, 0u, A.rng_stream);
                            keep = (gt_u01(p.x) > A.drop_rate) ? 1.f : 0.f;
                        }
                        v = v * A.drop_scale * keep;      // tf.nn.dropout: x * scale * mask
                    }
                }
                if (gcol < A.n_split) A.out[(size_t)grow * A.ldo + gcol] = v;
                else A.out2[(size_t)grow * A.ldo2 + (gcol - A.n_split)] = v;
            }
        }
    }
}

template <int EPI>
static hipError_t launch_epi(const SkinnyArgs& a0, const SkinnyArgs* a1, int ntiles, hipStream_t stream) {
    dim3 grid(ntiles, (a0.M + 31) / 32, a1 ? 2 : 1);
    const SkinnyArgs& b = a1 ? *a1 : a0;
    // K-split width: 8 waves when the K loop is long enough to feed them
    if (a0.nkb >= 32) {
        hipLaunchKernelGGL((gt_skinny_kernel<EPI, 8>), grid, dim3(512), 0, stream, a0, b);
    } else {
        hipLaunchKernelGGL((gt_skinny_kernel<EPI, 4>), grid, dim3(256), 0, stream, a0, b);
    }
    return hipGetLastError();
}

hipError_t gt_launch_skinny(int epi, const SkinnyArgs& a0, const SkinnyArgs* a1, int ntiles, hipStream_t stream) {
    switch (epi) {
        case EPI_LINEAR: return launch_epi<EPI_LINEAR>(a0, a1, ntiles, stream);
        case EPI_RELU_DROP: return launch_epi<EPI_RELU_DROP>(a0, a1, ntiles, stream);
        case EPI_LSTM: return launch_epi<EPI_LSTM>(a0, a1, ntiles, stream);
    }
    return hipErrorInvalidValue;
}
