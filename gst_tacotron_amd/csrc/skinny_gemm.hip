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

// TAG only separates kernel symbols per call site (decode LSTM layer 1 / 2, encoder BiLSTM) so that
// rocprofv3 --stats reports them on separate lines.
template <int EPI, int NW, int TAG>
__global__ __launch_bounds__(NW * 64) void gt_skinny_kernel(SkinnyArgs a0, SkinnyArgs a1) {
    const SkinnyArgs& A = (blockIdx.z == 0) ? a0 : a1;
    // k-blocks a wave keeps in flight at once: 3 x 16-byte loads each -> 12 VGPRs per k-block
    constexpr int MAXI = NW == 8 ? 16 : 8;
    // weights of the big LSTM GEMMs are read by exactly one CU once per step: stream them non-temporally so
    // they do not evict the activations / prenet weights / processed memory that every step re-reads from L2
    constexpr bool NT_WEIGHTS = (TAG == TAG_DEC_LSTM1 || TAG == TAG_DEC_LSTM2);
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

    // epilogue operands are requested first so they are never on the dependent tail
    const int e_row = threadIdx.x >> 4, e_col = threadIdx.x & 15;       // (only the first 512 threads' worth is used)
    float bias_v[512 / (NW * 64) > 0 ? 512 / (NW * 64) : 1];
#pragma unroll
    for (int i = 0; i < 512 / (NW * 64); ++i) bias_v[i] = A.bias[tile * 16 + e_col];
    float c_prev = 0.f;
    if (EPI == EPI_LSTM && threadIdx.x < 128) {
        const int grow = m0 + (threadIdx.x >> 2), unit = tile * 4 + (threadIdx.x & 3);
        if (grow < M && unit < A.N) c_prev = A.c[(size_t)grow * A.N + unit];
    }

    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc1 = {0.f, 0.f, 0.f, 0.f};
    GT_STAMP(A.dbg, 0);

    const float4* wp = reinterpret_cast<const float4*>(A.wp) + (size_t)tile * A.nkb * 64 + lane;
    const int nkb = A.nkb;
    const int e0 = A.seg[0].nkb, e1 = e0 + A.seg[1].nkb;
    // per-segment lane base pointers for the two M-tiles (rows m0..m0+15, m0+16..m0+31) and the k-block stride
    const int MT = A.MT;
    const int mt0 = blockIdx.y * 2, mt1 = min(mt0 + 1, MT - 1);
    const float *sp0[3], *sp1[3];
    int sstep[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const SkinnySeg& S = A.seg[s].nkb ? A.seg[s] : A.seg[0];
        if (S.blocked) {
            sp0[s] = S.ptr + (size_t)mt0 * 256 + lane * 4;
            sp1[s] = S.ptr + (size_t)mt1 * 256 + lane * 4;
            sstep[s] = MT * 256;
        } else {
            sp0[s] = S.ptr + (size_t)row0 * S.ld + 4 * q;
            sp1[s] = S.ptr + (size_t)row1 * S.ld + 4 * q;
            sstep[s] = 16;
        }
    }

    for (int base = wave; base < nkb; base += NW * MAXI) {
        float4 b[MAXI], x0[MAXI], x1[MAXI];
        // issue every load of this chunk before the first MFMA: the wave's whole K range is in flight at once
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int kb = base + i * NW;               // wave-uniform
            if (kb < nkb) {
                const float *p0, *p1;
                int lk, st;
                if (kb < e0) { p0 = sp0[0]; p1 = sp1[0]; lk = kb; st = sstep[0]; }
                else if (kb < e1) { p0 = sp0[1]; p1 = sp1[1]; lk = kb - e0; st = sstep[1]; }
                else { p0 = sp0[2]; p1 = sp1[2]; lk = kb - e1; st = sstep[2]; }
                if (NT_WEIGHTS) {
                    const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + (size_t)kb * 64));
                    b[i] = make_float4(t[0], t[1], t[2], t[3]);
                } else {
                    b[i] = wp[(size_t)kb * 64];
                }
                x0[i] = *reinterpret_cast<const float4*>(p0 + (size_t)lk * st);
                x1[i] = *reinterpret_cast<const float4*>(p1 + (size_t)lk * st);
            }
        }
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int kb = base + i * NW;
            if (kb < nkb) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].x, b[i].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].x, b[i].x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].y, b[i].y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].y, b[i].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].z, b[i].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].z, b[i].z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].w, b[i].w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].w, b[i].w, acc1, 0, 0, 0);
            }
        }
    }

    GT_STAMP(A.dbg, 1);
    // C/D layout of 16x16x4: col = lane&15, row = (lane>>4)*4 + reg
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        part[wave][q * 4 + j][r] = acc0[j];
        part[wave][16 + q * 4 + j][r] = acc1[j];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 512 / (NW * 64); ++i) {
        const int row = e_row + i * (NW * 4);
        float z = bias_v[i];
#pragma unroll
        for (int w = 0; w < NW; ++w) z += part[w][row][e_col];
        zs[row][e_col] = z;
    }
    __syncthreads();
    GT_STAMP(A.dbg, 2);

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
                const float c2 = gf * c_prev + gi * gg;
                A.c[(size_t)grow * A.N + unit] = c2;
                const float hv = go * gt_tanh(c2);
                if (A.out_blocked) A.h[gt_blk_off(grow, unit, MT)] = hv;
                else A.h[(size_t)grow * A.ldh + unit] = hv;
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
                if (gcol < A.n_split) {
                    if (A.out_blocked) A.out[gt_blk_off(grow, gcol, MT)] = v;
                    else A.out[(size_t)grow * A.ldo + gcol] = v;
                }
                else A.out2[(size_t)grow * A.ldo2 + (gcol - A.n_split)] = v;
            }
        }
    }
    GT_STAMP(A.dbg, 3);
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
