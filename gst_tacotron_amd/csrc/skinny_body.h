// Body of the skinny (batch-rows) fp32 MFMA GEMM, shared by gt_skinny_kernel (skinny_gemm.hip) and by the worker
// workgroups of the fused decoder front kernel (dec_front.hip).  See skinny_gemm.hip for the design notes.
#pragma once
#include "device_utils.h"
#include "kernels.h"

// LDS needed by one workgroup running the body with NW waves (floats)
template <int NW>
struct SkinnyLds {
    static constexpr int kPart = NW * 32 * 17;
    static constexpr int kZs = 32 * 17;
    static constexpr int kFloats = kPart + kZs;
};

// One workgroup (NW waves, all NW*64 threads must call) computes output tile `tile` (16 columns) for batch rows
// [mchunk*32, mchunk*32+32).  `lds` points at SkinnyLds<NW>::kFloats floats of LDS.
template <int EPI, int NW, bool NT_WEIGHTS>
__device__ __forceinline__ void gt_skinny_body(const SkinnyArgs& A, const int tile, const int mchunk, float* lds) {
    // k-blocks a wave keeps in flight at once: 3 x 16-byte loads each -> 12 VGPRs per k-block
    constexpr int MAXI = NW == 8 ? 16 : (NW == 16 ? 4 : 8);
    float (*part)[32][17] = reinterpret_cast<float (*)[32][17]>(lds);
    float (*zs)[17] = reinterpret_cast<float (*)[17]>(lds + SkinnyLds<NW>::kPart);

    GT_STAMP(A.dbg, 4);
    const int m0 = mchunk * 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int M = A.M;
    const int row0 = min(m0 + r, M - 1);
    const int row1 = min(m0 + 16 + r, M - 1);

    // epilogue operands are requested first so they are never on the dependent tail
    const int e_row = threadIdx.x >> 4, e_col = threadIdx.x & 15;       // (only the first 512 threads' worth is used)
    constexpr int NE = (512 / (NW * 64)) > 0 ? (512 / (NW * 64)) : 1;     // outputs per thread of the 32x16 tile
    const bool e_act = NW * 64 <= 512 || threadIdx.x < 512;
    float bias_v[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        bias_v[i] = A.bias[tile * 16 + e_col];
        if (EPI == EPI_LSTM && A.partial_in && e_act && m0 + e_row + i * (NW * 4) < A.MT * 16)
            bias_v[i] += A.partial_in[((size_t)tile * A.MT * 16 + m0 + e_row + i * (NW * 4)) * 16 + e_col];   // recurrent half + bias
    }
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
    const int mt0 = mchunk * 2, mt1 = min(mt0 + 1, MT - 1);
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

    if (A.bf16) {
        // ---- mixed precision: K in blocks of 32 = two consecutive 16-blocks (2j, 2j+1).  MFMA k-slot (q = lane>>4, i = 0..7)
        // <-> k = 32 j + 16 (i>>2) + 4 q + (i&3): a lane's A fragment is exactly the two float4 it loads today from the two
        // 16-blocks, rounded to bf16 (v_cvt_pk_bf16_f32, RNE); the weights are packed to the same slot order at finalize,
        // one 16-byte load per lane per 32 k.
        constexpr int MAXB = MAXI / 2;
        const int nkb32 = (nkb + 1) >> 1;
        const uint4* wq = reinterpret_cast<const uint4*>(A.wp) + (size_t)tile * nkb32 * 64 + lane;
        for (int base = wave; base < nkb32; base += NW * MAXB) {
            uint4 b[MAXB];
            float4 x0[MAXB][2], x1[MAXB][2];
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int kb32 = base + i * NW;             // wave-uniform
                if (kb32 < nkb32) {
                    if (NT_WEIGHTS && !A.keep_weights) {
                        const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(wq + (size_t)kb32 * 64));
                        b[i] = make_uint4(t[0], t[1], t[2], t[3]);
                    } else {
                        b[i] = wq[(size_t)kb32 * 64];
                    }
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int kb = 2 * kb32 + hf;
                        x0[i][hf] = make_float4(0.f, 0.f, 0.f, 0.f);
                        x1[i][hf] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (kb < nkb) {
                            const float *p0, *p1;
                            int lk, st;
                            if (kb < e0) { p0 = sp0[0]; p1 = sp1[0]; lk = kb; st = sstep[0]; }
                            else if (kb < e1) { p0 = sp0[1]; p1 = sp1[1]; lk = kb - e0; st = sstep[1]; }
                            else { p0 = sp0[2]; p1 = sp1[2]; lk = kb - e1; st = sstep[2]; }
                            x0[i][hf] = *reinterpret_cast<const float4*>(p0 + (size_t)lk * st);
                            x1[i][hf] = *reinterpret_cast<const float4*>(p1 + (size_t)lk * st);
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int kb32 = base + i * NW;
                if (kb32 < nkb32) {
                    bf16x8 a0, a1, bw;
                    a0[0] = (__bf16)x0[i][0].x; a0[1] = (__bf16)x0[i][0].y; a0[2] = (__bf16)x0[i][0].z; a0[3] = (__bf16)x0[i][0].w;
                    a0[4] = (__bf16)x0[i][1].x; a0[5] = (__bf16)x0[i][1].y; a0[6] = (__bf16)x0[i][1].z; a0[7] = (__bf16)x0[i][1].w;
                    a1[0] = (__bf16)x1[i][0].x; a1[1] = (__bf16)x1[i][0].y; a1[2] = (__bf16)x1[i][0].z; a1[3] = (__bf16)x1[i][0].w;
                    a1[4] = (__bf16)x1[i][1].x; a1[5] = (__bf16)x1[i][1].y; a1[6] = (__bf16)x1[i][1].z; a1[7] = (__bf16)x1[i][1].w;
                    __builtin_memcpy(&bw, &b[i], 16);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bw, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bw, acc1, 0, 0, 0);
                }
            }
        }
    } else {
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
                    if (NT_WEIGHTS && !A.keep_weights) {
                        const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(wp + (size_t)kb * 64));
                        b[i] = make_float4(t[0], t[1], t[2], t[3]);
                    } else {
                        b[i] = wp[(size_t)kb * 64];
                    }
                    x0[i] = *reinterpret_cast<const float4*>(p0 + (size_t)lk * st);
                    x1[i] = *reinterpret_cast<const float4*>(p1 + (size_t)lk * st);
                }
            }
            GT_STAMP(A.dbg, 5);
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
    for (int i = 0; i < NE; ++i) {
        if (!e_act) break;
        const int row = e_row + i * (NW * 4);
        float z = bias_v[i];
#pragma unroll
        for (int w = 0; w < NW; ++w) z += part[w][row][e_col];
        zs[row][e_col] = z;
    }
    __syncthreads();
    GT_STAMP(A.dbg, 2);

    if (EPI == EPI_PARTIAL) {
        // pre-activation partial sums in tile order [tile][MT*16 rows][16 cols]
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int row = e_row + i * (NW * 4);
            if (e_act && m0 + row < A.MT * 16) A.partial_out[((size_t)tile * A.MT * 16 + m0 + row) * 16 + e_col] = zs[row][e_col];
        }
    } else if (EPI == EPI_LSTM) {
        // tile-local column g*4+u  <->  gate g (i,f,c~,o) of hidden unit tile*4+u
        const int e = threadIdx.x;
        if (e < 128) {
            const int row = e >> 2, u = e & 3;
            const int grow = m0 + row;
            const int unit = tile * 4 + u;
            if (grow < M && unit < A.N && A.row_len && A.t_index >= A.row_len[grow]) {
                // masked mode: this time step does not exist for this utterance
                if (A.out_blocked) A.h[gt_blk_off(grow, unit, MT)] = 0.f;
                else A.h[(size_t)grow * A.ldh + unit] = 0.f;
            } else if (grow < M && unit < A.N) {
                const float gi = gt_sigmoid(zs[row][u]);
                const float gf = gt_sigmoid(zs[row][4 + u]);
                const float gg = gt_tanh(zs[row][8 + u]);
                const float go = gt_sigmoid(zs[row][12 + u]);
                const float c2 = __builtin_fmaf(gf, c_prev, gi * gg);     // (explicit: the same contraction in every body)
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
                            keep = gt_drop_keep(*A.seed_ptr, A.rng_step, A.rng_stream - GT_RNG_PRENET0, (uint32_t)grow, (uint32_t)gcol,
                                                (uint32_t)A.N, A.drop_rate);
                        }
                        v = v * A.drop_scale * keep;      // tf.nn.dropout: x * scale * mask
                    }
                }
                if (A.out3 && gcol >= A.col3) {
                    A.out3[(size_t)grow * A.ldo3 + (gcol - A.col3)] = v;       // third column region (fused prenet-0)
                } else if (gcol < A.n_split) {
                    if (A.out_blocked) A.out[gt_blk_off(grow, gcol, MT)] = v;
                    else A.out[(size_t)grow * A.ldo + gcol] = v;
                } else if (!A.out3 || gcol < A.n_valid2) {
                    A.out2[(size_t)grow * A.ldo2 + (gcol - A.n_split)] = v;
                }
            }
        }
    }
    GT_STAMP(A.dbg, 3);
}

// ---------------------------------------------------------------------------------------------------------------------
// Worker variant: EPI_PARTIAL only, NT adjacent tiles (16 columns each) per workgroup sharing ONE pass over the
// activations.  A worker with a single tile pulls 64 KB of weights and 128 KB of (L2-served) batch activations per
// K = 1024; two tiles per pass pull 128 + 128 KB, so twice the tiles cost 1.33x the bytes of one and the launch needs one
// round of workers instead of two.  The operand is seg[0], blocked (the decode loop's h1 / h2).  The per-wave k-block
// assignment and the summation order over waves equal gt_skinny_body's, so the partial sums are bitwise the same.
template <int NW, int NT>
struct SkinnyMultiLds {
    static constexpr int kFloats = NT * NW * 32 * 17;
};

template <int NW, int NT, bool NT_WEIGHTS>
__device__ __forceinline__ void gt_skinny_partial_multi(const SkinnyArgs& A, const int tile0, const int ntile, const int mchunk,
                                                        float* lds) {
    constexpr int MAXI = NW == 16 ? 4 : 8;
    float (*part)[NW][32][17] = reinterpret_cast<float (*)[NW][32][17]>(lds);
    const int m0 = mchunk * 32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int MT = A.MT;
    const int mt0 = mchunk * 2, mt1 = min(mt0 + 1, MT - 1);
    const int nkb = A.nkb;
    const float* sp0 = A.seg[0].ptr + (size_t)mt0 * 256 + lane * 4;
    const float* sp1 = A.seg[0].ptr + (size_t)mt1 * 256 + lane * 4;
    const size_t sstep = (size_t)MT * 256;

    // epilogue mapping: thread e -> (tile j, row, col); bias requested before the K loop
    constexpr int NE = (NT * 512 + NW * 64 - 1) / (NW * 64);
    float bias_v[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = threadIdx.x + i * NW * 64;
        const int j = e >> 9;
        bias_v[i] = (e < NT * 512 && j < ntile) ? A.bias[(tile0 + j) * 16 + (e & 15)] : 0.f;
    }

    f32x4 acc0[NT], acc1[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) { acc0[j] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1[j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    if (A.bf16) {
        constexpr int MAXB = MAXI / 2;
        const int nkb32 = (nkb + 1) >> 1;
        const uint4* wq = reinterpret_cast<const uint4*>(A.wp) + (size_t)tile0 * nkb32 * 64 + lane;
        for (int base = wave; base < nkb32; base += NW * MAXB) {
            uint4 b[MAXB][NT];
            float4 x0[MAXB][2], x1[MAXB][2];
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int kb32 = base + i * NW;             // wave-uniform
                if (kb32 < nkb32) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        b[i][j] = make_uint4(0u, 0u, 0u, 0u);
                        if (j < ntile) {
                            const uint4* src = wq + ((size_t)j * nkb32 + kb32) * 64;
                            if (NT_WEIGHTS && !A.keep_weights) {
                                const u32x4 t = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src));
                                b[i][j] = make_uint4(t[0], t[1], t[2], t[3]);
                            } else {
                                b[i][j] = *src;
                            }
                        }
                    }
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int kb = 2 * kb32 + hf;
                        x0[i][hf] = make_float4(0.f, 0.f, 0.f, 0.f);
                        x1[i][hf] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (kb < nkb) {
                            x0[i][hf] = *reinterpret_cast<const float4*>(sp0 + (size_t)kb * sstep);
                            x1[i][hf] = *reinterpret_cast<const float4*>(sp1 + (size_t)kb * sstep);
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < MAXB; ++i) {
                const int kb32 = base + i * NW;
                if (kb32 < nkb32) {
                    bf16x8 a0, a1;
                    a0[0] = (__bf16)x0[i][0].x; a0[1] = (__bf16)x0[i][0].y; a0[2] = (__bf16)x0[i][0].z; a0[3] = (__bf16)x0[i][0].w;
                    a0[4] = (__bf16)x0[i][1].x; a0[5] = (__bf16)x0[i][1].y; a0[6] = (__bf16)x0[i][1].z; a0[7] = (__bf16)x0[i][1].w;
                    a1[0] = (__bf16)x1[i][0].x; a1[1] = (__bf16)x1[i][0].y; a1[2] = (__bf16)x1[i][0].z; a1[3] = (__bf16)x1[i][0].w;
                    a1[4] = (__bf16)x1[i][1].x; a1[5] = (__bf16)x1[i][1].y; a1[6] = (__bf16)x1[i][1].z; a1[7] = (__bf16)x1[i][1].w;
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        bf16x8 bw;
                        __builtin_memcpy(&bw, &b[i][j], 16);
                        acc0[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, bw, acc0[j], 0, 0, 0);
                        acc1[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, bw, acc1[j], 0, 0, 0);
                    }
                }
            }
        }
    } else {
        const float4* wp = reinterpret_cast<const float4*>(A.wp) + (size_t)tile0 * nkb * 64 + lane;
        for (int base = wave; base < nkb; base += NW * MAXI) {
            float4 b[MAXI][NT], x0[MAXI], x1[MAXI];
#pragma unroll
            for (int i = 0; i < MAXI; ++i) {
                const int kb = base + i * NW;               // wave-uniform
                if (kb < nkb) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        b[i][j] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (j < ntile) {
                            const float4* src = wp + ((size_t)j * nkb + kb) * 64;
                            if (NT_WEIGHTS && !A.keep_weights) {
                                const f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src));
                                b[i][j] = make_float4(t[0], t[1], t[2], t[3]);
                            } else {
                                b[i][j] = *src;
                            }
                        }
                    }
                    x0[i] = *reinterpret_cast<const float4*>(sp0 + (size_t)kb * sstep);
                    x1[i] = *reinterpret_cast<const float4*>(sp1 + (size_t)kb * sstep);
                }
            }
#pragma unroll
            for (int i = 0; i < MAXI; ++i) {
                const int kb = base + i * NW;
                if (kb < nkb) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        acc0[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].x, b[i][j].x, acc0[j], 0, 0, 0);
                        acc1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].x, b[i][j].x, acc1[j], 0, 0, 0);
                        acc0[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].y, b[i][j].y, acc0[j], 0, 0, 0);
                        acc1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].y, b[i][j].y, acc1[j], 0, 0, 0);
                        acc0[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].z, b[i][j].z, acc0[j], 0, 0, 0);
                        acc1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].z, b[i][j].z, acc1[j], 0, 0, 0);
                        acc0[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x0[i].w, b[i][j].w, acc0[j], 0, 0, 0);
                        acc1[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(x1[i].w, b[i][j].w, acc1[j], 0, 0, 0);
                    }
                }
            }
        }
    }

#pragma unroll
    for (int j = 0; j < NT; ++j) {
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            part[j][wave][q * 4 + v][r] = acc0[j][v];
            part[j][wave][16 + q * 4 + v][r] = acc1[j][v];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = threadIdx.x + i * NW * 64;
        const int j = e >> 9, row = (e >> 4) & 31, col = e & 15;
        if (e < NT * 512 && j < ntile && m0 + row < MT * 16) {
            float z = bias_v[i];
#pragma unroll
            for (int w = 0; w < NW; ++w) z += part[j][w][row][col];
            A.partial_out[((size_t)(tile0 + j) * MT * 16 + m0 + row) * 16 + col] = z;
        }
    }
}
