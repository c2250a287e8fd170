// Small reductions shared by the per-utterance chain of the fused front kernel (dec_front.hip) and of the persistent decode
// kernel (persist_decode.hip): the two must sum in the same order (their results are compared bitwise).
#pragma once
#include <hip/hip_runtime.h>
#include "device_utils.h"

// acc += x * r, one fused multiply-add per component.  EXPLICIT: `a += b * c` leaves the fusion to the optimiser, call site by call
// site; the chain exists in three kernels (general, lean, persistent) whose outputs are compared bitwise.
__device__ __forceinline__ void gt_fma4(float4& acc, const float x, const float4& r) {
    acc.x = __builtin_fmaf(x, r.x, acc.x); acc.y = __builtin_fmaf(x, r.y, acc.y);
    acc.z = __builtin_fmaf(x, r.z, acc.z); acc.w = __builtin_fmaf(x, r.w, acc.w);
}

// sum of the k-part partials of one column; 8 independent LDS reads in flight per round (a plain
// `z += partial[...]` loop serialises ~100-cycle LDS round trips: 32 of them cost >1 us per phase)
__device__ __forceinline__ float reduce_partial(const float* partial, int kparts, int N, int col) {
    float z = 0.f;
    int p = 0;
    for (; p + 8 <= kparts; p += 8) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = partial[(size_t)(p + j) * N + col];
        z += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
    }
    for (; p < kparts; ++p) z += partial[(size_t)p * N + col];
    return z;
}

__device__ __forceinline__ float front_wave_incl_scan(float x, int lane) { return gt_wave_incl_scan(x, lane); }
