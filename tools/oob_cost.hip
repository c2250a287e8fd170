// Micro-benchmark 9 (round 2): what does an out-of-range buffer load cost the CU's load pipe?
// One workgroup of 1024 threads per CU issues 16 16-byte buffer loads per lane from a 256 KB L2-resident matrix, with a given
// share of the loads' offsets out of range (they return zero without touching memory), or skipped by a wave-uniform branch.
// Reports cycles from the first request to the last datum (s_memtime), median over workgroups, per variant.
//   hipcc --offload-arch=gfx950 -O3 -o tools/oob_cost tools/oob_cost.hip && tools/oob_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));

// mode 0: all 16 in range; 1: odd rows out of range; 2: odd rows skipped by a wave-uniform branch; 3: all out of range; 4: EXEC = 0
template <int MODE>
__global__ __launch_bounds__(1024) void k(const float* w, unsigned long long* out, float* sink, unsigned mask) {
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, 256 * 256 * 4, 0x00020000);
    const int tid = threadIdx.x;
    const unsigned off = (unsigned)(((tid >> 6) * 16 * 256 + (tid & 63) * 4) * 4);
    f32x4 r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) r[i] = f32x4{0, 0, 0, 0};
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const bool keep = (mask >> i) & 1u;
        if (MODE == 0) r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, i * 256 * 4, 0));
        if (MODE == 1 || MODE == 3) r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(keep ? off : 0x80000000u), i * 256 * 4, 0));
        if (MODE == 2) { if (keep) r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, i * 256 * 4, 0)); }
    }
    if (MODE == 4) {        // dropped rows: the load executes with EXEC = 0 (no branch, static instruction stream)
        const unsigned long long a = reinterpret_cast<unsigned long long>(w);
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 d = {(unsigned)a, (unsigned)(a >> 32) & 0xFFFFu, 256u * 256u * 4u, 0x00020000u};
        unsigned long long sv; unsigned so;
#define ROW(i) "s_bitcmp1_b32 %[m], " #i "\n\ts_cselect_b64 exec, %[sv], 0\n\tbuffer_load_dwordx4 %" #i ", %[off], %[d], %[so] offen\n\ts_add_u32 %[so], %[so], %[st]\n\t"
        asm volatile("s_mov_b64 %[sv], exec\n\ts_mov_b32 %[so], 0\n\t" ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8) ROW(9) ROW(10) ROW(11)
                         ROW(12) ROW(13) ROW(14) ROW(15) "s_mov_b64 exec, %[sv]"
                     : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]), "+v"(r[10]),
                       "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]), [sv] "=&s"(sv), [so] "=&s"(so)
                     : [off] "v"(off), [d] "s"(d), [m] "s"(mask), [st] "s"(256u * 4u)
                     : "memory", "scc");
#undef ROW
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]), "+v"(r[8]), "+v"(r[9]),
                       "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]) : : "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += r[i];
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (acc[0] + acc[1] + acc[2] + acc[3] == 1.2345f) sink[0] = acc[0];
    if (tid == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = t2 - t0; }
}
template <int MODE> void run(const char* name, const float* w, unsigned long long* out, float* sink, unsigned mask) {
    for (int rep = 0; rep < 3; ++rep) { hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, w, out, sink, mask); CK(hipDeviceSynchronize()); }
    std::vector<unsigned long long> h(512); CK(hipMemcpy(h.data(), out, 512 * 8, hipMemcpyDeviceToHost));
    std::vector<unsigned long long> a, b;
    for (int i = 0; i < 256; ++i) { a.push_back(h[2 * i]); b.push_back(h[2 * i + 1]); }
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    printf("%-60s wave 0 issued its 16 after %5llu cycles, all data in after %5llu (medians over 256 CUs)\n", name, a[128], b[128]);
}
int main() {
    float* w; unsigned long long* out; float* sink;
    CK(hipMalloc(&w, 256 * 256 * 4)); CK(hipMemset(w, 0, 256 * 256 * 4)); CK(hipMalloc(&out, 512 * 8)); CK(hipMalloc(&sink, 64));
    run<0>("16 loads per lane, all in range (256 KB per CU)", w, out, sink, 0xFFFFu);
    run<1>("8 in range + 8 out-of-range offsets", w, out, sink, 0x5555u);
    run<2>("8 in range, 8 skipped by a wave-uniform branch", w, out, sink, 0x5555u);
    run<3>("16 out-of-range offsets", w, out, sink, 0u);
    run<4>("8 in range, 8 executed with EXEC = 0 (inline asm)", w, out, sink, 0x5555u);
    run<4>("16 in range through the same asm", w, out, sink, 0xFFFFu);
    return 0;
}
