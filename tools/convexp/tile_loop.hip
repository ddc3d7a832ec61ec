// tile_loop.hip -- what does the per-wave REGISTER TILE cost?  A bare MFMA loop fed from LDS only (no global traffic, no barriers,
// no convolution bookkeeping), the same total matrix work as one 8 192 x 17 x 17 tower launch (2.79 TFLOP), in three shapes:
//   mode 0: 128 x 64 outputs per wave, two 256-thread workgroups per CU (two waves per SIMD), per K = 32 step
//           { 12 ds_read_b128 | lgkmcnt(0) | 32 MFMA } -- the operand traffic and phase shape of k_conv8w / k_conv4w / k_conv4r;
//   mode 2: the same tile with a second fragment set: the next step's 12 reads issued between this step's MFMAs;
//   mode 1: 128 x 128 outputs per wave, ONE 256-thread workgroup per CU (one wave per SIMD, 512 registers), per step 16 reads
//           for 64 MFMAs, software-pipelined like mode 2 (one read behind every 4th MFMA).
// Operands: post-ReLU-like activations and small weights (random), re-read from the wave's own LDS region every step; fragments
// are 1 KB contiguous per wave instruction (conflict-free).  Reports ms per launch; the host prints TFLOP/s.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef int intx4 __attribute__((ext_vector_type(4)));

#define TL_READ(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory")

// The 128 x 128 tile keeps 256 accumulator registers: with the builtin hipcc spreads them over VGPRs and AGPRs and moves them
// (1 124 v_accvgpr instructions and spills inside the loop); as inline asm with an "a" (AGPR) operand they stay where they are.
// No hazard is left to the compiler here: an accumulator is revisited only after 63 other MFMAs, and the fragment registers an
// MFMA reads were retired by an s_waitcnt the fragments are tied through.
#define TL_MFMA(acc_, w_, p_)                                                                                          \
    do {                                                                                                               \
        if constexpr (NT == 8) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc_) : "v"(w_), "v"(p_)); \
        else acc_ = __builtin_amdgcn_mfma_f32_16x16x32_f16(w_, p_, acc_, 0, 0, 0);                                     \
    } while (0)

template <int NT>   // channel tiles per wave: 4 (128 x 64) or 8 (128 x 128)
struct Frags {
    half8 p[8];
    half8 w[NT];
};

template <int NT>
__device__ __forceinline__ void tie(Frags<NT> &f) {
    // s_waitcnt lgkmcnt(0) with every fragment register as an in/out operand: nothing that uses them is placed above
    if constexpr (NT == 4)
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(f.p[0]), "+v"(f.p[1]), "+v"(f.p[2]), "+v"(f.p[3]), "+v"(f.p[4]), "+v"(f.p[5]), "+v"(f.p[6]), "+v"(f.p[7]),
                       "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3])::"memory");
    else
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(f.p[0]), "+v"(f.p[1]), "+v"(f.p[2]), "+v"(f.p[3]), "+v"(f.p[4]), "+v"(f.p[5]), "+v"(f.p[6]), "+v"(f.p[7]),
                       "+v"(f.w[0]), "+v"(f.w[1]), "+v"(f.w[2]), "+v"(f.w[3]), "+v"(f.w[4]), "+v"(f.w[5]), "+v"(f.w[6]), "+v"(f.w[7])::"memory");
}

// fragment i of the wave's region (8 pixel fragments, then NT weight fragments)
#define TL_READ_I(f, i, base)                                    \
    do {                                                         \
        if ((i) < 8) TL_READ((f).p[(i)], base, (i) * 1024);      \
        else TL_READ((f).w[(i) - 8], base, (i) * 1024);          \
    } while (0)

template <int MODE>
__global__ __launch_bounds__(256, MODE == 1 ? 1 : 2) void k_tile_loop(const char *__restrict__ src, float *__restrict__ out, int steps) {
    constexpr int NT = MODE == 1 ? 8 : 4, NF = 8 + NT;
    __shared__ __attribute__((aligned(1024))) char smem[4 * NF * 1024];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // fill: the wave's NF KB from a random bank (pixel fragments from its first 8 KB: activations; the rest: weights)
    for (int i = 0; i < NF; i++)
        *reinterpret_cast<intx4 *>(smem + (wid * NF + i) * 1024 + lane * 16) =
            *reinterpret_cast<const intx4 *>(src + ((size_t)((blockIdx.x * 4 + wid) % 61) * 16 + i) * 1024 + lane * 16);
    __syncthreads();
    int base = wid * NF * 1024 + lane * 16;
    floatx4 acc[8][NT];
#pragma unroll
    for (int a = 0; a < 8; a++)
#pragma unroll
        for (int b = 0; b < NT; b++) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};

    if constexpr (MODE == 0) {
        Frags<NT> f;
        for (int s = 0; s < steps; s++) {
            asm volatile("" : "+v"(base));
#pragma unroll
            for (int i = 0; i < NF; i++) TL_READ_I(f, i, base);
            tie(f);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mt = 0; mt < 8; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) TL_MFMA(acc[mt][nt], f.w[nt], f.p[mt]);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // software-pipelined: fragment sets f0 / f1 alternate; one read of the NEXT step's set behind every (8 * NT / NF)-th MFMA
        Frags<NT> f0, f1;
        constexpr int NM = 8 * NT, EVERY = NM / NF > 0 ? (NM + NF - 1) / NF : 1;
#pragma unroll
        for (int i = 0; i < NF; i++) TL_READ_I(f0, i, base);
        tie(f0);
#define TL_STEP(cur, nxt)                                                                                              \
    do {                                                                                                               \
        asm volatile("" : "+v"(base));                                                                                 \
        _Pragma("unroll") for (int m = 0; m < NM; m++) {                                                               \
            const int mt = m / NT, nt = m % NT;                                                                        \
            TL_MFMA(acc[mt][nt], cur.w[nt], cur.p[mt]);                                                                \
            if (m % EVERY == EVERY - 1 && m / EVERY < NF) {                                                            \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
                TL_READ_I(nxt, m / EVERY, base);                                                                       \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }                                                                                                          \
        }                                                                                                              \
        _Pragma("unroll") for (int i = NM / EVERY; i < NF; i++) TL_READ_I(nxt, i, base);                               \
        tie(nxt);                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
    } while (0)
        for (int s = 0; s < steps; s += 2) {
            TL_STEP(f0, f1);
            TL_STEP(f1, f0);
        }
#undef TL_STEP
    }
    float t = 0.f;
#pragma unroll
    for (int a = 0; a < 8; a++)
#pragma unroll
        for (int b = 0; b < NT; b++) t += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    out[blockIdx.x * 256 + tid] = t;
}

extern "C" int tile_loop_run(int mode, int steps, int iters, const void *src, void *out, float *ms) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = mode == 1 ? 256 : 512;
    for (int it = 0; it < iters + 2; it++) {
        if (it == 2) hipEventRecord(e0, 0);
        if (mode == 0) hipLaunchKernelGGL(k_tile_loop<0>, dim3(grid), dim3(256), 0, 0, (const char *)src, (float *)out, steps);
        else if (mode == 1) hipLaunchKernelGGL(k_tile_loop<1>, dim3(grid), dim3(256), 0, 0, (const char *)src, (float *)out, steps);
        else hipLaunchKernelGGL(k_tile_loop<2>, dim3(grid), dim3(256), 0, 0, (const char *)src, (float *)out, steps);
    }
    hipEventRecord(e1, 0);
    hipError_t rc = hipDeviceSynchronize();
    hipEventElapsedTime(ms, e0, e1);
    *ms /= iters;
    return (int)rc;
}
