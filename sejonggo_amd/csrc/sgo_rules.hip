// sgo_rules.hip -- batch Go-rules kernels for gfx950 and the stateless half of the C ABI (include/sgo.h).
//
// Kernels (all one-lane-per-position unless noted, see sgo_bits.hpp):
//   k_advance_legal  board_advance: make_play + legal set of the new position      (play.py:226-242, :71-104)
//   k_legal          legal set of a position                                       (play.py:71-104)
//   k_score          area score                                                    (play.py:244-292)
//   k_pack/k_unpack  int32 [S][S][17] board tensor <-> packed bit-planes (one thread per point, ballot)
//   k_nn_pack        packed position -> fp16/fp32 network input, symmetry fused    (symmetry.py:45-114)
//   k_sym_apply      symmetry on the raw int32 tensor                              (symmetry.py:45-114)
//   k_sym_policy     out[a] = in[SWAP[a]]                                          (symmetry.py reverse_*)
#include <hip/hip_fp16.h>
#include <math.h>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "sgo_bits.hpp"
#include "sgo_rows.hpp"
#include "sgo_common.hpp"

namespace sgo {

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
void set_error(const std::string &msg) { g_err = msg; }
int hip_fail(hipError_t e, const char *what, const char *file, int line) {
    char buf[512];
    snprintf(buf, sizeof buf, "HIP error %d (%s) at %s:%d: %s", (int)e, hipGetErrorString(e), file, line, what);
    g_err = buf;
    return SGO_ERR_HIP;
}

// ------------------------------------------------------------------------------------ symmetry maps
// transformed[i][j] = source[si][sj]  (row i, column j), symmetry.py:45-114
__host__ __device__ inline void sym_src(int S, int k, int i, int j, int &si, int &sj) {
    si = i; sj = j;
    switch (k) {
    case 1: si = j; sj = i; break;
    case 2: sj = S - 1 - j; break;
    case 3: si = S - 1 - i; break;
    case 4: si = j; sj = S - 1 - i; break;
    case 5: si = S - 1 - i; sj = S - 1 - j; break;
    case 6: si = S - 1 - j; sj = i; break;
    case 7: si = S - 1 - j; sj = S - 1 - i; break;
    default: break;
    }
}

void build_sym_lut(int S, int k, int32_t *lut) {
    const double pi = 3.141592653589793;
    const double c = (S - 1) / 2.0;
    for (int i = 0; i <= S * S; i++) lut[i] = i;
    if (k == 0) return;
    const bool rot = (k >= 4 && k <= 6);
    double angle = 0;
    switch (k) {
    case 1: angle = pi / 4.; break;       // left diagonal   (axis_symmetry_indexes)
    case 2: angle = pi / 2.; break;       // vertical axis
    case 3: angle = 0; break;             // horizontal axis
    case 4: angle = pi / 2.; break;       // rotation_indexes
    case 5: angle = pi; break;
    case 6: angle = 3 * pi / 2; break;
    case 7: angle = 3 * pi / 4.; break;   // right diagonal
    }
    for (int x = 0; x < S; x++)
        for (int y = 0; y < S; y++) {
            double fx = x - c, fy = y - c, nx, ny;
            if (rot) {
                nx = cos(angle) * fx - sin(angle) * fy;
                ny = sin(angle) * fx + cos(angle) * fy;
            } else {
                nx = cos(2 * angle) * fx + sin(2 * angle) * fy;
                ny = sin(2 * angle) * fx - cos(2 * angle) * fy;
            }
            lut[x + S * y] = (int32_t)lrint((nx + c) + S * (ny + c));
        }
}

// ------------------------------------------------------------------------------------ kernels
template <int S>
__global__ __launch_bounds__(256) void k_advance_legal(int n, const uint32_t *in, const int32_t *in_idx,
                                                       const int32_t *moves, const int32_t *colors, uint32_t *out,
                                                       const int32_t *out_idx, uint32_t *legal,
                                                       const int32_t *legal_idx, int32_t *status) {
    using G = Geo<S>;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *src = in + (size_t)(in_idx ? in_idx[i] : i) * G::RW;
    uint32_t *dst = out + (size_t)(out_idx ? out_idx[i] : i) * G::RW;
    uint32_t *lg = legal ? legal + (size_t)(legal_idx ? legal_idx[i] : i) * G::NW : nullptr;
    bool swap_first = false;
    int mover = white_to_play<S>(src) ? -1 : 1;
    if (colors) {
        int c = colors[i];
        if (c != 0 && c != mover) { swap_first = true; mover = -mover; }
    }
    int st = advance_record<S>(src, dst, moves[i], swap_first, lg);
    if (status) status[i] = st ? st : mover;
}

// ---- split form of board_advance for non-aliasing in/out (the engine's leaf step, dense out-of-place batches)
// k_history_shift: pure streaming move of the 14 history planes (new plane p <- old plane p-1 / p-3, i.e. the
// reference's shift + pair swap), one thread per 16-B chunk (4-B word when a plane is not a multiple of 16 B):
// consecutive threads touch consecutive chunks of one record, so every wave instruction covers whole records.
template <int S, bool NT = true>
__device__ __forceinline__ void history_shift_body(int n, long t0, long stride, const uint32_t *in, const int32_t *in_idx,
                                                   const int32_t *colors, uint32_t *out, const int32_t *out_idx) {
    using G = Geo<S>;
    (void)colors;                                   // colours are absolute in the record: an override moves no plane
    constexpr bool V4 = (G::NW % 4 == 0);
    constexpr int WPC = V4 ? 4 : 1;                 // words per chunk
    constexpr int CPR = 14 * G::NW / WPC;           // chunks per record: planes 0..13 -> planes 2..15, one contiguous run
    const long total = (long)n * CPR;
    auto addr = [&](long t, const uint32_t *&sp_, uint32_t *&dp_) {
        const int i = (int)(t / CPR), c = (int)(t - (long)i * CPR);
        sp_ = in + (size_t)(in_idx ? in_idx[i] : i) * G::RW + c * WPC;
        dp_ = out + (size_t)(out_idx ? out_idx[i] : i) * G::RW + 2 * G::NW + c * WPC;
    };
    for (long t = t0; t < total; t += 4 * stride) {
        // four independent chunks in flight per thread
        const uint32_t *s0, *s1, *s2, *s3;
        uint32_t *d0, *d1, *d2, *d3;
        const bool b1 = t + stride < total, b2 = t + 2 * stride < total, b3 = t + 3 * stride < total;
        addr(t, s0, d0);
        addr(b1 ? t + stride : t, s1, d1);
        addr(b2 ? t + 2 * stride : t, s2, d2);
        addr(b3 ? t + 3 * stride : t, s3, d3);
        if constexpr (V4) {
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            // streamed once: non-temporal loads and stores keep the history move out of the L2 working set
            if constexpr (NT) {
                u32x4 v0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(s0));
                u32x4 v1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(s1));
                u32x4 v2 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(s2));
                u32x4 v3 = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(s3));
                __builtin_nontemporal_store(v0, reinterpret_cast<u32x4 *>(d0));
                if (b1) __builtin_nontemporal_store(v1, reinterpret_cast<u32x4 *>(d1));
                if (b2) __builtin_nontemporal_store(v2, reinterpret_cast<u32x4 *>(d2));
                if (b3) __builtin_nontemporal_store(v3, reinterpret_cast<u32x4 *>(d3));
            } else {
                u32x4 v0 = *reinterpret_cast<const u32x4 *>(s0), v1 = *reinterpret_cast<const u32x4 *>(s1);
                u32x4 v2 = *reinterpret_cast<const u32x4 *>(s2), v3 = *reinterpret_cast<const u32x4 *>(s3);
                *reinterpret_cast<u32x4 *>(d0) = v0;
                if (b1) *reinterpret_cast<u32x4 *>(d1) = v1;
                if (b2) *reinterpret_cast<u32x4 *>(d2) = v2;
                if (b3) *reinterpret_cast<u32x4 *>(d3) = v3;
            }
        } else {
            uint32_t v0 = *s0, v1 = *s1, v2 = *s2, v3 = *s3;
            *d0 = v0;
            if (b1) *d1 = v1;
            if (b2) *d2 = v2;
            if (b3) *d3 = v3;
        }
    }
}

template <int S, bool NT = true>
__global__ __launch_bounds__(256) void k_history_shift(int n, const int *n_dev, const uint32_t *in, const int32_t *in_idx,
                                                       const int32_t *colors, uint32_t *out, const int32_t *out_idx) {
    if (n_dev) n = *n_dev;
    history_shift_body<S, NT>(n, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x, in, in_idx, colors, out,
                          out_idx);
}

template <int S>
__global__ __launch_bounds__(64) void k_advance_planes(int n, const int *n_dev, const uint32_t *in, const int32_t *in_idx,
                                                       const int32_t *moves, const int32_t *colors, uint32_t *out,
                                                       const int32_t *out_idx, uint32_t *legal, const int32_t *legal_idx,
                                                       int32_t *status) {
    using G = Geo<S>;
    if (n_dev) n = *n_dev;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *src = in + (size_t)(in_idx ? in_idx[i] : i) * G::RW;
    uint32_t *dst = out + (size_t)(out_idx ? out_idx[i] : i) * G::RW;
    uint32_t *lg = legal ? legal + (size_t)(legal_idx ? legal_idx[i] : i) * G::NW : nullptr;
    bool swap_first = false;
    int mover = white_to_play<S>(src) ? -1 : 1;
    if (colors) {
        int c = colors[i];
        if (c != 0 && c != mover) { swap_first = true; mover = -mover; }
    }
    int st = advance_planes<S>(src, dst, moves[i], swap_first, lg);
    if (status) status[i] = st ? st : mover;
}

// One launch, two kinds of 64-thread blocks: blocks [0, n_compute_blocks) run the register kernel (one lane per leaf),
// the rest stream the history planes.  For the engine's small per-step launches this halves the launch latency of
// board_advance; the two kinds touch disjoint words of the output records.
template <int S>
__global__ __launch_bounds__(64) void k_board_advance(int n, const int *n_dev, int n_compute_blocks, const uint32_t *in,
                                                      const int32_t *in_idx, const int32_t *moves, const int32_t *colors,
                                                      uint32_t *out, const int32_t *out_idx, uint32_t *legal,
                                                      const int32_t *legal_idx, int32_t *status) {
    using G = Geo<S>;
    if (n_dev) n = *n_dev;
    if ((int)blockIdx.x >= n_compute_blocks) {
        const long b = (long)blockIdx.x - n_compute_blocks;
        history_shift_body<S>(n, b * 64 + threadIdx.x, (long)(gridDim.x - n_compute_blocks) * 64, in, in_idx, colors, out, out_idx);
        return;
    }
    int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const uint32_t *src = in + (size_t)(in_idx ? in_idx[i] : i) * G::RW;
    uint32_t *dst = out + (size_t)(out_idx ? out_idx[i] : i) * G::RW;
    uint32_t *lg = legal ? legal + (size_t)(legal_idx ? legal_idx[i] : i) * G::NW : nullptr;
    bool swap_first = false;
    int mover = white_to_play<S>(src) ? -1 : 1;
    if (colors) {
        int c = colors[i];
        if (c != 0 && c != mover) { swap_first = true; mover = -mover; }
    }
    int st = advance_planes<S>(src, dst, moves[i], swap_first, lg);
    if (status) status[i] = st ? st : mover;
}

// The row-per-lane form (sgo_rows.hpp): one 32-lane half per leaf, history move included.  For the engine's small
// per-step launches: 8 192 leaves are 4 096 wavefronts here instead of 128, and a leaf's latency is a few hundred
// instructions instead of a few thousand.
template <int S>
__global__ __launch_bounds__(64) void k_board_advance_rows(int n, const int *n_dev, const uint32_t *in, const int32_t *in_idx,
                                                           const int32_t *moves, const int32_t *colors, uint32_t *out,
                                                           const int32_t *out_idx, uint32_t *legal, const int32_t *legal_idx,
                                                           int32_t *status) {
    using G = Geo<S>;
    if (n_dev) n = *n_dev;
    const int half = threadIdx.x >> 5, y = threadIdx.x & 31;
    const int i = blockIdx.x * 2 + half;
    if (i >= n) return;
    const uint32_t *src = in + (size_t)(in_idx ? in_idx[i] : i) * G::RW;
    uint32_t *dst = out + (size_t)(out_idx ? out_idx[i] : i) * G::RW;
    uint32_t *lg = legal ? legal + (size_t)(legal_idx ? legal_idx[i] : i) * G::NW : nullptr;
    bool swap_first = false;
    int mover = white_to_play<S>(src) ? -1 : 1;
    if (colors) {
        int c = colors[i];
        if (c != 0 && c != mover) { swap_first = true; mover = -mover; }
    }
    const int st = rows::advance_record_rows<S>(src, dst, moves[i], swap_first, lg, half, y);
    if (status && y == 0) status[i] = st ? st : mover;
}


template <int S>
__global__ __launch_bounds__(256) void k_legal(int n, const uint32_t *packed, const int32_t *idx, uint32_t *legal) {
    using G = Geo<S>;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    legal_record<S>(packed + (size_t)(idx ? idx[i] : i) * G::RW, legal + (size_t)i * G::NW);
}

template <int S>
__global__ __launch_bounds__(256) void k_score(int n, const uint32_t *packed, const int32_t *idx, double komi,
                                               int32_t *result) {
    using G = Geo<S>;
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int bp, wp;
    score_record<S>(packed + (size_t)(idx ? idx[i] : i) * G::RW, bp, wp);
    double white = (double)wp + komi;  // play.py:278
    int w = ((double)bp > white) ? 1 : (((double)bp == white) ? 0 : -1);
    result[3 * i] = w;
    result[3 * i + 1] = bp;
    result[3 * i + 2] = wp;
}

// one block per board, one thread per point (rounded up to whole waves); wave ballots build the words.
// The board tensor's planes are relative to the side to move (2k = to-play, 2k+1 = opponent); the record's are
// absolute (2k = black, 2k+1 = white): relative plane c maps to absolute plane c ^ (white to play).
// Group / territory queries on plain boards for the drop-in helpers play.capture_group / get_liberties / color_board
// (play.py:55-69, :159-180, :244-271).  cells: int8 [n][S][S], +1 black, -1 white, 0 empty, anything else = wall (neither
// a stone nor a liberty: used to embed smaller or rectangular arrays).  One lane per board.
//   mode 0: member = the seed point plus every stone of colour `color` connected to it through stones of that colour;
//           liberty = empty points next to a member (members themselves excluded)
//   mode 1: member = empty points connected, through empty points, to a stone of colour `color` (color_board's fill)
template <int S>
__global__ __launch_bounds__(64) void k_board_query(int n, int mode, const int8_t *cells, const int32_t *xs, const int32_t *ys,
                                                     const int32_t *colors, uint8_t *member, uint8_t *liberty) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int8_t *c = cells + (size_t)i * S * S;
    uint32_t bl[S], wh[S], em[S], grp[S], lib[S], t[S];
#pragma unroll
    for (int y = 0; y < S; y++) {
        uint32_t b = 0, w = 0, e = 0;
        for (int x = 0; x < S; x++) {
            const int v = c[y * S + x];
            b |= (uint32_t)(v == 1) << x;
            w |= (uint32_t)(v == -1) << x;
            e |= (uint32_t)(v == 0) << x;
        }
        bl[y] = b; wh[y] = w; em[y] = e;
    }
    const int col = colors[i];
    if (mode == 0) {
        const int sx = xs[i], sy = ys[i];
        uint32_t m[S];
#pragma unroll
        for (int y = 0; y < S; y++) {
            grp[y] = (y == sy) ? (1u << sx) : 0u;
            m[y] = (col == 1 ? bl[y] : col == -1 ? wh[y] : 0u) | grp[y];
        }
        flood<S>(grp, m);
        nbr4<S>(grp, t);
#pragma unroll
        for (int y = 0; y < S; y++) lib[y] = t[y] & em[y] & ~grp[y];
    } else {
#pragma unroll
        for (int y = 0; y < S; y++) t[y] = col == 1 ? bl[y] : wh[y];
        nbr4<S>(t, grp);
#pragma unroll
        for (int y = 0; y < S; y++) {
            grp[y] &= em[y];
            lib[y] = 0;
        }
        flood<S>(grp, em);
    }
    uint8_t *mo = member + (size_t)i * S * S, *lo = liberty + (size_t)i * S * S;
#pragma unroll
    for (int y = 0; y < S; y++)
        for (int x = 0; x < S; x++) {
            mo[y * S + x] = (grp[y] >> x) & 1u;
            lo[y * S + x] = (lib[y] >> x) & 1u;
        }
}

// take_stones (play.py:182-217) on board tensors in place: planes 0 (to-play side) and 1 (opponent) only.
template <int S>
__global__ __launch_bounds__(64) void k_take_stones(int n, int32_t *boards, const int32_t *xs, const int32_t *ys) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t *b = boards + (size_t)i * S * S * 17;
    uint32_t own[S], opp[S];
#pragma unroll
    for (int y = 0; y < S; y++) {
        uint32_t o = 0, p = 0;
        for (int x = 0; x < S; x++) {
            o |= (uint32_t)(b[(y * S + x) * 17] != 0) << x;
            p |= (uint32_t)(b[(y * S + x) * 17 + 1] != 0) << x;
        }
        own[y] = o; opp[y] = p;
    }
    take_core<S>(own, opp, ys[i] * S + xs[i]);
#pragma unroll
    for (int y = 0; y < S; y++)
        for (int x = 0; x < S; x++) {
            if (!((own[y] >> x) & 1u)) b[(y * S + x) * 17] = 0;
            if (!((opp[y] >> x) & 1u)) b[(y * S + x) * 17 + 1] = 0;
        }
}

template <int S>
__global__ void k_pack(int n, const int32_t *boards, uint32_t *packed) {
    using G = Geo<S>;
    int b = blockIdx.x;
    int t = threadIdx.x;
    const int32_t *src = boards + (size_t)b * G::N * 17;
    uint32_t *dst = packed + (size_t)b * G::RW;
    int wave = t >> 6, lane = t & 63;
    const int flip = (src[16] == -1) ? 1 : 0;
    for (int c = 0; c < 16; c++) {
        int v = (t < G::N) ? (src[t * 17 + (c ^ flip)] != 0) : 0;   // absolute plane c comes from relative plane c ^ flip
        unsigned long long m = __ballot(v);
        if (lane == 0) {
            uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
            if (c == 0 && flip) {
                if (2 * wave == G::META_WORD) lo |= G::META_BIT;
                if (2 * wave + 1 == G::META_WORD) hi |= G::META_BIT;
            }
            if (2 * wave < G::NW) dst[c * G::NW + 2 * wave] = lo;
            if (2 * wave + 1 < G::NW) dst[c * G::NW + 2 * wave + 1] = hi;
        }
    }
}

template <int S>
__global__ void k_unpack(int n, const uint32_t *packed, int32_t *boards) {
    using G = Geo<S>;
    int b = blockIdx.x;
    int t = threadIdx.x;
    if (t >= G::N) return;
    const uint32_t *src = packed + (size_t)b * G::RW;
    int32_t *dst = boards + ((size_t)b * G::N + t) * 17;
    const int flip = white_to_play<S>(src) ? 1 : 0;
    for (int c = 0; c < 16; c++) dst[c] = (src[(c ^ flip) * G::NW + (t >> 5)] >> (t & 31)) & 1u;
    dst[16] = flip ? -1 : 1;
}

// network input: out[i] = symmetry_k(position idx[i]).  One thread per (entry, point).
template <int S, typename T>
__global__ __launch_bounds__(256) void k_nn_pack(int n, const uint32_t *packed, const int32_t *idx, int k, int layout,
                                                 T *out) {
    using G = Geo<S>;
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * G::N) return;
    int e = gid / G::N, pt = gid - e * G::N;
    int i = pt / S, j = pt - i * S, si, sj;
    sym_src(S, k, i, j, si, sj);
    int sp = si * S + sj;
    const uint32_t *rec = packed + (size_t)(idx ? idx[e] : e) * G::RW;
    T vals[17];
    const int flip = white_to_play<S>(rec) ? 1 : 0;   // network planes are relative to the side to move
#pragma unroll
    for (int c = 0; c < 16; c++) vals[c] = (T)(float)((rec[(c ^ flip) * G::NW + (sp >> 5)] >> (sp & 31)) & 1u);
    vals[16] = (T)(flip ? -1.0f : 1.0f);
    if (layout == 0) {  // NHWC
        T *o = out + ((size_t)e * G::N + pt) * 17;
#pragma unroll
        for (int c = 0; c < 17; c++) o[c] = vals[c];
    } else if (layout == 2) {  // NHWC, channels zero-padded to 32 (MFMA-friendly K for the stem convolution)
        T *o = out + ((size_t)e * G::N + pt) * 32;
        if constexpr (sizeof(T) == 2) {
            // 64 B per point as four 16-B stores
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            h8 v[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int c = 0; c < 8; c++) v[q][c] = (q * 8 + c < 17) ? (_Float16)(float)vals[q * 8 + c] : (_Float16)0.0f;
#pragma unroll
            for (int q = 0; q < 4; q++) reinterpret_cast<h8 *>(o)[q] = v[q];
        } else {
#pragma unroll
            for (int c = 0; c < 17; c++) o[c] = vals[c];
#pragma unroll
            for (int c = 17; c < 32; c++) o[c] = (T)0.0f;
        }
    } else {  // NCHW
        T *o = out + (size_t)e * 17 * G::N + pt;
#pragma unroll
        for (int c = 0; c < 17; c++) o[(size_t)c * G::N] = vals[c];
    }
}

__global__ void k_sym_apply(int S, int k, int n, const int32_t *in, int32_t *out) {
    int N = S * S;
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * N) return;
    int b = gid / N, pt = gid - b * N;
    int i = pt / S, j = pt - i * S, si, sj;
    sym_src(S, k, i, j, si, sj);
    const int32_t *s = in + ((size_t)b * N + si * S + sj) * 17;
    int32_t *d = out + ((size_t)b * N + pt) * 17;
    for (int c = 0; c < 17; c++) d[c] = s[c];
}

__global__ void k_sym_policy(int A, int n, const int32_t *lut, const float *in, float *out) {
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * A) return;
    int b = gid / A, a = gid - b * A;
    out[gid] = in[(size_t)b * A + lut[a]];
}

__global__ void k_legal_to_mask(int A, int NW, int n, const uint32_t *legal, uint8_t *mask) {
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n * A) return;
    int b = gid / A, a = gid - b * A;
    uint32_t bit = (legal[(size_t)b * NW + (a >> 5)] >> (a & 31)) & 1u;
    mask[gid] = bit ? 0 : 1;  // the reference's mask: 1 = illegal
}

// Fused convolution epilogue for the resident net: out = relu(x + bias[c] (+ skip)), NHWC fp16, 8 halves
// (16 B) per thread.  One pass instead of MIOpen's separate bias / add / clamp passes.
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_bias_act(long n8, int C8, const half8_t *x, const half8_t *bias,
                                                  const half8_t *skip, half8_t *out) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    const half8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
    for (; i < n8; i += stride) {
        half8_t r = x[i] + bias[i % C8];
        if (skip) r = r + skip[i];
        out[i] = __builtin_elementwise_max(r, zero);
    }
}

// ------------------------------------------------------------------------------------ launchers

// Two launches on the caller's stream: the streaming history move, then the register kernel.  (Running the two
// on separate streams was measured: they contend for the memory system at low stone density and the fork/join
// events cost more than the overlap saves at the engine's 8 192-leaf launches.)
// SGO_ADV_MODE / sgo_advance_mode(): 0 two launches (history stream + lane-per-leaf), 1 one heterogeneous launch of the
// same two, 2 row-per-lane (one half-wave per leaf), -1 (default): by batch size
static constexpr int ROWS_MAX_LEAVES = 1 << 15;   // above this the lane-per-leaf form has the chip filled
static int g_adv_mode = -2;
static int adv_mode() {
    if (g_adv_mode == -2) {
        const char *e = getenv("SGO_ADV_MODE");
        g_adv_mode = e ? atoi(e) : -1;
    }
    return g_adv_mode;
}
int set_advance_mode(int m) {
    const int old = adv_mode();
    if (m >= -1 && m <= 2) g_adv_mode = m;
    return old;
}

int launch_advance_split(int S, int n_max, const int *d_n, const uint32_t *d_in, const int32_t *d_in_idx,
                         const int32_t *d_moves, const int32_t *d_colors, uint32_t *d_out, const int32_t *d_out_idx,
                         uint32_t *d_legal, const int32_t *d_legal_idx, int32_t *d_status, hipStream_t st) {
    if (n_max <= 0) return SGO_OK;
    int mode = adv_mode();
    if (mode < 0) mode = (n_max <= ROWS_MAX_LEAVES) ? 2 : (n_max <= (1 << 16)) ? 1 : 0;
    SGO_DISPATCH(S, {
        if (mode == 2) {
            k_board_advance_rows<kS><<<dim3(cdiv(n_max, 2)), dim3(64), 0, st>>>(n_max, d_n, d_in, d_in_idx, d_moves, d_colors, d_out,
                                                                               d_out_idx, d_legal, d_legal_idx, d_status);
        } else
        {
        constexpr int CPR = (Geo<kS>::NW % 4 == 0) ? 14 * Geo<kS>::NW / 4 : 14 * Geo<kS>::NW;
        if (mode == 1) {
            const int nbc = cdiv(n_max, 64);
            long nbs = ((long)n_max * CPR + 255) / 256;
            if (nbs > 256 * 32) nbs = 256 * 32;
            if (nbs < 1) nbs = 1;
            k_board_advance<kS><<<dim3((unsigned)(nbc + nbs)), dim3(64), 0, st>>>(n_max, d_n, nbc, d_in, d_in_idx, d_moves, d_colors,
                                                                                 d_out, d_out_idx, d_legal, d_legal_idx, d_status);
        } else {
            long blocks = ((long)n_max * CPR + 1023) / 1024;
            if (blocks > 256 * 64) blocks = 256 * 64;
            if (blocks < 1) blocks = 1;
            static const int nt = getenv("SGO_SHIFT_NT") ? atoi(getenv("SGO_SHIFT_NT")) : 1;
            static const int bs = getenv("SGO_SHIFT_BLOCKS") ? atoi(getenv("SGO_SHIFT_BLOCKS")) : 0;
            if (bs > 0) blocks = bs;
            if (nt)
                k_history_shift<kS, true><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(n_max, d_n, d_in, d_in_idx, d_colors, d_out, d_out_idx);
            else
                k_history_shift<kS, false><<<dim3((unsigned)blocks), dim3(256), 0, st>>>(n_max, d_n, d_in, d_in_idx, d_colors, d_out, d_out_idx);
            k_advance_planes<kS><<<dim3(cdiv(n_max, 64)), dim3(64), 0, st>>>(n_max, d_n, d_in, d_in_idx, d_moves, d_colors, d_out,
                                                                             d_out_idx, d_legal, d_legal_idx, d_status);
        }
        }
    });
    SGO_HIP(hipGetLastError());
    return SGO_OK;
}


int launch_advance_legal(int S, int n, const uint32_t *d_in, const int32_t *d_in_idx, const int32_t *d_moves,
                         const int32_t *d_colors, uint32_t *d_out, const int32_t *d_out_idx, uint32_t *d_legal,
                         const int32_t *d_legal_idx, int32_t *d_status, hipStream_t st) {
    if (n <= 0) return SGO_OK;
    const size_t rw = (size_t)sgo_packed_words(S);
    // dense, provably non-overlapping in/out => streaming history move + register kernel; anything that may alias
    // (in place, index lists) => the fused per-lane kernel, which is safe record-for-record in place
    if (!d_in_idx && !d_out_idx && (d_out + (size_t)n * rw <= d_in || d_in + (size_t)n * rw <= d_out) && size_ok(S))
        return launch_advance_split(S, n, nullptr, d_in, nullptr, d_moves, d_colors, d_out, nullptr, d_legal, d_legal_idx,
                                    d_status, st);
    // 64-thread blocks: a block is one wavefront = 64 positions, so small batches still spread over CUs
    SGO_DISPATCH(S, k_advance_legal<kS><<<dim3(cdiv(n, 64)), dim3(64), 0, st>>>(n, d_in, d_in_idx, d_moves, d_colors, d_out, d_out_idx, d_legal, d_legal_idx, d_status));
    SGO_HIP(hipGetLastError());
    return SGO_OK;
}
int launch_nn_pack(int S, int n, const uint32_t *d_packed, const int32_t *d_idx, int k, int layout, int dtype,
                   void *d_out, hipStream_t st) {
    if (n <= 0) return SGO_OK;
    if (dtype == 0) {
        SGO_DISPATCH(S, k_nn_pack<kS, __half><<<dim3(cdiv((long)n * kS * kS, 256)), dim3(256), 0, st>>>(n, d_packed, d_idx, k, layout, (__half *)d_out));
    } else {
        SGO_DISPATCH(S, k_nn_pack<kS, float><<<dim3(cdiv((long)n * kS * kS, 256)), dim3(256), 0, st>>>(n, d_packed, d_idx, k, layout, (float *)d_out));
    }
    SGO_HIP(hipGetLastError());
    return SGO_OK;
}
int launch_score(int S, int n, const uint32_t *d_packed, const int32_t *d_idx, double komi, int32_t *d_result,
                 hipStream_t st) {
    if (n <= 0) return SGO_OK;
    SGO_DISPATCH(S, k_score<kS><<<dim3(cdiv(n, 64)), dim3(64), 0, st>>>(n, d_packed, d_idx, komi, d_result));
    SGO_HIP(hipGetLastError());
    return SGO_OK;
}

// grow-only device scratch for the host-buffer entry points
struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return SGO_OK;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 2 + 4096;
        SGO_HIP(hipMalloc(&p, want));
        cap = want;
        return SGO_OK;
    }
};
static std::mutex g_mu;
static Scratch g_s[4];

template <int S>
static constexpr int pack_threads() { return ((Geo<S>::N + 63) / 64) * 64; }

}  // namespace sgo

using namespace sgo;

extern "C" {

const char *sgo_last_error(void) { return g_err.c_str(); }
int sgo_version(void) { return SGO_ABI_VERSION; }
int sgo_advance_mode(int mode) { return set_advance_mode(mode); }
int sgo_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int sgo_set_device(int d) {
    SGO_HIP(hipSetDevice(d));
    return SGO_OK;
}

int sgo_plane_words(int S) { return size_ok(S) ? (S * S + 31) / 32 : SGO_ERR_ARG; }
int sgo_packed_words(int S) { return size_ok(S) ? 16 * ((S * S + 31) / 32) : SGO_ERR_ARG; }
int sgo_apad(int S) { return size_ok(S) ? 32 * ((S * S + 31) / 32) : SGO_ERR_ARG; }

int sgo_sym_lut(int S, int k, int32_t *lut) {
    if (!size_ok(S) || k < 0 || k > 7 || !lut) { set_error("sgo_sym_lut: bad argument"); return SGO_ERR_ARG; }
    build_sym_lut(S, k, lut);
    return SGO_OK;
}

// ---- device-pointer API
int sgo_pack_dev(int S, int n, const int32_t *d_board17, uint32_t *d_packed, void *stream) {
    if (n <= 0) return SGO_OK;
    SGO_DISPATCH(S, k_pack<kS><<<dim3(n), dim3(pack_threads<kS>()), 0, (hipStream_t)stream>>>(n, d_board17, d_packed));
    SGO_HIP(hipGetLastError());
    return SGO_OK;
}
int sgo_unpack_dev(int S, int n, const uint32_t *d_packed, int32_t *d_board17, void *stream) {
    if (n <= 0) return SGO_OK;
    SGO_DISPATCH(S, k_unpack<kS><<<dim3(n), dim3(pack_threads<kS>()), 0, (hipStream_t)stream>>>(n, d_packed, d_board17));
    SGO_HIP(hipGetLastError());
    return SGO_OK;
}
int sgo_advance_legal_dev(int S, int n, const uint32_t *d_in, const int32_t *d_in_idx, const int32_t *d_moves,
                          const int32_t *d_colors, uint32_t *d_out, const int32_t *d_out_idx, uint32_t *d_legal,
                          int32_t *d_status, void *stream) {
    if (!size_ok(S) || n < 0 || !d_in || !d_out || !d_moves) { set_error("sgo_advance_legal_dev: bad argument"); return SGO_ERR_ARG; }
    return launch_advance_legal(S, n, d_in, d_in_idx, d_moves, d_colors, d_out, d_out_idx, d_legal, nullptr, d_status,
                                (hipStream_t)stream);
}
int sgo_legal_dev(int S, int n, const uint32_t *d_packed, const int32_t *d_idx, uint32_t *d_legal, void *stream) {
    if (n <= 0) return SGO_OK;
    SGO_DISPATCH(S, k_legal<kS><<<dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream>>>(n, d_packed, d_idx, d_legal));
    SGO_HIP(hipGetLastError());
    return SGO_OK;
}
int sgo_score_dev(int S, int n, const uint32_t *d_packed, const int32_t *d_idx, double komi, int32_t *d_result,
                  void *stream) {
    if (!size_ok(S)) { set_error("sgo_score_dev: bad size"); return SGO_ERR_ARG; }
    return launch_score(S, n, d_packed, d_idx, komi, d_result, (hipStream_t)stream);
}
int sgo_nn_pack_dev(int S, int n, const uint32_t *d_packed, const int32_t *d_idx, int k, int layout, int dtype,
                    void *d_out, void *stream) {
    if (!size_ok(S) || k < 0 || k > 7 || layout < 0 || layout > 2 || dtype < 0 || dtype > 1) {
        set_error("sgo_nn_pack_dev: bad argument");
        return SGO_ERR_ARG;
    }
    return launch_nn_pack(S, n, d_packed, d_idx, k, layout, dtype, d_out, (hipStream_t)stream);
}

int sgo_bias_act_dev(long n_elems, int channels, const void *d_x, const void *d_bias, const void *d_skip, void *d_out,
                     void *stream) {
    if (n_elems < 0 || channels <= 0 || channels % 8 || n_elems % 8 || !d_x || !d_bias || !d_out) {
        set_error("sgo_bias_act_dev: sizes must be multiples of 8 halves");
        return SGO_ERR_ARG;
    }
    if (n_elems == 0) return SGO_OK;
    const long n8 = n_elems / 8;
    long blocks = (n8 + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    k_bias_act<<<dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream>>>(n8, channels / 8, (const half8_t *)d_x,
                                                                               (const half8_t *)d_bias, (const half8_t *)d_skip,
                                                                               (half8_t *)d_out);
    SGO_HIP(hipGetLastError());
    return SGO_OK;
}

// ---- host-buffer API (drop-in for play.py / symmetry.py)
int sgo_game_init(int S, int n, int32_t *board17) {
    if (!size_ok(S) || n < 0 || !board17) { set_error("sgo_game_init: bad argument"); return SGO_ERR_ARG; }
    if (sgo_device_count() <= 0) { set_error("no HIP device"); return SGO_ERR_HIP; }
    memset(board17, 0, sizeof(int32_t) * (size_t)n * S * S * 17);
    for (size_t i = 0; i < (size_t)n * S * S; i++) board17[i * 17 + 16] = 1;
    return SGO_OK;
}

int sgo_make_play(int S, int n, int32_t *board17, const int32_t *xs, const int32_t *ys, const int32_t *colors,
                  int32_t *movers, int32_t *status) {
    if (!size_ok(S) || n < 0 || !board17 || !xs || !ys) { set_error("sgo_make_play: bad argument"); return SGO_ERR_ARG; }
    if (n == 0) return SGO_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t bsz = sizeof(int32_t) * (size_t)n * S * S * 17;
    const size_t psz = sizeof(uint32_t) * (size_t)n * sgo_packed_words(S);
    int r;
    if ((r = g_s[0].ensure(bsz))) return r;
    if ((r = g_s[1].ensure(psz))) return r;
    if ((r = g_s[2].ensure(sizeof(int32_t) * (size_t)n * 3))) return r;
    std::vector<int32_t> mv(n), st(n), col(n);
    std::vector<char> bad(n, 0);
    for (int i = 0; i < n; i++) {
        int x = xs[i], y = ys[i];
        if (y == S) mv[i] = S * S;                                  // pass (play.py:232)
        else if (x < 0 || x >= S || y < 0 || y >= S) { mv[i] = -1; bad[i] = 1; }
        else mv[i] = y * S + x;
        col[i] = colors ? colors[i] : 0;
    }
    int32_t *d_mv = (int32_t *)g_s[2].p, *d_col = d_mv + n, *d_st = d_mv + 2 * n;
    SGO_HIP(hipMemcpy(g_s[0].p, board17, bsz, hipMemcpyHostToDevice));
    SGO_HIP(hipMemcpy(d_mv, mv.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
    SGO_HIP(hipMemcpy(d_col, col.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice));
    if ((r = sgo_pack_dev(S, n, (const int32_t *)g_s[0].p, (uint32_t *)g_s[1].p, nullptr))) return r;
    if ((r = launch_advance_legal(S, n, (const uint32_t *)g_s[1].p, nullptr, d_mv, d_col, (uint32_t *)g_s[1].p, nullptr,
                                  nullptr, nullptr, d_st, nullptr)))
        return r;
    if ((r = sgo_unpack_dev(S, n, (const uint32_t *)g_s[1].p, (int32_t *)g_s[0].p, nullptr))) return r;
    SGO_HIP(hipMemcpy(st.data(), d_st, sizeof(int32_t) * n, hipMemcpyDeviceToHost));
    std::vector<int32_t> outb((size_t)n * S * S * 17);
    SGO_HIP(hipMemcpy(outb.data(), g_s[0].p, bsz, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        bool ok = st[i] == 1 || st[i] == -1;
        if (ok) memcpy(board17 + (size_t)i * S * S * 17, outb.data() + (size_t)i * S * S * 17, sizeof(int32_t) * S * S * 17);
        if (movers) movers[i] = ok ? st[i] : 0;
        if (status) status[i] = ok ? SGO_OK : st[i];
    }
    return SGO_OK;
}

int sgo_legal_moves(int S, int n, const int32_t *board17, uint8_t *mask) {
    if (!size_ok(S) || n < 0 || !board17 || !mask) { set_error("sgo_legal_moves: bad argument"); return SGO_ERR_ARG; }
    if (n == 0) return SGO_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    const int A = S * S + 1, NW = sgo_plane_words(S);
    const size_t bsz = sizeof(int32_t) * (size_t)n * S * S * 17;
    int r;
    if ((r = g_s[0].ensure(bsz))) return r;
    if ((r = g_s[1].ensure(sizeof(uint32_t) * (size_t)n * sgo_packed_words(S)))) return r;
    if ((r = g_s[2].ensure(sizeof(uint32_t) * (size_t)n * NW))) return r;
    if ((r = g_s[3].ensure((size_t)n * A))) return r;
    SGO_HIP(hipMemcpy(g_s[0].p, board17, bsz, hipMemcpyHostToDevice));
    if ((r = sgo_pack_dev(S, n, (const int32_t *)g_s[0].p, (uint32_t *)g_s[1].p, nullptr))) return r;
    if ((r = sgo_legal_dev(S, n, (const uint32_t *)g_s[1].p, nullptr, (uint32_t *)g_s[2].p, nullptr))) return r;
    k_legal_to_mask<<<dim3((n * A + 255) / 256), dim3(256), 0, nullptr>>>(A, NW, n, (const uint32_t *)g_s[2].p, (uint8_t *)g_s[3].p);
    SGO_HIP(hipGetLastError());
    SGO_HIP(hipMemcpy(mask, g_s[3].p, (size_t)n * A, hipMemcpyDeviceToHost));
    return SGO_OK;
}

int sgo_get_winner(int S, int n, const int32_t *board17, double komi, int32_t *winner, int32_t *black, double *white) {
    if (!size_ok(S) || n < 0 || !board17) { set_error("sgo_get_winner: bad argument"); return SGO_ERR_ARG; }
    if (n == 0) return SGO_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t bsz = sizeof(int32_t) * (size_t)n * S * S * 17;
    int r;
    if ((r = g_s[0].ensure(bsz))) return r;
    if ((r = g_s[1].ensure(sizeof(uint32_t) * (size_t)n * sgo_packed_words(S)))) return r;
    if ((r = g_s[2].ensure(sizeof(int32_t) * (size_t)n * 3))) return r;
    SGO_HIP(hipMemcpy(g_s[0].p, board17, bsz, hipMemcpyHostToDevice));
    if ((r = sgo_pack_dev(S, n, (const int32_t *)g_s[0].p, (uint32_t *)g_s[1].p, nullptr))) return r;
    if ((r = launch_score(S, n, (const uint32_t *)g_s[1].p, nullptr, komi, (int32_t *)g_s[2].p, nullptr))) return r;
    std::vector<int32_t> res((size_t)n * 3);
    SGO_HIP(hipMemcpy(res.data(), g_s[2].p, sizeof(int32_t) * n * 3, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; i++) {
        if (winner) winner[i] = res[3 * i];
        if (black) black[i] = res[3 * i + 1];
        if (white) white[i] = (double)res[3 * i + 2] + komi;
    }
    return SGO_OK;
}

int sgo_board_query(int S, int n, int mode, const int8_t *cells, const int32_t *xs, const int32_t *ys, const int32_t *colors,
                    uint8_t *member, uint8_t *liberty) {
    if (!size_ok(S) || n < 0 || (mode != 0 && mode != 1) || !cells || !colors || !member || !liberty || (mode == 0 && (!xs || !ys))) {
        set_error("sgo_board_query: bad argument");
        return SGO_ERR_ARG;
    }
    if (n == 0) return SGO_OK;
    for (int i = 0; i < n; i++) {
        if (mode == 0 && (xs[i] < 0 || xs[i] >= S || ys[i] < 0 || ys[i] >= S)) { set_error("sgo_board_query: seed outside the board"); return SGO_ERR_RANGE; }
        if (colors[i] < -1 || colors[i] > 1 || (mode == 1 && colors[i] == 0)) { set_error("sgo_board_query: bad colour"); return SGO_ERR_ARG; }
    }
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t csz = (size_t)n * S * S;
    int r;
    if ((r = g_s[0].ensure(csz))) return r;
    if ((r = g_s[1].ensure(2 * csz))) return r;
    if ((r = g_s[2].ensure(sizeof(int32_t) * (size_t)n * 3))) return r;
    int32_t *d_x = (int32_t *)g_s[2].p, *d_y = d_x + n, *d_c = d_x + 2 * n;
    SGO_HIP(hipMemcpy(g_s[0].p, cells, csz, hipMemcpyHostToDevice));
    if (mode == 0) {
        SGO_HIP(hipMemcpy(d_x, xs, sizeof(int32_t) * n, hipMemcpyHostToDevice));
        SGO_HIP(hipMemcpy(d_y, ys, sizeof(int32_t) * n, hipMemcpyHostToDevice));
    }
    SGO_HIP(hipMemcpy(d_c, colors, sizeof(int32_t) * n, hipMemcpyHostToDevice));
    uint8_t *d_m = (uint8_t *)g_s[1].p, *d_l = d_m + csz;
    SGO_DISPATCH(S, (k_board_query<kS><<<dim3((n + 63) / 64), dim3(64), 0, nullptr>>>(n, mode, (const int8_t *)g_s[0].p, d_x, d_y, d_c, d_m, d_l)));
    SGO_HIP(hipGetLastError());
    SGO_HIP(hipMemcpy(member, d_m, csz, hipMemcpyDeviceToHost));
    SGO_HIP(hipMemcpy(liberty, d_l, csz, hipMemcpyDeviceToHost));
    return SGO_OK;
}

int sgo_take_stones(int S, int n, int32_t *board17, const int32_t *xs, const int32_t *ys) {
    if (!size_ok(S) || n < 0 || !board17 || !xs || !ys) { set_error("sgo_take_stones: bad argument"); return SGO_ERR_ARG; }
    if (n == 0) return SGO_OK;
    for (int i = 0; i < n; i++)
        if (xs[i] < 0 || xs[i] >= S || ys[i] < 0 || ys[i] >= S) { set_error("sgo_take_stones: point outside the board"); return SGO_ERR_RANGE; }
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t bsz = sizeof(int32_t) * (size_t)n * S * S * 17;
    int r;
    if ((r = g_s[0].ensure(bsz))) return r;
    if ((r = g_s[2].ensure(sizeof(int32_t) * (size_t)n * 2))) return r;
    int32_t *d_x = (int32_t *)g_s[2].p, *d_y = d_x + n;
    SGO_HIP(hipMemcpy(g_s[0].p, board17, bsz, hipMemcpyHostToDevice));
    SGO_HIP(hipMemcpy(d_x, xs, sizeof(int32_t) * n, hipMemcpyHostToDevice));
    SGO_HIP(hipMemcpy(d_y, ys, sizeof(int32_t) * n, hipMemcpyHostToDevice));
    SGO_DISPATCH(S, (k_take_stones<kS><<<dim3((n + 63) / 64), dim3(64), 0, nullptr>>>(n, (int32_t *)g_s[0].p, d_x, d_y)));
    SGO_HIP(hipGetLastError());
    SGO_HIP(hipMemcpy(board17, g_s[0].p, bsz, hipMemcpyDeviceToHost));
    return SGO_OK;
}

int sgo_sym_apply(int S, int k, int n, const int32_t *in17, int32_t *out17) {
    if (!size_ok(S) || k < 0 || k > 7 || n < 0 || !in17 || !out17) { set_error("sgo_sym_apply: bad argument"); return SGO_ERR_ARG; }
    if (n == 0) return SGO_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    const size_t bsz = sizeof(int32_t) * (size_t)n * S * S * 17;
    int r;
    if ((r = g_s[0].ensure(bsz))) return r;
    if ((r = g_s[1].ensure(bsz))) return r;
    SGO_HIP(hipMemcpy(g_s[0].p, in17, bsz, hipMemcpyHostToDevice));
    k_sym_apply<<<dim3((n * S * S + 255) / 256), dim3(256), 0, nullptr>>>(S, k, n, (const int32_t *)g_s[0].p, (int32_t *)g_s[1].p);
    SGO_HIP(hipGetLastError());
    SGO_HIP(hipMemcpy(out17, g_s[1].p, bsz, hipMemcpyDeviceToHost));
    return SGO_OK;
}

int sgo_sym_invert_policy(int S, int k, int n, const float *in, float *out) {
    if (!size_ok(S) || k < 0 || k > 7 || n < 0 || !in || !out) { set_error("sgo_sym_invert_policy: bad argument"); return SGO_ERR_ARG; }
    if (n == 0) return SGO_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    const int A = S * S + 1;
    const size_t psz = sizeof(float) * (size_t)n * A;
    int r;
    if ((r = g_s[0].ensure(psz))) return r;
    if ((r = g_s[1].ensure(psz))) return r;
    if ((r = g_s[2].ensure(sizeof(int32_t) * A))) return r;
    std::vector<int32_t> lut(A);
    build_sym_lut(S, k, lut.data());
    SGO_HIP(hipMemcpy(g_s[2].p, lut.data(), sizeof(int32_t) * A, hipMemcpyHostToDevice));
    SGO_HIP(hipMemcpy(g_s[0].p, in, psz, hipMemcpyHostToDevice));
    k_sym_policy<<<dim3((n * A + 255) / 256), dim3(256), 0, nullptr>>>(A, n, (const int32_t *)g_s[2].p, (const float *)g_s[0].p, (float *)g_s[1].p);
    SGO_HIP(hipGetLastError());
    SGO_HIP(hipMemcpy(out, g_s[1].p, psz, hipMemcpyDeviceToHost));
    return SGO_OK;
}

}  // extern "C"
