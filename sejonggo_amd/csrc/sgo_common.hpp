// sgo_common.hpp -- host-side helpers shared by the translation units of libsgo_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/sgo.h"

namespace sgo {

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define SGO_HIP(call)                                                            \
    do {                                                                         \
        hipError_t _e = (call);                                                  \
        if (_e != hipSuccess) return ::sgo::hip_fail(_e, #call, __FILE__, __LINE__); \
    } while (0)

inline bool size_ok(int S) { return S == 5 || S == 7 || S == 9 || S == 13 || S == 19; }

// dispatch a template on the runtime board size
#define SGO_DISPATCH(S, ...)                                    \
    switch (S) {                                                \
    case 5: { constexpr int kS = 5; __VA_ARGS__; } break;              \
    case 7: { constexpr int kS = 7; __VA_ARGS__; } break;              \
    case 9: { constexpr int kS = 9; __VA_ARGS__; } break;              \
    case 13: { constexpr int kS = 13; __VA_ARGS__; } break;            \
    case 19: { constexpr int kS = 19; __VA_ARGS__; } break;            \
    default: ::sgo::set_error("unsupported board size"); return SGO_ERR_ARG; \
    }

// symmetry.py:12-42 -- SWAP tables by float rotation + round, as the reference builds them
void build_sym_lut(int S, int k, int32_t *lut);

// kernels launched from more than one translation unit
int launch_advance_legal(int S, int n, const uint32_t *d_in, const int32_t *d_in_idx, const int32_t *d_moves,
                         const int32_t *d_colors, uint32_t *d_out, const int32_t *d_out_idx, uint32_t *d_legal,
                         const int32_t *d_legal_idx, int32_t *d_status, hipStream_t st);
// split form (history stream + register kernel); in/out records must not alias.  d_n (device int) overrides n_max.
int launch_advance_split(int S, int n_max, const int *d_n, const uint32_t *d_in, const int32_t *d_in_idx,
                         const int32_t *d_moves, const int32_t *d_colors, uint32_t *d_out, const int32_t *d_out_idx,
                         uint32_t *d_legal, const int32_t *d_legal_idx, int32_t *d_status, hipStream_t st);
int launch_nn_pack(int S, int n, const uint32_t *d_packed, const int32_t *d_idx, int k, int layout, int dtype,
                   void *d_out, hipStream_t st);
int launch_score(int S, int n, const uint32_t *d_packed, const int32_t *d_idx, double komi, int32_t *d_result,
                 hipStream_t st);

}  // namespace sgo
