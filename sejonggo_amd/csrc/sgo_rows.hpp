// sgo_rows.hpp -- the same rules as sgo_bits.hpp in ROW-PER-LANE form, for launches too small to fill the chip with one
// lane per position (the engine's per-step leaf list: games x energy positions).
//
// Execution model: a 32-lane HALF of a wavefront owns one position; lane y of the half holds row y of every stone set as
// one 32-bit mask (rows >= S are all-zero padding in every set, which also isolates the two halves of a wave from each
// other).  Left/right neighbours are shifts, up/down neighbours come from the adjacent lanes (DPP wave shifts, no
// LDS), set-wide predicates are half-wave ballots.  Flood fill is Jacobi over rows (all rows step at once, the carry
// trick of sgo_bits.hpp::row_fill finishes each row in one step), so a board costs a few hundred instructions of latency
// instead of the few thousand of the lane-per-position form -- at 1/32 of its throughput per lane, which is why the dense
// entry points keep the other form.  Results are bit-identical to sgo_bits.hpp (tests: every engine golden runs through
// this path; tests/test_gpu_rules.py compares the two forms directly).
//
// Restated reference functions (drsagitn/sejonggo): play.py:226-242 make_play, :182-217 take_stones, :159-180
// capture_group, :71-104 legal_moves.
#pragma once
#include "sgo_bits.hpp"

namespace sgo {
namespace rows {

SGO_DEV uint32_t row_above(uint32_t v) {   // the value of lane-1 (row y-1); lane 0 of the wave gets 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
}
SGO_DEV uint32_t row_below(uint32_t v) {   // the value of lane+1 (row y+1); lane 63 gets 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
}
// half-wave ballot of a per-lane predicate, delivered to every lane of that half
SGO_DEV uint32_t half_ballot(bool p, int half) {
    const unsigned long long b = __ballot(p);
    return half ? (uint32_t)(b >> 32) : (uint32_t)b;
}
SGO_DEV bool half_any(uint32_t v, int half) { return half_ballot(v != 0, half) != 0; }
// exactly one bit set over all rows of the half's set
SGO_DEV bool half_one_bit(uint32_t v, int half) {
    const uint32_t nz = half_ballot(v != 0, half), multi = half_ballot((v & (v - 1)) != 0, half);
    return __popc(nz) == 1 && multi == 0;
}

template <int S>
struct Board {
    int half, y;          // which half of the wave, row index inside the half
    uint32_t M;           // ROWMASK for rows < S, 0 for padding rows

    SGO_DEV uint32_t nbr4(uint32_t a) const { return (((a << 1) | (a >> 1)) | row_above(a) | row_below(a)) & M; }

    // grow x (subset of m) through m until stable; both halves of the wave iterate until neither changes
    SGO_DEV uint32_t flood(uint32_t x, uint32_t m) const {
        const uint32_t mrev = __brev(m);
        for (;;) {
            const uint32_t s = x | ((row_above(x) | row_below(x)) & m);
            const uint32_t t = row_fill<S>(s, m, mrev);
            const bool ch = t != x;
            x = t;
            if (!__any(ch)) break;
        }
        return x;
    }

    // one ply (advance_core of sgo_bits.hpp).  a in [0, N] (N = pass), the same for every lane of the half.
    // Returns 0 or -101 (occupied; own / opp untouched).
    SGO_DEV int advance(uint32_t &own, uint32_t &opp, int a) const {
        using G = Geo<S>;
        const int my = a / S, mx = a - my * S;
        const uint32_t pb = (a < G::N && y == my) ? (1u << mx) : 0u;
        if (half_any((own | opp) & pb, half)) return -101;
        const uint32_t np = nbr4(pb);
        own |= pb;
        uint32_t emp = ~(own | opp) & M;
        // opponent groups next to the stone that are left without an empty neighbour (skipped by the whole wave when neither
        // of its two positions has an opponent stone next to the new one: nothing could be captured)
        if (__any((np & opp) != 0)) {
            const uint32_t alive = flood(opp & nbr4(emp), opp);
            const uint32_t dead = opp & ~alive;
            const uint32_t cap = flood(dead & np, dead);
            opp &= ~cap;
            emp |= cap;
        }
        // the stone's own group: removed when it has no liberty left (suicide is executed, play.py:200-215); a stone with an
        // empty neighbour is alive, so the wave skips the fill when that holds for both of its positions
        const bool no_lib = half_any(pb, half) && !half_any(np & emp, half);   // the same in every lane of the half
        if (__any(no_lib)) {
            const uint32_t alive = flood(own & nbr4(emp), own);
            const bool hit = half_any(alive & pb, half);
            const uint32_t dead = own & ~alive;
            const uint32_t sg = flood(pb & dead, dead);
            own &= hit ? ~0u : ~sg;
        }
        return 0;
    }

    // legal set of the position (legal_core of sgo_bits.hpp): own = side to move, prev = its stones one ply ago
    SGO_DEV uint32_t legal(uint32_t own, uint32_t opp, uint32_t prev) const {
        const uint32_t emp = ~(own | opp) & M;
        const uint32_t t0 = nbr4(emp);
        const uint32_t e1 = emp & t0;      // empty with an empty neighbour: always legal
        const uint32_t c = emp & ~t0;      // empty, no empty neighbour: legal only if it captures
        uint32_t lg = e1;
        if (__any(c != 0)) {
            // opponent groups with a liberty outside c can never be captured by a c-move
            const uint32_t safe = flood(opp & nbr4(e1), opp);
            uint32_t r = opp & ~safe;
            // single-stone groups, all at once: exactly one empty neighbour => that point captures it
            {
                const uint32_t fr = nbr4(opp);
                const uint32_t U = row_above(emp), D = row_below(emp) & ((y < S - 1) ? ~0u : 0u);
                const uint32_t Lf = (emp << 1) & M, Rt = emp >> 1;
                const uint32_t x1 = U ^ D, a1 = U & D, x2 = Lf ^ Rt, a2 = Lf & Rt;
                const uint32_t one = (x1 ^ x2) & ~((a1 & x2) | (a2 & x1));
                const uint32_t single = r & ~fr;
                const uint32_t s1 = single & one;
                r &= ~single;
                lg |= nbr4(s1) & emp;
            }
            while (__any(r != 0)) {   // remaining groups: exactly one liberty => capturable there
                const uint32_t nz = half_ballot(r != 0, half);
                const int first = nz ? (__ffs((int)nz) - 1) : -1;
                const uint32_t seed = (y == first) ? (r & (0u - r)) : 0u;
                const uint32_t g = flood(seed, r);
                const uint32_t lib = nbr4(g) & emp;
                if (half_one_bit(lib, half)) lg |= lib;
                r &= ~g;
            }
        }
        // ko approximation (play.py:78-80): exactly one of the mover's stones vanished on the last ply
        const uint32_t gone = prev & ~own;
        if (half_one_bit(gone, half)) lg &= ~gone;
        return lg;
    }
};

// row y of a bit-plane stored as NW packed words (bit index = y*S + x); 0 for padding rows
template <int S>
SGO_DEV uint32_t load_row(const uint32_t *plane, int y) {
    using G = Geo<S>;
    if (y >= S) return 0u;
    const int bit = y * S, wi = bit >> 5, sh = bit & 31;
    uint32_t v = plane[wi] >> sh;
    if (sh + S > 32) v |= plane[(wi + 1 < G::NW) ? wi + 1 : wi] << ((32 - sh) & 31);
    return v & G::ROWMASK;
}
// word j of the packed plane, assembled in lane j (j < NW) of the half from the rows that overlap it
template <int S>
SGO_DEV uint32_t gather_word(uint32_t row, int half, int j) {
    constexpr int KMAX = 31 / S + 2;
    const int jj = (j < Geo<S>::NW) ? j : 0;
    const int y0 = (32 * jj) / S;
    uint32_t w = 0;
#pragma unroll
    for (int k = 0; k < KMAX; k++) {
        const int yy = y0 + k;
        const uint32_t r = (uint32_t)__shfl((int)row, half * 32 + (yy < 32 ? yy : 31), 64);   // padding rows are 0
        const int pos = yy * S - 32 * jj;
        if (yy < S) w |= (pos >= 0) ? ((pos < 32) ? (r << pos) : 0u) : (r >> (-pos));
    }
    return w;
}

// Full ply of one packed record by one half-wave: new pair + legal set of the new position + history move.
// in / out must not alias.  Every lane of the half returns the same status (0, -101 occupied, -102 bad move).
template <int S>
SGO_DEV int advance_record_rows(const uint32_t *in, uint32_t *out, int a, bool swap_first, uint32_t *legal_out, int half, int y) {
    using G = Geo<S>;
    if (a < 0 || a > G::N) return -102;
    Board<S> bd;
    bd.half = half;
    bd.y = y;
    bd.M = (y < S) ? G::ROWMASK : 0u;
    const bool mover_white = ((in[G::META_WORD] & G::META_BIT) != 0) != swap_first;
    const uint32_t black = load_row<S>(in, y), white = load_row<S>(in + G::NW, y);
    uint32_t own = mover_white ? white : black, opp = mover_white ? black : white;
    const uint32_t before_opp = opp;
    const int st = bd.advance(own, opp, a);
    if (st) return st;
    // new side to move = old opponent; its stones one ply ago = the opponent plane as loaded
    uint32_t lg = 0;
    if (legal_out) lg = bd.legal(opp, own, before_opp);
    const uint32_t nb = mover_white ? opp : own, nw = mover_white ? own : opp;
    uint32_t w0 = gather_word<S>(nb, half, y), w1 = gather_word<S>(nw, half, y), wl = legal_out ? gather_word<S>(lg, half, y) : 0u;
    if (y < G::NW) {
        if (y == G::META_WORD && !mover_white) w0 |= G::META_BIT;   // black moved => white to play
        out[y] = w0;
        out[G::NW + y] = w1;
        if (legal_out) {
            if (y == (G::N >> 5)) wl |= 1u << (G::N & 31);          // pass is always legal
            legal_out[y] = wl;
        }
    }
    // history: planes 0..13 of the parent become planes 2..15 (one contiguous run of 14*NW words)
    if constexpr (G::NW % 4 == 0) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 *s = reinterpret_cast<const u32x4 *>(in);
        u32x4 *d = reinterpret_cast<u32x4 *>(out + 2 * G::NW);
#pragma unroll
        for (int c0 = 0; c0 < 14 * G::NW / 4; c0 += 32) {
            const int c = c0 + y;
            if (c < 14 * G::NW / 4) d[c] = s[c];
        }
    } else {
#pragma unroll
        for (int c0 = 0; c0 < 14 * G::NW; c0 += 32) {
            const int c = c0 + y;
            if (c < 14 * G::NW) out[2 * G::NW + c] = in[c];
        }
    }
    return 0;
}

}  // namespace rows

// What the fused board_advance + nn_input_pack kernel keeps from the ply: word `lane` of the child's planes 0 / 1 (valid in
// lanes 0..NW-1 of the half) and who is to move in the child.
template <int S>
struct Board_rows_result {
    uint32_t w0, w1;      // word `lane` of the child's planes 0 / 1
    uint32_t rb, rw;      // row `lane` of the child's black / white stones
    bool child_white;
};

// advance_record_rows of a LEGAL move (the engine's leaves: the move comes from the legal set), keeping the new pair in
// registers.  out / legal_out may be null for an idle half (nothing is written; the wave stays converged).
template <int S>
SGO_DEV void rows_advance_keep(const uint32_t *in, uint32_t *out, int a, uint32_t *legal_out, int half, int y,
                               Board_rows_result<S> &res) {
    using G = Geo<S>;
    rows::Board<S> bd;
    bd.half = half;
    bd.y = y;
    bd.M = (y < S) ? G::ROWMASK : 0u;
    const bool mover_white = (in[G::META_WORD] & G::META_BIT) != 0;
    const uint32_t black = rows::load_row<S>(in, y), white = rows::load_row<S>(in + G::NW, y);
    uint32_t own = mover_white ? white : black, opp = mover_white ? black : white;
    const uint32_t before_opp = opp;
    (void)bd.advance(own, opp, a);
    const uint32_t lg = bd.legal(opp, own, before_opp);
    const uint32_t nb = mover_white ? opp : own, nw = mover_white ? own : opp;
    uint32_t w0 = rows::gather_word<S>(nb, half, y), w1 = rows::gather_word<S>(nw, half, y), wl = rows::gather_word<S>(lg, half, y);
    if (y == G::META_WORD && !mover_white) w0 |= G::META_BIT;   // black moved => white to play
    if (y == (G::N >> 5)) wl |= 1u << (G::N & 31);              // pass is always legal
    res.w0 = w0;
    res.w1 = w1;
    res.rb = nb;
    res.rw = nw;
    res.child_white = !mover_white;
    if (!out) return;
    if (y < G::NW) {
        out[y] = w0;
        out[G::NW + y] = w1;
        legal_out[y] = wl;
    }
    if constexpr (G::NW % 4 == 0) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 *s = reinterpret_cast<const u32x4 *>(in);
        u32x4 *d = reinterpret_cast<u32x4 *>(out + 2 * G::NW);
#pragma unroll
        for (int c0 = 0; c0 < 14 * G::NW / 4; c0 += 32) {
            const int c = c0 + y;
            if (c < 14 * G::NW / 4) d[c] = s[c];
        }
    } else {
#pragma unroll
        for (int c0 = 0; c0 < 14 * G::NW; c0 += 32) {
            const int c = c0 + y;
            if (c < 14 * G::NW) out[2 * G::NW + c] = in[c];
        }
    }
}
}  // namespace sgo
