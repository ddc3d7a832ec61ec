// sgo_bits.hpp -- register-resident bitboard Go rules for CDNA4 (gfx950).
//
// Execution model: ONE LANE PER POSITION.  A lane keeps each colour as S row-bitmasks (bit x of row
// y) in VGPRs; every loop over rows is fully unrolled so all indexing is static (no scratch).  Group
// and liberty logic is whole-board bit-parallel flood fill: left/right neighbours are shifts, up/down
// neighbours are the adjacent registers, so there is no cross-lane traffic and no LDS in the rules
// core at all.  A wavefront therefore advances 64 positions at once and its only divergence is the
// flood-fill trip count (max over the 64 lanes).
//
// What is restated here (reference = drsagitn/sejonggo, file:line):
//   advance_core  play.py:226-242 make_play, :182-217 take_stones, :159-180 capture_group
//   legal_core    play.py:71-104 legal_moves (incl. the "exactly one stone vanished" ko rule :78-80
//                 and the "no empty neighbour and no capture => illegal" rule :93-100)
//   score_core    play.py:244-292 color_board/_get_points/get_winner
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sgo {

template <int S>
struct Geo {
    static constexpr int N = S * S;
    static constexpr int A = N + 1;
    static constexpr int NW = (N + 31) / 32;   // words per bit-plane
    static constexpr int MW = (A + 31) / 32;   // words per legal bitset (incl. pass bit)
    static constexpr int APAD = NW * 32;       // child slots per tree block
    static constexpr int RW = 16 * NW;         // words per packed position record (19x19: 768 B = six 128-B lines)
    // Record layout: plane 2k = BLACK stones k plies ago, plane 2k+1 = WHITE stones k plies ago (absolute colours,
    // so that one ply is "pairs 0..6 move to pairs 1..7" -- a contiguous copy -- and a colour override needs no plane
    // swap).  The to-play bit lives in the spare top bit of plane 0's last word (S*S < 32*NW for every supported size).
    static constexpr int META_WORD = NW - 1;
    static constexpr uint32_t META_BIT = 0x80000000u;
    static_assert(S * S < 32 * NW, "need a spare bit in the last plane word");
    static constexpr uint32_t ROWMASK = (1u << S) - 1u;
    static_assert(MW == NW, "pass bit must fit in the last plane word");
    static_assert(S >= 2 && S <= 19, "board size");
};

#define SGO_DEV __device__ __forceinline__

// ---- packed words <-> row registers --------------------------------------------------------------
template <int S>
SGO_DEV void unpack_rows(const uint32_t (&w)[Geo<S>::NW], uint32_t (&r)[S]) {
#pragma unroll
    for (int y = 0; y < S; y++) {
        const int bit = y * S, wi = bit >> 5, sh = bit & 31;
        uint32_t v = w[wi] >> sh;
        if (sh + S > 32) v |= w[(wi + 1 < Geo<S>::NW) ? wi + 1 : wi] << ((32 - sh) & 31);
        r[y] = v & Geo<S>::ROWMASK;
    }
}
template <int S>
SGO_DEV void pack_rows(const uint32_t (&r)[S], uint32_t (&w)[Geo<S>::NW]) {
#pragma unroll
    for (int i = 0; i < Geo<S>::NW; i++) w[i] = 0;
#pragma unroll
    for (int y = 0; y < S; y++) {
        const int bit = y * S, wi = bit >> 5, sh = bit & 31;
        w[wi] |= r[y] << sh;
        if (sh + S > 32) w[(wi + 1 < Geo<S>::NW) ? wi + 1 : wi] |= r[y] >> ((32 - sh) & 31);
    }
}

// ---- neighbourhood ---------------------------------------------------------------------------------
// 4-neighbourhood of a set (the set itself NOT included)
template <int S>
SGO_DEV void nbr4(const uint32_t (&a)[S], uint32_t (&o)[S]) {
#pragma unroll
    for (int y = 0; y < S; y++) {
        uint32_t v = ((a[y] << 1) | (a[y] >> 1)) & Geo<S>::ROWMASK;
        if (y > 0) v |= a[y - 1];
        if (y < S - 1) v |= a[y + 1];
        o[y] = v;
    }
}
template <int S>
SGO_DEV uint32_t any_rows(const uint32_t (&a)[S]) {
    uint32_t v = 0;
#pragma unroll
    for (int y = 0; y < S; y++) v |= a[y];
    return v;
}
template <int S>
SGO_DEV int popc_rows(const uint32_t (&a)[S]) {
    int c = 0;
#pragma unroll
    for (int y = 0; y < S; y++) c += __popc(a[y]);
    return c;
}

// Flood fill: grow x (must start inside m) through m until stable.  Each iteration is one downward
// and one upward Gauss-Seidel sweep (a row sees its already-updated neighbour row), and inside a row
// the fill runs to the ends of the run in one step with the carry trick:
//   up-fill   (towards higher bits): ((m + s) ^ m) & m | s   -- the add ripples through the 1-run
//   down-fill: same on the bit-reversed row.
template <int S>
SGO_DEV uint32_t row_fill(uint32_t s, uint32_t m, uint32_t mrev) {
    // s subset of m, both within ROWMASK; bit S of (m + s) may be set: masked out by "& m".
    uint32_t up = (((m + s) ^ m) & m) | s;
    uint32_t sr = __brev(up);                       // reversed seeds (include the up-fill result)
    uint32_t dn = (((mrev + sr) ^ mrev) & mrev) | sr;
    return __brev(dn);
}
template <int S>
SGO_DEV void flood(uint32_t (&x)[S], const uint32_t (&m)[S]) {
    // nothing can grow? (common: every stone of the set already touches a seed)
    uint32_t grow = 0;
#pragma unroll
    for (int y = 0; y < S; y++) {
        uint32_t v = ((x[y] << 1) | (x[y] >> 1));
        if (y > 0) v |= x[y - 1];
        if (y < S - 1) v |= x[y + 1];
        grow |= v & m[y] & ~x[y];
    }
    if (!grow) return;
    uint32_t mrev[S];
#pragma unroll
    for (int y = 0; y < S; y++) mrev[y] = __brev(m[y]);
    uint32_t diff;
    do {
        // downward sweep: row y sees the already-updated row y-1
#pragma unroll
        for (int y = 0; y < S; y++) {
            uint32_t s = x[y];
            if (y > 0) s |= x[y - 1] & m[y];
            x[y] = row_fill<S>(s, m[y], mrev[y]);
        }
        // upward sweep; if it changes nothing the set is closed in all four directions
        diff = 0;
#pragma unroll
        for (int y = S - 2; y >= 0; y--) {
            uint32_t s = x[y] | (x[y + 1] & m[y]);
            uint32_t t = row_fill<S>(s, m[y], mrev[y]);
            diff |= t ^ x[y];
            x[y] = t;
        }
    } while (diff != 0);
}

// ---- one ply ---------------------------------------------------------------------------------------
// own = to-play side, opp = other side (rows).  Places `a` (a == N: pass) for own, removes captured opp
// groups adjacent to the stone, then the stone's own group if it is left without liberties (suicide is
// executed, play.py:200-215).  Returns 0, or SGO_ERR_OCCUPIED(-101) and leaves own/opp untouched.
template <int S>
SGO_DEV int advance_core(uint32_t (&own)[S], uint32_t (&opp)[S], int a) {
    constexpr uint32_t M = Geo<S>::ROWMASK;
    if (a >= Geo<S>::N) return 0;
    const int my = a / S, mx = a - my * S;
    const uint32_t bit = 1u << mx;
    uint32_t pb[S], np[S];  // the stone, and its 4-neighbourhood
    uint32_t occ_at = 0;
#pragma unroll
    for (int y = 0; y < S; y++) {
        pb[y] = (y == my) ? bit : 0u;
        occ_at |= (own[y] | opp[y]) & pb[y];
    }
    if (occ_at) return -101;
    nbr4<S>(pb, np);
    uint32_t emp[S], t[S];
#pragma unroll
    for (int y = 0; y < S; y++) {
        own[y] |= pb[y];
        emp[y] = ~(own[y] | opp[y]) & M;
    }
    // opponent groups that still touch an empty point
    uint32_t touch = 0;
#pragma unroll
    for (int y = 0; y < S; y++) touch |= np[y] & opp[y];
    if (touch) {
        uint32_t alive[S];
        nbr4<S>(emp, t);
#pragma unroll
        for (int y = 0; y < S; y++) alive[y] = opp[y] & t[y];
        flood<S>(alive, opp);
        uint32_t dead[S], cap[S], anyd = 0;
#pragma unroll
        for (int y = 0; y < S; y++) {
            dead[y] = opp[y] & ~alive[y];
            cap[y] = dead[y] & np[y];   // only groups adjacent to the new stone are examined
            anyd |= cap[y];
        }
        if (anyd) {
            flood<S>(cap, dead);
#pragma unroll
            for (int y = 0; y < S; y++) {
                opp[y] &= ~cap[y];
                emp[y] |= cap[y];
            }
        }
    }
    // own group of the new stone
    uint32_t lib = 0;
#pragma unroll
    for (int y = 0; y < S; y++) lib |= np[y] & emp[y];
    if (!lib) {
        uint32_t alive[S];
        nbr4<S>(emp, t);
#pragma unroll
        for (int y = 0; y < S; y++) alive[y] = own[y] & t[y];
        flood<S>(alive, own);
        uint32_t hit = 0;
#pragma unroll
        for (int y = 0; y < S; y++) hit |= alive[y] & pb[y];
        if (!hit) {
            uint32_t dead[S], sg[S];
#pragma unroll
            for (int y = 0; y < S; y++) {
                dead[y] = own[y] & ~alive[y];
                sg[y] = pb[y];
            }
            flood<S>(sg, dead);
#pragma unroll
            for (int y = 0; y < S; y++) own[y] &= ~sg[y];
        }
    }
    return 0;
}

// take_stones (play.py:182-217) as a stand-alone step on a position that may or may not hold the stone at `a`:
// pass 1 removes the opponent groups adjacent to `a` that have no liberty; pass 2 removes the own groups among the
// four neighbours AND `a` itself that have no liberty.  With the stone present this equals advance_core minus the
// placement; without it the (up to four) own neighbour groups are examined one by one, as the reference does.
template <int S>
SGO_DEV void take_core(uint32_t (&own)[S], uint32_t (&opp)[S], int a) {
    constexpr uint32_t M = Geo<S>::ROWMASK;
    const int my = a / S, mx = a - my * S;
    uint32_t pb[S], np[S], emp[S], t[S], alive[S], dead[S], hit[S];
#pragma unroll
    for (int y = 0; y < S; y++) pb[y] = (y == my) ? (1u << mx) : 0u;
    nbr4<S>(pb, np);
#pragma unroll
    for (int y = 0; y < S; y++) emp[y] = ~(own[y] | opp[y]) & M;
    nbr4<S>(emp, t);
#pragma unroll
    for (int y = 0; y < S; y++) alive[y] = opp[y] & t[y];
    flood<S>(alive, opp);
    uint32_t any = 0;
#pragma unroll
    for (int y = 0; y < S; y++) {
        dead[y] = opp[y] & ~alive[y];
        hit[y] = dead[y] & np[y];
        any |= hit[y];
    }
    if (any) {
        flood<S>(hit, dead);
#pragma unroll
        for (int y = 0; y < S; y++) {
            opp[y] &= ~hit[y];
            emp[y] |= hit[y];
        }
    }
    nbr4<S>(emp, t);
#pragma unroll
    for (int y = 0; y < S; y++) alive[y] = own[y] & t[y];
    flood<S>(alive, own);
    any = 0;
#pragma unroll
    for (int y = 0; y < S; y++) {
        dead[y] = own[y] & ~alive[y];
        hit[y] = dead[y] & (np[y] | pb[y]);
        any |= hit[y];
    }
    if (any) {
        flood<S>(hit, dead);
#pragma unroll
        for (int y = 0; y < S; y++) own[y] &= ~hit[y];
    }
}

// ---- legal-move set --------------------------------------------------------------------------------
// own = to-play stones (plane 0), opp = plane 1, prev = plane 2 (to-play side's stones one ply ago).
// legal[] receives 1-bits for LEGAL board points (pass is handled by the caller).
template <int S>
SGO_DEV void legal_core(const uint32_t (&own)[S], const uint32_t (&opp)[S], const uint32_t (&prev)[S],
                        uint32_t (&legal)[S]) {
    constexpr uint32_t M = Geo<S>::ROWMASK;
    uint32_t emp[S], e1[S], c[S], t[S];
#pragma unroll
    for (int y = 0; y < S; y++) emp[y] = ~(own[y] | opp[y]) & M;
    nbr4<S>(emp, t);
    uint32_t anyc = 0;
#pragma unroll
    for (int y = 0; y < S; y++) {
        e1[y] = emp[y] & t[y];     // empty with an empty neighbour: always legal
        c[y] = emp[y] & ~t[y];     // empty, no empty neighbour: legal only if it captures
        legal[y] = e1[y];
        anyc |= c[y];
    }
    if (anyc) {
        // opponent groups with a liberty outside c can never be captured by a c-move
        uint32_t safe[S], r[S];
        nbr4<S>(e1, t);
        uint32_t anyr = 0;
#pragma unroll
        for (int y = 0; y < S; y++) safe[y] = opp[y] & t[y];
        flood<S>(safe, opp);
#pragma unroll
        for (int y = 0; y < S; y++) {
            r[y] = opp[y] & ~safe[y];
            anyr |= r[y];
        }
        // single-stone groups, all at once: a lone stone's liberties are its empty neighbours; exactly one of the
        // four => that point captures it.  (Bit-sliced "exactly one of U, D, L, R".)
        {
            uint32_t fr[S];
            nbr4<S>(opp, fr);
            uint32_t anys = 0;
#pragma unroll
            for (int y = 0; y < S; y++) anys |= r[y] & ~fr[y];
            if (anys) {
                uint32_t s1[S];
#pragma unroll
                for (int y = 0; y < S; y++) {
                    const uint32_t U = (y > 0) ? emp[y - 1] : 0u, D = (y < S - 1) ? emp[y + 1] : 0u;
                    const uint32_t Lf = (emp[y] << 1) & M, Rt = emp[y] >> 1;
                    const uint32_t x1 = U ^ D, a1 = U & D, x2 = Lf ^ Rt, a2 = Lf & Rt;
                    const uint32_t one = (x1 ^ x2) & ~((a1 & x2) | (a2 & x1));
                    const uint32_t single = r[y] & ~fr[y];
                    s1[y] = single & one;
                    r[y] &= ~single;
                }
                nbr4<S>(s1, fr);
                anyr = 0;
#pragma unroll
                for (int y = 0; y < S; y++) {
                    legal[y] |= fr[y] & emp[y];
                    anyr |= r[y];
                }
            }
        }
        while (anyr) {  // remaining groups: every liberty lies in c; exactly one liberty => capturable there
            uint32_t g[S];
            bool found = false;
#pragma unroll
            for (int y = 0; y < S; y++) {
                uint32_t low = r[y] & (0u - r[y]);
                g[y] = found ? 0u : low;
                found = found || (r[y] != 0);
            }
            flood<S>(g, r);
            nbr4<S>(g, t);
            int nl = 0;
#pragma unroll
            for (int y = 0; y < S; y++) {
                t[y] &= emp[y];
                nl += __popc(t[y]);
            }
            anyr = 0;
#pragma unroll
            for (int y = 0; y < S; y++) {
                if (nl == 1) legal[y] |= t[y];
                r[y] &= ~g[y];
                anyr |= r[y];
            }
        }
    }
    // ko approximation (play.py:78-80): exactly one of to-play's stones vanished on the last ply
    int kc = 0;
#pragma unroll
    for (int y = 0; y < S; y++) kc += __popc(prev[y] & ~own[y]);
    if (kc == 1) {
#pragma unroll
        for (int y = 0; y < S; y++) legal[y] &= ~(prev[y] & ~own[y]);
    }
}

// ---- area score ------------------------------------------------------------------------------------
// black/white = absolute colours.  Returns black points and white points (without komi).
template <int S>
SGO_DEV void score_core(const uint32_t (&black)[S], const uint32_t (&white)[S], int &bp, int &wp) {
    constexpr uint32_t M = Geo<S>::ROWMASK;
    uint32_t emp[S], rb[S], rw[S], t[S];
#pragma unroll
    for (int y = 0; y < S; y++) emp[y] = ~(black[y] | white[y]) & M;
    nbr4<S>(black, t);
#pragma unroll
    for (int y = 0; y < S; y++) rb[y] = t[y] & emp[y];
    flood<S>(rb, emp);
    nbr4<S>(white, t);
#pragma unroll
    for (int y = 0; y < S; y++) rw[y] = t[y] & emp[y];
    flood<S>(rw, emp);
    bp = 0;
    wp = 0;
#pragma unroll
    for (int y = 0; y < S; y++) {
        bp += __popc(black[y]) + __popc(rb[y] & ~rw[y]);
        wp += __popc(white[y]) + __popc(rw[y] & ~rb[y]);
    }
}

// ---- record access ---------------------------------------------------------------------------------
template <int S>
SGO_DEV void load_plane(const uint32_t *rec, int plane, uint32_t (&w)[Geo<S>::NW]) {
    const uint32_t *p = rec + plane * Geo<S>::NW;
    if constexpr (Geo<S>::NW % 4 == 0) {
#pragma unroll
        for (int i = 0; i < Geo<S>::NW / 4; i++) {
            uint4 v = reinterpret_cast<const uint4 *>(p)[i];
            w[4 * i] = v.x; w[4 * i + 1] = v.y; w[4 * i + 2] = v.z; w[4 * i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < Geo<S>::NW; i++) w[i] = p[i];
    }
}
template <int S>
SGO_DEV void store_plane(uint32_t *rec, int plane, const uint32_t (&w)[Geo<S>::NW]) {
    uint32_t *p = rec + plane * Geo<S>::NW;
    if constexpr (Geo<S>::NW % 4 == 0) {
#pragma unroll
        for (int i = 0; i < Geo<S>::NW / 4; i++)
            reinterpret_cast<uint4 *>(p)[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
    } else {
#pragma unroll
        for (int i = 0; i < Geo<S>::NW; i++) p[i] = w[i];
    }
}

template <int S>
SGO_DEV bool white_to_play(const uint32_t *rec) { return (rec[Geo<S>::META_WORD] & Geo<S>::META_BIT) != 0; }

// Core of one ply on the current pair.  p0/p1 are filled with planes 0,1 (black, white) of `in`, and on return hold
// the new pair (meta bit set for the new side to move); optionally the legal set of the new position is produced.
template <int S>
SGO_DEV int advance_pair(const uint32_t *in, int a, bool swap_first, uint32_t (&p0)[Geo<S>::NW], uint32_t (&p1)[Geo<S>::NW],
                         uint32_t *legal_out) {
    using G = Geo<S>;
    if (a < 0 || a > G::N) return -102;
    load_plane<S>(in, 0, p0);
    load_plane<S>(in, 1, p1);
    // make_play's `color != to-play` branch (play.py:227-228) only changes who moves: colours are absolute here
    const bool mover_white = ((p0[G::META_WORD] & G::META_BIT) != 0) != swap_first;
    // mover / opponent are picked at the packed-word level (NW swaps, no extra row arrays stay live)
#pragma unroll
    for (int i = 0; i < G::NW; i++) {
        const uint32_t b = p0[i], w = p1[i];
        p0[i] = mover_white ? w : b;
        p1[i] = mover_white ? b : w;
    }
    uint32_t own[S], opp[S];
    unpack_rows<S>(p0, own);
    unpack_rows<S>(p1, opp);
    int st = advance_core<S>(own, opp, a);
    if (st) return st;
    if (legal_out) {
        // new side to move = old opponent; its stones one ply ago = the opponent plane as loaded (still in p1)
        uint32_t before_opp[S], legal[S];
        unpack_rows<S>(p1, before_opp);
        legal_core<S>(opp, own, before_opp, legal);
        uint32_t lw[G::NW];
        pack_rows<S>(legal, lw);
        lw[G::N >> 5] |= 1u << (G::N & 31);               // pass is always legal
#pragma unroll
        for (int i = 0; i < G::NW; i++) legal_out[i] = lw[i];
    }
    pack_rows<S>(own, p0);
    pack_rows<S>(opp, p1);
#pragma unroll
    for (int i = 0; i < G::NW; i++) {
        const uint32_t o = p0[i], p = p1[i];
        p0[i] = mover_white ? p : o;   // black
        p1[i] = mover_white ? o : p;   // white
    }
    if (!mover_white) p0[G::META_WORD] |= G::META_BIT;   // black moved => white to play
    return 0;
}

// Full ply on a packed record, fused with the legal set of the resulting position.  in / out may be the same
// record: the history pairs move from the highest plane down.  Returns 0 or a negative status (record untouched).
template <int S>
SGO_DEV int advance_record(const uint32_t *in, uint32_t *out, int a, bool swap_first, uint32_t *legal_out) {
    using G = Geo<S>;
    uint32_t n0[G::NW], n1[G::NW];
    int st = advance_pair<S>(in, a, swap_first, n0, n1, legal_out);
    if (st) return st;
#pragma unroll 1
    for (int p = 15; p >= 2; p--) {
        uint32_t h[G::NW];
        load_plane<S>(in, p - 2, h);
        store_plane<S>(out, p, h);
    }
    store_plane<S>(out, 0, n0);
    store_plane<S>(out, 1, n1);
    return 0;
}

// The same ply WITHOUT the history planes: writes planes 0,1 of `out` and the legal set.  Planes 2..15 are moved by
// the streaming kernel k_history_shift (only valid when in and out do not alias).
template <int S>
SGO_DEV int advance_planes(const uint32_t *in, uint32_t *out, int a, bool swap_first, uint32_t *legal_out) {
    using G = Geo<S>;
    uint32_t n0[G::NW], n1[G::NW];
    int st = advance_pair<S>(in, a, swap_first, n0, n1, legal_out);
    if (st) return st;
    store_plane<S>(out, 0, n0);
    store_plane<S>(out, 1, n1);
    return 0;
}

template <int S>
SGO_DEV void legal_record(const uint32_t *rec, uint32_t *legal_out) {
    using G = Geo<S>;
    uint32_t w[G::NW], own[S], opp[S], prev[S], legal[S], lw[G::NW];
    const bool wtp = white_to_play<S>(rec);
    load_plane<S>(rec, wtp ? 1 : 0, w);
    unpack_rows<S>(w, own);
    load_plane<S>(rec, wtp ? 0 : 1, w);
    unpack_rows<S>(w, opp);
    load_plane<S>(rec, wtp ? 3 : 2, w);     // the side to move, one ply ago
    unpack_rows<S>(w, prev);
    legal_core<S>(own, opp, prev, legal);
    pack_rows<S>(legal, lw);
    lw[G::N >> 5] |= 1u << (G::N & 31);
#pragma unroll
    for (int i = 0; i < G::NW; i++) legal_out[i] = lw[i];
}

template <int S>
SGO_DEV void score_record(const uint32_t *rec, int &bp, int &wp) {
    using G = Geo<S>;
    uint32_t w[G::NW], bl[S], wh[S];
    load_plane<S>(rec, 0, w);
    unpack_rows<S>(w, bl);
    load_plane<S>(rec, 1, w);
    unpack_rows<S>(w, wh);
    score_core<S>(bl, wh, bp, wp);
}

}  // namespace sgo
