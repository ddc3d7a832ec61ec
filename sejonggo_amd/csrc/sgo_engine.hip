// sgo_engine.hip -- device-resident self-play engine: virtual-loss PUCT search + game loop for many
// concurrent games on one MI355X.  Second half of the C ABI in include/sgo.h.
//
// What it replaces (reference = drsagitn/sejonggo, file:line):
//   play.py:308-323 top_one_with_virtual_loss, tree_util.py:4-24 find_best_leaf_virtual_loss,
//   play.py:376-421 new_tree/new_subtree, simulation_workers.py:42-54 basic_tasks2,
//   nomodel_self_play.py:40-56 back_propagation, :59-82 async_simulate2, :114-140 select_play,
//   :142-271 play_game_async, and the request side of predicting_queue_worker.py:40-102.
//
// Execution model
//   * ONE WAVEFRONT PER GAME runs the game's state machine (k_search): the descent does a 64-lane
//     argmax over the <=362 child slots of a node (6 slots per lane at 19x19, butterfly reduce), busy
//     flags and back-off exactly as the reference; expansion and garbage collection use wave ballots +
//     prefix popcounts.  Games never talk to each other, so there is no inter-workgroup hand-off.
//   * Per engine step: k_search (consume evaluations -> back-propagate -> select next leaves / play a
//     move) -> k_compact (prefix sums over games: dense evaluation list + leaf list) -> board_advance
//     (k_board_advance, one LANE per leaf + history-stream blocks, sized for the worst case and guarded by the
//     device-side leaf count so no host round trip sits in between) -> [host runs the network] -> next step.
//   * Tree storage: every game owns `cap` fixed-size BLOCKS.  A block = one expanded node: its packed
//     position, legal bitset, and APAD child slots in struct-of-arrays form (P, N, W, Q, child block,
//     busy) so that the lanes of the selecting wave read consecutive slots (coalesced).  Child slot i
//     is only ever touched by lane (i & 63) of the game's wave.  Blocks are recycled by a
//     mark-and-rebuild pass when the tree is re-rooted after a move.
//
// Float regime (must match oracle/sgo_oracle.c, i.e. the reference under numpy>=2): W/Q/score in
// float32; at a root whose priors were mixed with Dirichlet noise priors and score are float64.
#include <math.h>
#include <string.h>
#include <algorithm>
#include <vector>

#include "sgo_bits.hpp"
#include "sgo_common.hpp"

namespace sgo {

enum { PH_IDLE = 0, PH_WAIT_ROOT = 1, PH_SEARCH = 2, PH_DONE = 3 };
#define MAXE 64

struct GameState {
    int32_t phase, root_blk, move_n, player;
    int32_t temperature, skipped_last, has_value, end_reason;
    float value, last_value, resign;
    int32_t has_resign;
    int32_t error, rounds_left, e_left, pre_bp;
    int32_t need_bp, original_player, fifo_head, fifo_tail;
    int32_t free_top, root_f64, root_count, halt_at;
    float root_value, root_mean;
    int32_t i_uniform, n_uniform;
    int32_t noise_used, game_seq, n_req, req_kind;
    int32_t eval_base, root_requested, winner, black;
    int32_t list_base, pad2_;              // k_compact: base of the game's requests in the leaf list (or among the root requests)
    double white;
    int32_t n_moves, last_player;
    int64_t n_predict, none_events;
    // two-model (evaluation) games, nomodel_self_play.py:203-218: the side NOT to move keeps its own tree
    int32_t cur_model, first_model;       // 0 = model1, 1 = model2: who searches now / who moved first (plays black)
    int32_t other_root, other_count;      // the other player's tree: root block (-1: none) and root statistics
    float other_value, other_mean;
    float resign2;
    int32_t has_resign2;
    int32_t min_free;                     // fewest free local ids the game ever had (high-water mark = L - min_free)
    int32_t ovf_hi;                       // overflow local ids [cap, cap + ovf_hi) have been backed at some time in this game
};

struct Counters {
    int32_t rec_count;
    int32_t pad;
    unsigned long long total_moves, total_evals, none_events;
    unsigned long long dbg[8];   // diagnostic build (-DSGO_KSEARCH_PROFILE): cycles per phase of k_search, summed over games
};

struct DevStatus {  // written by k_compact, copied to the host once per step
    int32_t n_eval, n_leaf, n_records, n_active, n_done, error, error_game, n_root;
    unsigned long long total_moves, total_evals, none_events;
};

struct Ctx {
    sgo_config cfg;
    int S, A, APAD, NW, RW, G, E, cap;
    // Block ids of a game are LOCAL: [0, cap) live in the game's private region (physical block g * cap + id), [cap, L) are
    // overflow ids, backed on demand by blocks of a pool SHARED by all games of the context (physical block G * cap + ovfMap).
    int ovf_cap, L;
    long pool_blocks;
    int max_moves;       // effective num_moves
    int rec_cap;
    // device arrays
    GameState *gs;
    uint32_t *pos;        // [G*cap][RW]
    uint32_t *legal;      // [G*cap][NW]
    float *cP, *cW, *cQ;  // [G*cap][APAD]
    int32_t *cN, *cB;     // counts, child block (local index, -1 = not expanded)
    uint8_t *cBusy;
    int32_t *bParent;     // [G*cap] local parent block (-1 root)
    int32_t *bSlot;       // [G*cap] slot in parent
    int32_t *freeList;    // [G][L] stack of free local ids: the private ones on top, overflow ids (largest first) at the bottom
    int32_t *ovfMap;      // [G][ovf_cap] shared block behind overflow id cap + j, -1 = not backed
    int32_t *poolFree;    // [pool_blocks] stack of free shared blocks: popped inside k_search, refilled by k_compact only
    int32_t *poolRet;     // [pool_blocks] shared blocks released by re-roots / restarts since the last k_compact
    int32_t *poolCtl;     // [0] top of poolFree, [1] entries of poolRet, [2] low-water mark of [0]
    double *rootP64;      // [G][APAD]
    double *noise;        // [G][APAD]
    double *uniforms;     // [G][max_moves]
    // fifo
    int32_t *fParent, *fSlot, *fBlk, *fEvalLocal, *fEvaluated;  // [G][2E]
    float *fValue;
    // requests of the current step
    int32_t *reqBlk, *reqParent, *reqMove;  // [G][E]; block ids are GLOBAL (g*cap + local)
    // compacted lists
    int32_t *evalIdx, *leafIn, *leafMv, *leafOut;  // [G*E]
    int32_t *evalModel;   // [G*E] which model evaluates each row of the evaluation list (two-model games; 0 otherwise)
    // records
    sgo_move_record *recs;
    uint32_t *recPacked;
    double *recPolicy;
    Counters *counters;
    DevStatus *dstatus;
    DevStatus *hstatus;   // pinned host
    int32_t *symLut;      // [8][A]
    uint8_t *stage;       // device staging area of sgo_start_games (one H2D copy per call)
    int last_n_eval;      // positions listed by the previous step
};

struct StageLayout {  // byte offsets into the staging area for a batch of n restarts (all 8-byte aligned)
    size_t slots, resign, resign2, first, noise, uniforms, total;
    int nu;
};
static inline size_t al8(size_t v) { return (v + 7) & ~(size_t)7; }
static StageLayout stage_layout(int n, int APAD, int nu, bool has_noise) {
    StageLayout L;
    L.nu = nu;
    L.slots = 0;
    L.resign = al8(sizeof(int32_t) * (size_t)n);
    L.resign2 = L.resign + al8(sizeof(float) * (size_t)n);
    L.first = L.resign2 + al8(sizeof(float) * (size_t)n);
    L.noise = L.first + al8(sizeof(int32_t) * (size_t)n);
    L.uniforms = L.noise + (has_noise ? sizeof(double) * (size_t)n * APAD : 0);
    L.total = L.uniforms + sizeof(double) * (size_t)n * nu;
    return L;
}

struct HostSide {  // not passed to kernels
    uint8_t *stage = nullptr;                  // pinned host twin of Ctx::stage
    size_t stage_cap = 0;
    hipEvent_t ev_stage = nullptr;             // k_start has consumed the staged batch (host block and device twin)
    bool stage_busy = false;
    hipStream_t last_stream = nullptr;         // stream of the last sgo_step (records are drained behind it)
    bool lds_attr_set = false;                 // k_search's > 64 KiB dynamic-LDS attribute has been set
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // bracket board_advance inside sgo_step
    double adv_ms = 0;
    long long adv_launches = 0, adv_positions = 0;
};

// ---------------------------------------------------------------------------------------- device helpers
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// argmax over the wave with "higher score, then lower index"; idx < 0 = no candidate
template <typename F>
__device__ __forceinline__ void wave_argmax(F &score, int &idx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        F os = __shfl_xor(score, o);
        int oi = __shfl_xor(idx, o);
        bool take = (oi >= 0) && (idx < 0 || os > score || (os == score && oi < idx));
        if (take) { score = os; idx = oi; }
    }
}

template <int S>
struct Eng {
    using G = Geo<S>;
    const Ctx &c;
    int g, lane;
    size_t gb0;  // g * cap
    __device__ Eng(const Ctx &cc, int gg) : c(cc), g(gg), lane(threadIdx.x & 63), gb0((size_t)gg * cc.cap) {}

    // physical block behind local id `blk` (uniform over the wave): private region, or one dependent load for an overflow id
    __device__ __forceinline__ size_t ph(int blk) const {
        if (blk < c.cap) return gb0 + blk;
        return (size_t)c.G * c.cap + (size_t)c.ovfMap[(size_t)g * c.ovf_cap + (blk - c.cap)];
    }
    __device__ __forceinline__ size_t slot_base(int blk) const { return ph(blk) * (size_t)G::APAD; }
    __device__ __forceinline__ bool legal_bit(int blk, int i) const {
        return (c.legal[ph(blk) * G::NW + (i >> 5)] >> (i & 31)) & 1u;
    }
    // a block for overflow id `blk` from the shared pool (all lanes call; false = the pool is empty)
    __device__ bool back(int blk, GameState &st) const {
        int phys = -1;
        if (lane == 0) {
            const int t = atomicSub(&c.poolCtl[0], 1);
            if (t > 0) {
                phys = c.poolFree[t - 1];
                atomicMin(&c.poolCtl[2], t - 1);
                c.ovfMap[(size_t)g * c.ovf_cap + (blk - c.cap)] = phys;
            } else {
                atomicAdd(&c.poolCtl[0], 1);
            }
        }
        phys = __shfl(phys, 0);
        if (blk - c.cap + 1 > st.ovf_hi) st.ovf_hi = blk - c.cap + 1;
        return phys >= 0;
    }
    // pop a free local id, backed; -1 = out of blocks (private region used up and the shared pool empty, or the id space)
    __device__ int alloc(GameState &st) const {
        if (st.free_top <= 0) return -1;
        const int nb = c.freeList[(size_t)g * c.L + st.free_top - 1];
        if (nb >= c.cap && !back(nb, st)) return -1;
        st.free_top--;
        if (st.free_top < st.min_free) st.min_free = st.free_top;
        return nb;
    }

    // play.py:308-323 on block `blk`; returns chosen slot or -1, and in `child` the chosen slot's child block (-1: a leaf).
    // Every array of the block is loaded UNCONDITIONALLY for all slots (expand() initialises all APAD slots, illegal ones with
    // zeros), so that the ~30 loads of a node are in flight together: one memory round trip per tree level instead of four
    // (legal word -> N / busy under that mask -> P / Q under the not-busy mask -> the winner's child pointer), which is what a
    // descent through a late-game tree spent its time on (k_search averaged 0.9 ms per call over a full 19x19 game, 0.2 ms at
    // the first plies).
    __device__ int top_one(int blk, bool f64, int &child) const {
        const size_t pb = ph(blk), sb = pb * (size_t)G::APAD;
        constexpr int J = (G::APAD + 63) / 64;
        int n_[J], cb_[J];
        float p_[J], q_[J];
        bool ex[J], busy[J];
        int sum = 0;
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int i = lane + 64 * j;
            const bool in = i < G::APAD;
            const uint32_t lw = in ? c.legal[pb * G::NW + (i >> 5)] : 0u;
            const int nv = in ? c.cN[sb + i] : 0;
            const int bz = in ? (int)c.cBusy[sb + i] : 1;
            p_[j] = in ? c.cP[sb + i] : 0.f;
            q_[j] = in ? c.cQ[sb + i] : 0.f;
            cb_[j] = in ? c.cB[sb + i] : -1;
            ex[j] = in && ((lw >> (i & 31)) & 1u);
            n_[j] = ex[j] ? nv : 0;
            busy[j] = ex[j] ? (bz > 0) : true;
            sum += n_[j];
        }
        // A position whose only legal move is the pass (the endgame's pass-pass chains, hundreds of levels deep: the reference's
        // search has no terminal test) needs no scores: its single child is chosen unless it is busy -- what the general path
        // below computes too (any finite score beats -100), minus two wave reductions and the score arithmetic.
        {
            constexpr int jN = G::N >> 6, lN = G::N & 63;
            bool only_pass = true;
#pragma unroll
            for (int j = 0; j < J; j++) {
                const unsigned long long m = __ballot(ex[j]);
                only_pass = only_pass && (m == (j == jN ? (1ull << lN) : 0ull));
            }
            if (only_pass) {
                const int bz = __shfl((int)busy[jN], lN);
                child = bz ? -1 : __shfl(cb_[jN], lN);
                return bz ? -1 : G::N;
            }
        }
        sum = wave_sum_i(sum);
        double tn = sqrt((double)sum);
        if (tn == 0) tn = 1;
        int best = -1;
        if (!f64) {
            float bs = -100.0f;
            const float tnf = (float)tn;
#pragma unroll
            for (int j = 0; j < J; j++) {
                int i = lane + 64 * j;
                if (!busy[j]) {
                    float u = p_[j] * tnf;
                    u = u / (float)(1.0 + (double)n_[j]);
                    float v = q_[j] + u;
                    if (v > bs) { bs = v; best = i; }
                }
            }
            wave_argmax<float>(bs, best);
        } else {
            double bs = -100.0;
            const double *p64 = c.rootP64 + (size_t)g * G::APAD;
#pragma unroll
            for (int j = 0; j < J; j++) {
                int i = lane + 64 * j;
                if (!busy[j]) {
                    double u = p64[i] * tn / (1. + (double)n_[j]);
                    double v = (double)q_[j] + u;
                    if (v > bs) { bs = v; best = i; }
                }
            }
            wave_argmax<double>(bs, best);
        }
        int cb = -1;
        if (best >= 0) {
#pragma unroll
            for (int j = 0; j < J; j++)
                if ((best >> 6) == j) cb = cb_[j];
            cb = __shfl(cb, best & 63);
        }
        child = cb;
        return best;
    }

    // tree_util.py:4-24.  Returns true and (pblk, slot) of the leaf (flagged busy), or false ("None").
    // `start` >= 0 resumes below the root: between two selections of one round nothing changes but busy flags at and below
    // the previous leaf's parent (no statistics move until the round's back-propagation), so a walk from the root would make
    // the same choices down to that parent -- the descent continues there instead of re-walking a path that, in the endgame's
    // deep pass-pass chains, is hundreds of levels long (k_search: 0.08 ms per call up to move 250, 0.8 ms at move 325).
    __device__ bool find_best_leaf(const GameState &st, int &pblk, int &slot, int start) const {
        int node = start >= 0 ? start : st.root_blk;
        for (;;) {
            int cb = -1;
            int a = top_one(node, st.root_f64 && node == st.root_blk, cb);
            if (a < 0) {
                const size_t pn = ph(node);
                int par = c.bParent[pn];
                if (par < 0) return false;
                int ps = c.bSlot[pn];
                if (lane == (ps & 63)) c.cBusy[slot_base(par) + ps] = 2;
                node = par;
                continue;
            }
            if (cb < 0) {
                if (lane == (a & 63)) c.cBusy[slot_base(node) + a] = 2;
                pblk = node;
                slot = a;
                return true;
            }
            node = cb;
        }
    }

    // children of block `blk` from a policy row (play.py:391-421); legal[] of the block must be valid
    __device__ void expand(int blk, const float *policy, const int32_t *lut, const double *noise, double eps) const {
        const size_t pb = ph(blk), sb = pb * (size_t)G::APAD;
        double *p64 = c.rootP64 + (size_t)g * G::APAD;
        // all loads of the node first (legal words, the symmetry LUT, then the gathered priors: two dependent round trips for the
        // whole node), then the stores: the slot-by-slot loop paid three dependent round trips per 64 slots -- 43 % of k_search
        constexpr int J = (G::APAD + 63) / 64;
        bool ex[J];
        int src[J];
        float pr[J];
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int i = lane + 64 * j;
            ex[j] = i < G::APAD && ((c.legal[pb * G::NW + ((i < G::APAD ? i : 0) >> 5)] >> (i & 31)) & 1u);
            src[j] = i < G::A ? lut[i] : 0;
        }
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int i = lane + 64 * j;
            pr[j] = i < G::A ? policy[src[j]] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < J; j++) {
            const int i = lane + 64 * j;
            if (i >= G::APAD) continue;
            float p = (ex[j] && i < G::A) ? pr[j] : 0.0f;
            if (noise) {
                double t = (1.0 - eps) * (double)p;
                double pd = ex[j] ? t + eps * noise[i] : 0.0;
                p64[i] = pd;
                p = (float)pd;
            }
            c.cP[sb + i] = p;
            c.cN[sb + i] = 0;
            c.cW[sb + i] = 0.f;
            c.cQ[sb + i] = 0.f;
            c.cB[sb + i] = -1;
            c.cBusy[sb + i] = 0;
        }
    }

    // nomodel_self_play.py:40-56 + the stats part of simulation_workers.py:50-53
    __device__ void back_propagate(GameState &st, int fi) const {
        const size_t fo = (size_t)g * (2 * MAXE) + fi;
        const int pb = c.fParent[fo], slot = c.fSlot[fo], nb = c.fBlk[fo];
        const float vraw = c.fValue[fo];
        const int leaf_player = white_to_play<S>(c.pos + ph(nb) * G::RW) ? -1 : 1;
        const float v = (leaf_player == st.original_player) ? vraw : -vraw;
        float leaf_value = 0.f;
        if (lane == (slot & 63)) {
            const size_t o = slot_base(pb) + slot;
            int n = c.cN[o] + 1;
            float w = c.cW[o] + v;
            c.cN[o] = n;
            c.cW[o] = w;
            c.cQ[o] = w / (float)n;
            c.cBusy[o] = 0;
            c.cB[o] = nb;
            leaf_value = w;
        }
        leaf_value = __shfl(leaf_value, slot & 63);
        // Walk to the root.  One memory round trip per level: the next level's parent / slot are requested together with this
        // level's statistics (the walk was three dependent round trips per level: parent, then slot, then N / W).
        const size_t ppb = ph(pb);
        int par = c.bParent[ppb], ps = c.bSlot[ppb];
        while (par >= 0) {
            const size_t pp = ph(par);
            const int npar = c.bParent[pp], nps = c.bSlot[pp];
            if (lane == (ps & 63)) {
                const size_t o = slot_base(par) + ps;
                int n = c.cN[o] + 1;
                float w = c.cW[o] + leaf_value;
                c.cN[o] = n;
                c.cW[o] = w;
                c.cQ[o] = w / (float)n;
                c.cBusy[o] = 0;
            }
            par = npar;
            ps = nps;
        }
        st.root_count += 1;
        st.root_value += leaf_value;
        st.root_mean = st.root_value / (float)st.root_count;
    }
};

// ---------------------------------------------------------------------------------------- k_search
template <int S>
__global__ __launch_bounds__(64) void k_search(Ctx c, const float *policy, const float *value, int sym_k_imm, const int32_t *sym_k_dev) {
    using G = Geo<S>;
    extern __shared__ int32_t lds[];
    int32_t *queue = lds;                       // [L]
    int32_t *sN = lds + c.L;                    // [APAD]
    float *sQ = (float *)(sN + G::APAD);        // [APAD]
    uint32_t *marks = (uint32_t *)(sQ + G::APAD);  // [(L+31)/32]
    const int g = blockIdx.x;
    const int lane = threadIdx.x;
    Eng<S> e(c, g);
    GameState st = c.gs[g];
    const int sym_k = sym_k_dev ? (*sym_k_dev & 7) : sym_k_imm;   // device-side value: one captured launch chain serves every symmetry
    const int32_t *lut = c.symLut + (size_t)sym_k * G::A;
    const size_t fbase = (size_t)g * (2 * MAXE);
    const size_t rbase = (size_t)g * c.E;
    st.n_req = 0;
    if (st.phase == PH_IDLE || st.phase == PH_DONE) {
        if (lane == 0) c.gs[g].n_req = 0;
        return;
    }
    bool run = true;

    auto finish = [&](int reason) {
        st.end_reason = reason;
        int bp = 0, wp = 0;
        if (lane == 0) score_record<S>(c.pos + e.ph(st.root_blk) * G::RW, bp, wp);
        bp = __shfl(bp, 0);
        wp = __shfl(wp, 0);
        double white = (double)wp + c.cfg.komi;
        st.winner = ((double)bp > white) ? 1 : (((double)bp == white) ? 0 : -1);
        st.black = bp;
        st.white = white;
        st.n_moves = st.move_n;
        st.last_player = st.player;
        st.phase = PH_DONE;
        run = false;
    };
    auto fail = [&](int code) {
        if (!st.error) st.error = code;
        st.phase = PH_DONE;
        run = false;
    };

    // ---- consume the evaluations requested by the previous step
#ifdef SGO_KSEARCH_PROFILE
    long long tq0 = clock64(), tq1;
#define SGO_TICK(k) do { tq1 = clock64(); if (lane == 0) atomicAdd(&c.counters->dbg[k], (unsigned long long)(tq1 - tq0)); tq0 = tq1; } while (0)
#else
#define SGO_TICK(k) do { } while (0)
#endif
    if (st.phase == PH_WAIT_ROOT) {
        if (!st.root_requested) {
            st.root_requested = 1;
            if (lane == 0) c.reqBlk[rbase] = (int32_t)e.ph(st.root_blk);
            st.n_req = 1;
            st.req_kind = 0;
            run = false;
        } else {
            st.root_requested = 0;
            const float *prow = policy + (size_t)st.eval_base * G::A;
            st.value = value[st.eval_base];
            st.has_value = 1;
            st.n_predict++;
            if (lane == 0) atomicAdd(&c.counters->total_evals, 1ull);
            // resign = resign_model1 if current == model1 else resign_model2 (nomodel_self_play.py:170-173)
            const bool use2 = c.cfg.two_model && st.cur_model == 1;
            if (use2 ? (st.has_resign2 && st.value <= st.resign2) : (st.has_resign && st.value <= st.resign)) {
                finish(1);
            } else {
                // "if not mcts_tree or not mcts_tree['subtree']": the root block carries children iff flag set
                bool expanded = c.bSlot[e.ph(st.root_blk)] != -2;  // -2 marks "block holds no children yet"
                if (!expanded) {
                    const double *noise = nullptr;
                    if (c.cfg.self_play) {
                        if (st.noise_used) fail(SGO_ERR_DRAWS);
                        noise = c.noise + (size_t)g * G::APAD;
                        st.noise_used = 1;
                    }
                    if (run) {
                        e.expand(st.root_blk, prow, lut, noise, c.cfg.dirichlet_epsilon);
                        if (lane == 0) c.bSlot[e.ph(st.root_blk)] = -1;
                        st.root_f64 = noise ? 1 : 0;
                        st.root_count = 0;
                        st.root_value = 0.f;
                        st.root_mean = 0.f;
                    }
                }
                if (run) {
                    st.rounds_left = c.cfg.sims / c.cfg.energy;
                    st.e_left = -1;
                    st.original_player = white_to_play<S>(c.pos + e.ph(st.root_blk) * G::RW) ? -1 : 1;
                    st.phase = PH_SEARCH;
                    if (c.cfg.sims < c.cfg.energy) fail(SGO_ERR_STATE);  // zero simulations: the reference cannot pick a move
                }
            }
        }
    } else {  // PH_SEARCH: every not-yet-evaluated fifo entry was evaluated by the previous step
        for (int fi = st.fifo_head; fi < st.fifo_tail; fi++) {
            const size_t fo = fbase + (fi % (2 * MAXE));
            if (c.fEvaluated[fo]) continue;
            const int row = st.eval_base + c.fEvalLocal[fo];
            e.expand(c.fBlk[fo], policy + (size_t)row * G::A, lut, nullptr, 0.0);
            if (lane == 0) {
                c.fValue[fo] = value[row];
                c.fEvaluated[fo] = 1;
                atomicAdd(&c.counters->total_evals, 1ull);
            }
            st.n_predict++;
        }
        __syncthreads();
        if (st.need_bp) {
            st.need_bp = 0;
            e.back_propagate(st, st.fifo_head % (2 * MAXE));
            st.fifo_head++;
            st.pre_bp++;
        }
    }

    // ---- async_simulate2 rounds (nomodel_self_play.py:59-82) until evaluations are needed
    SGO_TICK(0);
    while (run && st.phase == PH_SEARCH) {
        SGO_TICK(7);
        if (st.rounds_left == 0) {
            // ================= select_play tail + play_game_async body (:125-138, :180-216)
            if (st.halt_at == st.move_n) { st.phase = PH_DONE; run = false; break; }
            const size_t sb = e.slot_base(st.root_blk);
            for (int i = lane; i < G::APAD; i += 64) {
                bool ex = e.legal_bit(st.root_blk, i);
                sN[i] = ex ? c.cN[sb + i] : -1;
                sQ[i] = ex ? c.cQ[sb + i] : 0.f;
            }
            __syncthreads();
            int selected = -1;
            int err = 0;
            if (lane == 0) {
                if (st.temperature == 1) {
                    long total = 0;
                    for (int i = 0; i < G::A; i++) if (sN[i] > 0) total += sN[i];
                    double last = 0;
                    for (int i = 0; i < G::A; i++) if (sN[i] > 0) last += (double)sN[i] / (double)total;  // np.cumsum
                    if (total == 0 || st.i_uniform >= st.n_uniform) err = SGO_ERR_DRAWS;
                    else {
                        double u = c.uniforms[(size_t)g * c.max_moves + st.i_uniform];
                        double acc = 0;
                        int lastmv = -1;
                        for (int i = 0; i < G::A; i++) {
                            if (sN[i] <= 0) continue;
                            acc += (double)sN[i] / (double)total;
                            lastmv = i;
                            if (acc / last > u) { selected = i; break; }   // searchsorted(cdf/cdf[-1], u, 'right')
                        }
                        if (selected < 0) selected = lastmv;
                    }
                } else {
                    int bc = -1, ba = -1;
                    float bm = 0;
                    for (int i = 0; i < G::A; i++) {
                        if (sN[i] < 0) continue;
                        if (ba < 0 || sN[i] > bc || (sN[i] == bc && (sQ[i] > bm || (sQ[i] == bm && i > ba)))) {
                            bc = sN[i]; bm = sQ[i]; ba = i;
                        }
                    }
                    selected = ba;
                }
            }
            selected = __shfl(selected, 0);
            err = __shfl(err, 0);
            if (err) { fail(err); break; }
            if (st.temperature == 1) st.i_uniform++;
            // move_data record
            int ri = 0;
            if (lane == 0) ri = atomicAdd(&c.counters->rec_count, 1);
            ri = __shfl(ri, 0);
            if (ri >= c.rec_cap) { fail(SGO_ERR_CAPACITY); break; }
            if (lane == 0) {
                sgo_move_record r;
                r.game = g; r.game_seq = st.game_seq; r.move_n = st.move_n; r.action = selected;
                r.player = st.player; r.value = st.value;
                c.recs[ri] = r;
                atomicAdd(&c.counters->total_moves, 1ull);
            }
            for (int i = lane; i < G::RW; i += 64) c.recPacked[(size_t)ri * G::RW + i] = c.pos[e.ph(st.root_blk) * G::RW + i];
            for (int i = lane; i < G::A; i += 64) {
                double p = 0;
                if (e.legal_bit(st.root_blk, i)) p = st.root_f64 ? c.rootP64[(size_t)g * G::APAD + i] : (double)c.cP[sb + i];
                c.recPolicy[(size_t)ri * G::A + i] = p;
            }
            st.n_moves = st.move_n + 1;
            const bool is_pass = (selected == G::N);
            if (st.skipped_last && is_pass) { st.move_n += 0; finish(2); st.n_moves = st.move_n + 1; break; }
            st.skipped_last = is_pass ? 1 : 0;
            // re-root onto the chosen child, recycle every block that is no longer reachable
            int nr = 0;
            if (lane == (selected & 63)) nr = c.cB[sb + selected];
            nr = __shfl(nr, selected & 63);
            if (nr < 0) { fail(SGO_ERR_STATE); break; }
            float rv = 0, rm = 0; int rc = 0;
            if (lane == (selected & 63)) { rc = c.cN[sb + selected]; rv = c.cW[sb + selected]; rm = c.cQ[sb + selected]; }
            st.root_count = __shfl(rc, selected & 63);
            st.root_value = __shfl(rv, selected & 63);
            st.root_mean = __shfl(rm, selected & 63);
            const int mover = white_to_play<S>(c.pos + e.ph(st.root_blk) * G::RW) ? -1 : 1;
            // Two-model games (self_play == False, nomodel_self_play.py:203-208): the other player's tree follows the move when
            // it holds it ("if other_mcts and index in other_mcts['subtree']"); a child that was never evaluated there has an
            // empty subtree, i.e. the tree is rebuilt by new_tree() when its owner moves next -- here: no block, onr stays -1.
            int onr = -1;
            if (c.cfg.two_model && st.other_root >= 0 && c.bSlot[e.ph(st.other_root)] != -2) {
                const size_t osb = e.slot_base(st.other_root);
                int ocb = -1, oc = 0; float ov = 0, om = 0;
                if (lane == (selected & 63) && e.legal_bit(st.other_root, selected)) {
                    ocb = c.cB[osb + selected]; oc = c.cN[osb + selected]; ov = c.cW[osb + selected]; om = c.cQ[osb + selected];
                }
                onr = __shfl(ocb, selected & 63);
                st.other_count = __shfl(oc, selected & 63);
                st.other_value = __shfl(ov, selected & 63);
                st.other_mean = __shfl(om, selected & 63);
            }
            st.root_blk = nr;
            st.root_f64 = 0;
            if (lane == 0) {
                c.bParent[e.ph(nr)] = -1; c.bSlot[e.ph(nr)] = -1;
                if (onr >= 0) { c.bParent[e.ph(onr)] = -1; c.bSlot[e.ph(onr)] = -1; }
            }
            // mark: which blocks hang below the new root(s)?  Every allocated block is linked from its parent exactly once (the
            // graft in back_propagate) and carries that parent in bParent, so "reachable from the new root" = "the parent chain
            // ends in it".  All chains are resolved together by pointer jumping on a copy of the parent array in LDS:
            // O(log depth) passes of cap / 64 coalesced steps, instead of a breadth-first walk that paid one dependent memory
            // round trip per CHILD ARRAY of every kept block (22 ms for a late-game tree, measured; now tens of microseconds).
            constexpr int KEEP = -3, DROP = -4;
            int *par = queue;
            // local ids in use or used before: the private region plus the overflow ids backed so far in this game; ids beyond
            // Lu have never left the bottom of the free stack (entries [0, L - Lu), untouched here)
            const int Lu = c.cap + st.ovf_hi, base = c.L - Lu;
            for (int b = lane; b < Lu; b += 64) {
                int pv = DROP;
                if (b < c.cap) pv = c.bParent[e.gb0 + b];
                else {
                    const int ob = c.ovfMap[(size_t)g * c.ovf_cap + (b - c.cap)];      // -1: not backed = free
                    if (ob >= 0) pv = c.bParent[(size_t)c.G * c.cap + ob];
                }
                par[b] = pv < 0 ? DROP : pv;                       // other roots (the old one): dropped
            }
            __syncthreads();
            for (int i = base + lane; i < st.free_top; i += 64) par[c.freeList[(size_t)g * c.L + i]] = DROP;   // stale parents of free blocks
            __syncthreads();
            if (lane == 0) { par[nr] = KEEP; if (onr >= 0) par[onr] = KEEP; }
            __syncthreads();
            for (;;) {
                bool open = false;
                for (int b = lane; b < Lu; b += 64) {
                    const int pv = par[b];
                    if (pv >= 0) {
                        const int pp = par[pv];                    // KEEP / DROP resolve b; otherwise jump to the grandparent
                        par[b] = pp;
                        open |= pp >= 0;
                    }
                }
                __syncthreads();
                if (!__any(open)) break;
            }
            // rebuild the stack above `base`, ids DESCENDING so that the private ids pop before the overflow ids; an overflow id
            // that is free now gives its block back to the shared pool (poolRet: merged into poolFree by k_compact, so a pop in
            // this launch never meets a push)
            int ft = base;
            for (int b1 = ((Lu + 63) & ~63); b1 > 0; b1 -= 64) {
                const int b = b1 - 1 - lane;
                const bool fr = b < Lu && par[b] != KEEP;
                const unsigned long long m = __ballot(fr);
                if (fr) {
                    c.freeList[(size_t)g * c.L + ft + __popcll(m & ((1ull << lane) - 1ull))] = b;
                    if (b >= c.cap) {
                        const size_t mi = (size_t)g * c.ovf_cap + (b - c.cap);
                        const int ob = c.ovfMap[mi];
                        if (ob >= 0) {
                            c.poolRet[atomicAdd(&c.poolCtl[1], 1)] = ob;
                            c.ovfMap[mi] = -1;
                        }
                    }
                }
                ft += __popcll(m);
            }
            st.free_top = ft;
            __syncthreads();
            if (c.cfg.two_model) {
                if (onr < 0) {
                    // the other player's tree is empty: a fresh root block holding the position after the move (a copy of the
                    // mover's new root), unexpanded -- new_tree() fills it when that player's root evaluation arrives
                    onr = e.alloc(st);
                    if (onr < 0) { fail(SGO_ERR_CAPACITY); break; }
                    for (int i = lane; i < G::RW; i += 64) c.pos[e.ph(onr) * G::RW + i] = c.pos[e.ph(nr) * G::RW + i];
                    for (int i = lane; i < G::NW; i += 64) c.legal[e.ph(onr) * G::NW + i] = c.legal[e.ph(nr) * G::NW + i];
                    if (lane == 0) { c.bParent[e.ph(onr)] = -1; c.bSlot[e.ph(onr)] = -2; }
                    st.other_count = 0; st.other_value = 0.f; st.other_mean = 0.f;
                    __syncthreads();
                }
                // mcts_tree, other_mcts = other_mcts, mcts_tree (:218) and the models swap (:217)
                const int tb = st.root_blk, tc = st.root_count; const float tv = st.root_value, tm = st.root_mean;
                st.root_blk = onr; st.root_count = st.other_count; st.root_value = st.other_value; st.root_mean = st.other_mean;
                st.other_root = tb; st.other_count = tc; st.other_value = tv; st.other_mean = tm;
                st.cur_model ^= 1;
            }
            // board, player = make_play(...): the new root block already holds the position after the move
            st.player = mover;
            st.move_n++;
            if (st.move_n >= c.max_moves) { finish(0); break; }
            st.last_value = st.value;
            if (st.move_n == c.cfg.stop_exploration) st.temperature = 0;
            st.phase = PH_WAIT_ROOT;
            st.root_requested = 1;
            if (lane == 0) c.reqBlk[rbase] = (int32_t)e.ph(st.root_blk);
            st.n_req = 1;
            st.req_kind = 0;
            run = false;
            break;
        }
        if (st.e_left < 0) { st.e_left = c.cfg.energy; st.pre_bp = 0; }
        SGO_TICK(1);
        bool blocked = false;
        int resume = -1;                         // parent of the leaf selected last in this round; -1 = walk from the root
        while (st.e_left > 0) {
            int pb = -1, slot = -1;
            bool found = e.find_best_leaf(st, pb, slot, resume);
            resume = found ? pb : -1;
            if (found) {
                int n = 0;
                if (lane == (slot & 63)) n = c.cN[e.slot_base(pb) + slot];
                n = __shfl(n, slot & 63);
                if (n > 0) { st.e_left--; st.pre_bp++; continue; }   // "already simulated leaf node"
            } else {
                st.none_events++;
                if (lane == 0) atomicAdd(&c.counters->none_events, 1ull);
                if (st.fifo_tail == st.fifo_head) { fail(SGO_ERR_STATE); break; }  // the reference would block forever
                if (!c.fEvaluated[fbase + (st.fifo_head % (2 * MAXE))]) { st.need_bp = 1; blocked = true; break; }
                e.back_propagate(st, st.fifo_head % (2 * MAXE));   // statistics moved: the next walk starts at the root (resume = -1)
                st.fifo_head++;
                st.pre_bp++;
                continue;
            }
            const int nb = e.alloc(st);
            if (nb < 0) { fail(SGO_ERR_CAPACITY); break; }
            const size_t fo = fbase + (st.fifo_tail % (2 * MAXE));
            if (lane == 0) {
                c.bParent[e.ph(nb)] = pb;
                c.bSlot[e.ph(nb)] = slot;
                c.fParent[fo] = pb; c.fSlot[fo] = slot; c.fBlk[fo] = nb;
                c.fEvalLocal[fo] = st.n_req; c.fEvaluated[fo] = 0;
                c.reqBlk[rbase + st.n_req] = (int32_t)e.ph(nb);
                c.reqParent[rbase + st.n_req] = (int32_t)e.ph(pb);
                c.reqMove[rbase + st.n_req] = slot;
            }
            st.fifo_tail++;
            st.n_req++;
            st.req_kind = 1;
            st.e_left--;
        }
        SGO_TICK(2);
        __syncthreads();
        if (!run || blocked) break;
        bool pending = false;
        for (int fi = st.fifo_head; fi < st.fifo_tail; fi++)
            if (!c.fEvaluated[fbase + (fi % (2 * MAXE))]) pending = true;
        if (pending) break;
        const int nbp = c.cfg.energy - st.pre_bp;
        bool bad = false;
        for (int i = 0; i < nbp; i++) {
            if (st.fifo_head == st.fifo_tail) { bad = true; break; }
            e.back_propagate(st, st.fifo_head % (2 * MAXE));
            st.fifo_head++;
        }
        if (bad) { fail(SGO_ERR_STATE); break; }
        st.e_left = -1;
        st.rounds_left--;
    }
    SGO_TICK(3);
#ifdef SGO_KSEARCH_PROFILE
    if (lane == 0) atomicAdd(&c.counters->dbg[4], 1ull);
#endif
    // a game that has just failed (its tree is abandoned; the slot waits for a restart) hands its shared blocks back at once,
    // so that one starved game does not starve its neighbours for the steps until the host reacts
    if (st.error && st.ovf_hi > 0) {
        for (int j = lane; j < st.ovf_hi; j += 64) {
            const size_t mi = (size_t)g * c.ovf_cap + j;
            const int ob = c.ovfMap[mi];
            if (ob >= 0) {
                c.poolRet[atomicAdd(&c.poolCtl[1], 1)] = ob;
                c.ovfMap[mi] = -1;
            }
        }
        st.ovf_hi = 0;
    }
    if (lane == 0) c.gs[g] = st;
}

// ---------------------------------------------------------------------------------------- k_compact
// One block.  Exclusive prefix sums of the per-game request counts -> dense evaluation list (block ids
// in game-major order) and dense leaf list for board_advance; also folds the status words.
// The six block-wide scans (evaluation / leaf / root counts, active, done, first failing game) run as wave-level shuffles
// + one pass over the 16 wave totals: two block barriers instead of the twenty of a Hillis-Steele scan over LDS arrays.
__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o);
        if (lane >= o) v += u;
    }
    return v;
}
__global__ __launch_bounds__(1024) void k_compact(Ctx c) {
    __shared__ int wE[16], wL[16], wR[16], wAct[16], wDone[16], wErr[16];
    __shared__ int tot[6];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int per = (c.G + 1023) / 1024;
    const int g0 = t * per, g1 = min(c.G, g0 + per);
    int ne = 0, nl = 0, nr = 0, act = 0, done = 0, err = 0x7fffffff;
    for (int g = g0; g < g1; g++) {
        const GameState &s = c.gs[g];
        ne += s.n_req;
        if (s.req_kind == 1) nl += s.n_req;
        else nr += s.n_req;
        if (s.phase == PH_WAIT_ROOT || s.phase == PH_SEARCH) act++;
        if (s.phase == PH_DONE) done++;
        if (s.error && err == 0x7fffffff) err = g;
    }
    // inclusive scans inside the wave
    int iE = wave_incl_scan(ne, lane), iL = wave_incl_scan(nl, lane), iR = wave_incl_scan(nr, lane);
    int sAct = act, sDone = done, sErr = err;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        sAct += __shfl_xor(sAct, o);
        sDone += __shfl_xor(sDone, o);
        sErr = min(sErr, __shfl_xor(sErr, o));
    }
    if (lane == 63) { wE[w] = iE; wL[w] = iL; wR[w] = iR; }
    if (lane == 0) { wAct[w] = sAct; wDone[w] = sDone; wErr[w] = sErr; }
    __syncthreads();
    if (w == 0) {
        // exclusive scan of the 16 wave totals (lanes 0..15), totals of everything in tot[]
        const int vE = lane < 16 ? wE[lane] : 0, vL = lane < 16 ? wL[lane] : 0, vR = lane < 16 ? wR[lane] : 0;
        const int xE = wave_incl_scan(vE, lane), xL = wave_incl_scan(vL, lane), xR = wave_incl_scan(vR, lane);
        int a = lane < 16 ? wAct[lane] : 0, d = lane < 16 ? wDone[lane] : 0, e = lane < 16 ? wErr[lane] : 0x7fffffff;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o);
            d += __shfl_xor(d, o);
            e = min(e, __shfl_xor(e, o));
        }
        if (lane < 16) { wE[lane] = xE - vE; wL[lane] = xL - vL; wR[lane] = xR - vR; }
        if (lane == 15) { tot[0] = xE; tot[1] = xL; tot[2] = xR; }
        if (lane == 0) { tot[3] = a; tot[4] = d; tot[5] = e; }
    }
    __syncthreads();
    // phase 1: every game's bases (thread t owns games [g0, g1))
    int be = wE[w] + iE - ne, bl = wL[w] + iL - nl, br = wR[w] + iR - nr;
    for (int g = g0; g < g1; g++) {
        GameState &s = c.gs[g];
        s.eval_base = be;
        s.list_base = (s.req_kind == 1) ? bl : br;
        be += s.n_req;
        if (s.req_kind == 1) bl += s.n_req;
        else br += s.n_req;
    }
    __syncthreads();
    // phase 2: one (game, request) pair per thread and pass -- all loads of a pair are independent of every other pair's, so
    // the 8 192 pairs of a 1 024-game step are a few memory round trips instead of eight dependent ones per game
    const int pairs = c.G * c.E;
    for (int e = t; e < pairs; e += 1024) {
        const int g = e / c.E, j = e - g * c.E;
        const GameState &s = c.gs[g];
        if (j >= s.n_req) continue;
        const int row = s.eval_base + j, blk = c.reqBlk[e];
        c.evalIdx[row] = blk;
        c.evalModel[row] = s.cur_model;
        if (s.req_kind == 1) {
            const int li = s.list_base + j;
            c.leafIn[li] = c.reqParent[e];
            c.leafMv[li] = c.reqMove[e];
            c.leafOut[li] = blk;
        }
    }
    // shared blocks released during this step's k_search (and by k_start since the previous step) go back on the free stack
    // here, between two k_search launches: pops and pushes never run concurrently
    {
        const int nret = c.poolCtl[1], top = c.poolCtl[0];
        for (int i = t; i < nret; i += 1024) c.poolFree[top + i] = c.poolRet[i];
        __syncthreads();
        if (t == 0 && nret > 0) { c.poolCtl[0] = top + nret; c.poolCtl[1] = 0; }
    }
    if (t == 1023) {
        DevStatus d;
        d.n_eval = tot[0]; d.n_leaf = tot[1]; d.n_records = c.counters->rec_count;
        d.n_active = tot[3]; d.n_done = tot[4];
        int eg = tot[5];
        d.error_game = (eg == 0x7fffffff) ? -1 : eg;
        d.error = (eg == 0x7fffffff) ? 0 : c.gs[eg].error;
        d.n_root = tot[2];
        d.total_moves = c.counters->total_moves; d.total_evals = c.counters->total_evals;
        d.none_events = c.counters->none_events;
        *c.dstatus = d;
    }
}

// (re)start listed game slots: empty board in block 0, everything else free
// Inputs come from the staging area filled by ONE host-to-device copy: slots[n], resign[n] (NaN = None), then
// (optionally) noise[n][APAD] and uniforms[n][nu].
template <int S>
__global__ __launch_bounds__(64) void k_start(Ctx c, int n, StageLayout L, int has_noise, int has_uniforms) {
    using G = Geo<S>;
    const int k = blockIdx.x;
    if (k >= n) return;
    const int g = reinterpret_cast<const int32_t *>(c.stage + L.slots)[k];
    const float *resign = reinterpret_cast<const float *>(c.stage + L.resign);
    const int lane = threadIdx.x;
    if (has_noise) {
        const double *src = reinterpret_cast<const double *>(c.stage + L.noise) + (size_t)k * c.APAD;
        for (int i = lane; i < c.APAD; i += 64) c.noise[(size_t)g * c.APAD + i] = src[i];
    }
    if (has_uniforms) {
        const double *src = reinterpret_cast<const double *>(c.stage + L.uniforms) + (size_t)k * L.nu;
        for (int i = lane; i < L.nu; i += 64) c.uniforms[(size_t)g * c.max_moves + i] = src[i];
    }
    const size_t gb0 = (size_t)g * c.cap;
    GameState st;
    memset(&st, 0, sizeof st);
    st.phase = PH_WAIT_ROOT;
    st.root_blk = 0;
    st.player = 1;
    st.temperature = (0 == c.cfg.stop_exploration) ? 0 : 1;
    st.e_left = -1;
    st.halt_at = -1;
    st.n_uniform = has_uniforms ? L.nu : 0;   // a game that samples a move without a supplied draw fails with SGO_ERR_DRAWS
    st.game_seq = c.gs[g].phase == PH_IDLE && c.gs[g].game_seq == 0 && c.gs[g].n_predict == 0 ? 0 : c.gs[g].game_seq + 1;
    float r = resign[k];
    st.has_resign = !(r != r) && r != 0.f;   // `if resign and ...` (nomodel_self_play.py:171): None and 0.0 never resign
    st.resign = st.has_resign ? r : 0.f;
    const float r2 = reinterpret_cast<const float *>(c.stage + L.resign2)[k];
    st.has_resign2 = !(r2 != r2) && r2 != 0.f;
    st.resign2 = st.has_resign2 ? r2 : 0.f;
    st.first_model = c.cfg.two_model ? (reinterpret_cast<const int32_t *>(c.stage + L.first)[k] & 1) : 0;
    st.cur_model = st.first_model;
    st.other_root = -1;
    if (c.max_moves == 0) st.phase = PH_DONE;
    for (int i = lane; i < G::RW; i += 64) c.pos[gb0 * G::RW + i] = 0;
    for (int i = lane; i < G::NW; i += 64) {
        uint32_t w = 0xffffffffu;
        if (i == G::NW - 1) {
            int bits = G::A - 32 * (G::NW - 1);
            w = (bits >= 32) ? 0xffffffffu : ((1u << bits) - 1u);
        }
        c.legal[gb0 * G::NW + i] = w;
    }
    // whatever the slot's previous game still holds of the shared pool goes back to it
    for (int j = lane; j < c.ovf_cap; j += 64) {
        const size_t mi = (size_t)g * c.ovf_cap + j;
        const int ob = c.ovfMap[mi];
        if (ob >= 0) {
            c.poolRet[atomicAdd(&c.poolCtl[1], 1)] = ob;
            c.ovfMap[mi] = -1;
        }
    }
    for (int b = lane; b < c.L - 1; b += 64) c.freeList[(size_t)g * c.L + b] = c.L - 1 - b;  // pops give 1,2,3,...: private ids first
    st.free_top = c.L - 1;
    st.min_free = c.L - 1;
    st.ovf_hi = 0;
    if (lane == 0) {
        c.bParent[gb0] = -1;
        c.bSlot[gb0] = -2;  // no children yet
        c.gs[g] = st;
    }
}

template <int S>
static size_t search_lds(const Ctx &c) {
    return sizeof(int32_t) * ((size_t)c.L + 2 * Geo<S>::APAD + (c.L + 31) / 32 + 4);
}

}  // namespace sgo

using namespace sgo;

struct sgo_ctx {
    Ctx c;
    HostSide h;
};

#define CK(call)                      \
    do {                              \
        int _r = (call);              \
        if (_r != SGO_OK) return _r;  \
    } while (0)

template <typename T>
static int dalloc(T **p, size_t n) {
    SGO_HIP(hipMalloc((void **)p, sizeof(T) * (n ? n : 1)));
    SGO_HIP(hipMemset(*p, 0, sizeof(T) * (n ? n : 1)));
    return SGO_OK;
}

static int ctx_alloc(Ctx &c) {
    const size_t nb = (size_t)c.G * c.cap + (size_t)c.pool_blocks;       // private regions, then the shared pool
    CK(dalloc(&c.gs, c.G));
    CK(dalloc(&c.pos, nb * c.RW));
    CK(dalloc(&c.legal, nb * c.NW));
    CK(dalloc(&c.cP, nb * c.APAD));
    CK(dalloc(&c.cW, nb * c.APAD));
    CK(dalloc(&c.cQ, nb * c.APAD));
    CK(dalloc(&c.cN, nb * c.APAD));
    CK(dalloc(&c.cB, nb * c.APAD));
    CK(dalloc(&c.cBusy, nb * c.APAD));
    CK(dalloc(&c.bParent, nb));
    CK(dalloc(&c.bSlot, nb));
    CK(dalloc(&c.freeList, (size_t)c.G * c.L));
    CK(dalloc(&c.ovfMap, (size_t)c.G * c.ovf_cap));
    CK(dalloc(&c.poolFree, (size_t)c.pool_blocks));
    CK(dalloc(&c.poolRet, (size_t)c.pool_blocks));
    CK(dalloc(&c.poolCtl, 4));
    if (c.ovf_cap > 0) SGO_HIP(hipMemset(c.ovfMap, 0xff, sizeof(int32_t) * (size_t)c.G * c.ovf_cap));   // -1: not backed
    {
        std::vector<int32_t> ids((size_t)c.pool_blocks);
        for (size_t i = 0; i < ids.size(); i++) ids[i] = (int32_t)(ids.size() - 1 - i);                  // pops give 0, 1, 2, ...
        if (!ids.empty()) SGO_HIP(hipMemcpy(c.poolFree, ids.data(), sizeof(int32_t) * ids.size(), hipMemcpyHostToDevice));
        const int32_t ctl[4] = {(int32_t)c.pool_blocks, 0, (int32_t)c.pool_blocks, 0};
        SGO_HIP(hipMemcpy(c.poolCtl, ctl, sizeof ctl, hipMemcpyHostToDevice));
    }
    CK(dalloc(&c.rootP64, (size_t)c.G * c.APAD));
    CK(dalloc(&c.noise, (size_t)c.G * c.APAD));
    CK(dalloc(&c.uniforms, (size_t)c.G * (c.max_moves ? c.max_moves : 1)));
    const size_t nf = (size_t)c.G * 2 * MAXE;
    CK(dalloc(&c.fParent, nf));
    CK(dalloc(&c.fSlot, nf));
    CK(dalloc(&c.fBlk, nf));
    CK(dalloc(&c.fEvalLocal, nf));
    CK(dalloc(&c.fEvaluated, nf));
    CK(dalloc(&c.fValue, nf));
    const size_t nr = (size_t)c.G * c.E;
    CK(dalloc(&c.reqBlk, nr));
    CK(dalloc(&c.reqParent, nr));
    CK(dalloc(&c.reqMove, nr));
    CK(dalloc(&c.evalIdx, nr));
    CK(dalloc(&c.leafIn, nr));
    CK(dalloc(&c.leafMv, nr));
    CK(dalloc(&c.leafOut, nr));
    CK(dalloc(&c.evalModel, nr));
    CK(dalloc(&c.recs, (size_t)c.rec_cap));
    CK(dalloc(&c.recPacked, (size_t)c.rec_cap * c.RW));
    CK(dalloc(&c.recPolicy, (size_t)c.rec_cap * c.A));
    CK(dalloc(&c.counters, 1));
    CK(dalloc(&c.dstatus, 1));
    CK(dalloc(&c.symLut, (size_t)8 * c.A));
    {
        const StageLayout L = stage_layout(c.G, c.APAD, c.max_moves, true);
        CK(dalloc(&c.stage, L.total));
    }
    SGO_HIP(hipHostMalloc((void **)&c.hstatus, sizeof(DevStatus), hipHostMallocDefault));
    memset(c.hstatus, 0, sizeof(DevStatus));
    std::vector<int32_t> lut((size_t)8 * c.A);
    for (int k = 0; k < 8; k++) build_sym_lut(c.S, k, lut.data() + (size_t)k * c.A);
    SGO_HIP(hipMemcpy(c.symLut, lut.data(), sizeof(int32_t) * lut.size(), hipMemcpyHostToDevice));
    return SGO_OK;
}

static void ctx_free(Ctx &c) {
    void *ptrs[] = {c.gs, c.pos, c.legal, c.cP, c.cW, c.cQ, c.cN, c.cB, c.cBusy, c.bParent, c.bSlot, c.freeList, c.ovfMap, c.poolFree, c.poolRet, c.poolCtl,
                    c.rootP64, c.noise, c.uniforms, c.fParent, c.fSlot, c.fBlk, c.fEvalLocal, c.fEvaluated, c.fValue,
                    c.reqBlk, c.reqParent, c.reqMove, c.evalIdx, c.leafIn, c.leafMv, c.leafOut, c.evalModel, c.recs, c.recPacked,
                    c.recPolicy, c.counters, c.dstatus, c.symLut, c.stage};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (c.hstatus) (void)hipHostFree(c.hstatus);
}

extern "C" {

sgo_ctx *sgo_ctx_create(const sgo_config *cfg) {
    if (!cfg || !size_ok(cfg->size) || cfg->n_games < 1 || cfg->energy < 1 || cfg->energy > MAXE || cfg->sims < 0) {
        set_error("sgo_ctx_create: bad config");
        return nullptr;
    }
    if (hipSetDevice(cfg->device_id) != hipSuccess) { set_error("sgo_ctx_create: hipSetDevice failed (no HIP device?)"); return nullptr; }
    sgo_ctx *x = new sgo_ctx();
    Ctx &c = x->c;
    memset((void *)&c.cfg, 0, sizeof c.cfg);
    c.cfg = *cfg;
    if (c.cfg.two_model && c.cfg.self_play) { set_error("sgo_ctx_create: two_model games are not self-play games"); delete x; return nullptr; }
    c.S = cfg->size; c.A = c.S * c.S + 1; c.NW = sgo_plane_words(c.S); c.RW = sgo_packed_words(c.S);
    c.APAD = 32 * c.NW; c.G = cfg->n_games; c.E = cfg->energy;
    // Tree blocks.  A search adds <= sims blocks and a move keeps the chosen child's subtree, so a tree settles at sims / (1 - f)
    // blocks, f = the share of the visits under the chosen child; over 512 full-length 19x19 / 400-sim games (20-block net) the
    // high-water mark was 4.2 sims in the median, 9.9 sims at the 99th percentile, 12.1 sims at most
    // (profiles/r03_fullgame_headline.json).  Sizing every game for the worst tree wastes four fifths of the memory (round 2:
    // 20 sims + 128 per game = 74 GB at 1 024 games) and still loses games where memory forces less (config 5 at 1 024 games:
    // 11.8 sims).  So: a PRIVATE region per game that covers most games (8 sims + 128), local ids beyond it backed on demand
    // from a pool SHARED by the context (2 sims per game, at least 12 sims), up to the id space k_search's LDS work queue
    // allows.  blocks_per_game > 0 fixes the private region; shared_blocks: > 0 fixes the pool, 0 = the default pool when the
    // private region is the default too and none otherwise (the round-2 behaviour: a fixed per-game pool), < 0 = default pool.
    const long lds_max = (160L * 1024 / 4 - 2L * c.APAD - 4) * 32 / 33 - 32;
    const size_t per_block = sizeof(uint32_t) * ((size_t)c.RW + c.NW) + (size_t)c.APAD * (4 * 5 + 1) + 3 * sizeof(int32_t);
    long priv = cfg->blocks_per_game > 0 ? cfg->blocks_per_game : 8L * cfg->sims + 128;
    long pool = cfg->shared_blocks > 0 ? cfg->shared_blocks
                : (cfg->shared_blocks < 0 || cfg->blocks_per_game <= 0) ? std::max(2L * cfg->sims * c.G, 12L * cfg->sims + 128) : 0;
    if (cfg->blocks_per_game <= 0 && priv > lds_max) priv = lds_max;
    {
        size_t free_b = 0, total_b = 0;
        if ((cfg->blocks_per_game <= 0 || cfg->shared_blocks <= 0) && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 0) {
            const double fit = (double)(free_b / 10 * 6) / (double)per_block;        // blocks that fit in 60 % of the free memory
            const double want = (double)priv * c.G + (double)pool;
            if (want > fit) {
                const double r = fit / want;
                if (cfg->blocks_per_game <= 0) priv = (long)(priv * r);
                if (cfg->shared_blocks <= 0) pool = (long)(pool * r);
            }
        }
    }
    if (priv < cfg->energy + 2) priv = cfg->energy + 2;
    long L = priv + pool;
    // default id space per game: 20 sims + 128 (the most any game has needed is 12.1 sims); at 19x19 / 400 sims that keeps
    // k_search's LDS work queue at 36 KB, i.e. four games per CU resident at once (at 40 sims: 65 KB, two per CU, +25 % per call)
    if (L > 20L * cfg->sims + 128 && cfg->shared_blocks <= 0) L = std::max(priv, 20L * cfg->sims + 128);
    if (L > lds_max) L = std::max(priv, lds_max);
    c.cap = (int)priv;
    c.L = (int)L;
    c.ovf_cap = (int)(L - priv);
    c.pool_blocks = c.ovf_cap > 0 ? pool : 0;
    c.max_moves = cfg->num_moves < 0 ? 2 * c.S * c.S : cfg->num_moves;
    c.rec_cap = 2 * c.G + 16;
    c.last_n_eval = 0;
    c.gs = nullptr; c.hstatus = nullptr;
    // LDS budget of k_search: queue[L] + sN/sQ + marks
    size_t lds = sizeof(int32_t) * ((size_t)c.L + 2 * c.APAD + (c.L + 31) / 32 + 4);
    if (lds > 160 * 1024) { set_error("sgo_ctx_create: blocks_per_game too large for the LDS work queue"); delete x; return nullptr; }
    if (ctx_alloc(c) != SGO_OK) { ctx_free(c); delete x; return nullptr; }
    x->h.stage_cap = stage_layout(c.G, c.APAD, c.max_moves, true).total;
    if (hipHostMalloc((void **)&x->h.stage, x->h.stage_cap, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&x->h.ev_stage, hipEventDisableTiming) != hipSuccess) {
        set_error("sgo_ctx_create: staging allocation failed");
        ctx_free(c); delete x; return nullptr;
    }
    if (hipEventCreate(&x->h.ev0) != hipSuccess || hipEventCreate(&x->h.ev1) != hipSuccess) {
        set_error("sgo_ctx_create: hipEventCreate failed");
        ctx_free(c); delete x; return nullptr;
    }
    return x;
}

int sgo_blocks_per_game(sgo_ctx *x) {
    if (!x) { set_error("sgo_blocks_per_game: bad argument"); return SGO_ERR_ARG; }
    return x->c.cap;
}

int sgo_pool_info(sgo_ctx *x, int64_t *out, int n) {
    if (!x || !out || n < 0) { set_error("sgo_pool_info: bad argument"); return SGO_ERR_ARG; }
    Ctx &c = x->c;
    int32_t ctl[4] = {0, 0, 0, 0};
    SGO_HIP(hipDeviceSynchronize());
    SGO_HIP(hipMemcpy(ctl, c.poolCtl, sizeof ctl, hipMemcpyDeviceToHost));
    const int64_t v[6] = {c.cap, c.L, c.pool_blocks, (int64_t)ctl[0] + ctl[1], ctl[2], c.G};
    for (int i = 0; i < n && i < 6; i++) out[i] = v[i];
    return SGO_OK;
}

void sgo_ctx_destroy(sgo_ctx *x) {
    if (!x) return;
    (void)hipSetDevice(x->c.cfg.device_id);
    (void)hipDeviceSynchronize();
    ctx_free(x->c);
    if (x->h.stage) (void)hipHostFree(x->h.stage);
    if (x->h.ev_stage) (void)hipEventDestroy(x->h.ev_stage);
    if (x->h.ev0) (void)hipEventDestroy(x->h.ev0);
    if (x->h.ev1) (void)hipEventDestroy(x->h.ev1);
    delete x;
}

static int start_games_impl(sgo_ctx *x, int n, const int32_t *slots, const double *noise, const double *uniforms, int n_uniforms,
                            const float *resign, const float *resign2, const int32_t *first_model, void *stream);

int sgo_start_games(sgo_ctx *x, int n, const int32_t *slots, const double *noise, const double *uniforms, int n_uniforms,
                    const float *resign, void *stream) {
    return start_games_impl(x, n, slots, noise, uniforms, n_uniforms, resign, nullptr, nullptr, stream);
}

int sgo_start_games2(sgo_ctx *x, int n, const int32_t *slots, const double *uniforms, int n_uniforms, const float *resign_model1,
                     const float *resign_model2, const int32_t *first_model, void *stream) {
    if (!x || !x->c.cfg.two_model) { set_error("sgo_start_games2: the context was not created with two_model = 1"); return SGO_ERR_STATE; }
    return start_games_impl(x, n, slots, nullptr, uniforms, n_uniforms, resign_model1, resign_model2, first_model, stream);
}

static int start_games_impl(sgo_ctx *x, int n, const int32_t *slots, const double *noise, const double *uniforms, int n_uniforms,
                            const float *resign, const float *resign2, const int32_t *first_model, void *stream) {
    if (!x || n < 0 || (n && !slots)) { set_error("sgo_start_games: bad argument"); return SGO_ERR_ARG; }
    Ctx &c = x->c;
    if (n == 0) return SGO_OK;
    if (n > c.G || n_uniforms < 0) { set_error("sgo_start_games: too many slots"); return SGO_ERR_ARG; }
    for (int i = 0; i < n; i++)
        if (slots[i] < 0 || slots[i] >= c.G) { set_error("sgo_start_games: slot out of range"); return SGO_ERR_ARG; }
    hipStream_t st = (hipStream_t)stream;
    SGO_HIP(hipSetDevice(c.cfg.device_id));
    // the pinned staging area is reused: wait (host side, this one copy only) until the previous batch has left it
    if (x->h.stage_busy) { SGO_HIP(hipEventSynchronize(x->h.ev_stage)); x->h.stage_busy = false; }
    const int nu = uniforms ? (n_uniforms < c.max_moves ? n_uniforms : c.max_moves) : 0;
    const StageLayout L = stage_layout(n, c.APAD, nu, noise != nullptr);
    uint8_t *h = x->h.stage;
    memcpy(h + L.slots, slots, sizeof(int32_t) * n);
    float *hr = reinterpret_cast<float *>(h + L.resign), *hr2 = reinterpret_cast<float *>(h + L.resign2);
    int32_t *hf = reinterpret_cast<int32_t *>(h + L.first);
    for (int i = 0; i < n; i++) {
        hr[i] = resign ? resign[i] : NAN;
        hr2[i] = resign2 ? resign2[i] : NAN;
        hf[i] = first_model ? first_model[i] : 0;
    }
    if (noise) {
        double *hn = reinterpret_cast<double *>(h + L.noise);
        for (int i = 0; i < n; i++) {
            memcpy(hn + (size_t)i * c.APAD, noise + (size_t)i * c.A, sizeof(double) * c.A);
            memset(hn + (size_t)i * c.APAD + c.A, 0, sizeof(double) * (c.APAD - c.A));
        }
    }
    if (nu > 0) {
        double *hu = reinterpret_cast<double *>(h + L.uniforms);
        for (int i = 0; i < n; i++) memcpy(hu + (size_t)i * nu, uniforms + (size_t)i * n_uniforms, sizeof(double) * nu);
    }
    // one host-to-device copy, then one kernel, both on the caller's stream: ordered against the steps before and after
    SGO_HIP(hipMemcpyAsync(c.stage, h, L.total, hipMemcpyHostToDevice, st));
    SGO_DISPATCH(c.S, k_start<kS><<<dim3(n), dim3(64), 0, st>>>(c, n, L, noise != nullptr, nu > 0));
    SGO_HIP(hipGetLastError());
    // recorded BEHIND k_start: the event guards the pinned block and its device twin `c.stage` alike, so the next batch
    // (whatever stream it arrives on) is staged only after this one's kernel has read its slots / draws
    SGO_HIP(hipEventRecord(x->h.ev_stage, st));
    x->h.stage_busy = true;
    return SGO_OK;
}

// Everything a step runs on the GPU, queued on `st` without waiting: k_search (consumes the evaluations of the list the previous
// step produced, selects / moves), k_compact (dense evaluation list + leaf list + status words), board_advance for the new
// leaves, and the copy of the status words to pinned host memory.  No host synchronisation, no host-side shape dependence: the
// chain can be captured in a hipGraph and replayed (timing = false then: event timestamps do not exist inside a graph).
static int step_enqueue(sgo_ctx *x, const float *d_policy, const float *d_value, int sym_k, const int32_t *d_sym_k, hipStream_t st,
                        bool timing) {
    Ctx &c = x->c;
    x->h.last_stream = st;
    SGO_DISPATCH(c.S, {
        const size_t lds = search_lds<kS>(c);
        if (lds > 64 * 1024 && !x->h.lds_attr_set) {
            SGO_HIP(hipFuncSetAttribute((const void *)k_search<kS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            x->h.lds_attr_set = true;
        }
        k_search<kS><<<dim3(c.G), dim3(64), lds, st>>>(c, d_policy, d_value, sym_k, d_sym_k);
    });
    SGO_HIP(hipGetLastError());
    k_compact<<<dim3(1), dim3(1024), 0, st>>>(c);
    SGO_HIP(hipGetLastError());
    // board_advance for the new leaves (parents = leafIn and freshly allocated blocks = leafOut are disjoint block sets => the
    // split form), bracketed by HIP events on this stream when timing
    if (timing) SGO_HIP(hipEventRecord(x->h.ev0, st));
    CK(launch_advance_split(c.S, c.G * c.E, &c.dstatus->n_leaf, c.pos, c.leafIn, c.leafMv, nullptr, c.pos, c.leafOut, c.legal,
                            c.leafOut, nullptr, st));
    if (timing) SGO_HIP(hipEventRecord(x->h.ev1, st));
    SGO_HIP(hipMemcpyAsync(c.hstatus, c.dstatus, sizeof(DevStatus), hipMemcpyDeviceToHost, st));
    return SGO_OK;
}

static void read_status(sgo_ctx *x, sgo_status *out) {
    Ctx &c = x->c;
    const DevStatus &d = *c.hstatus;
    out->n_eval = d.n_eval; out->n_records = d.n_records; out->n_active = d.n_active; out->n_done = d.n_done;
    out->error = d.error; out->error_game = d.error_game; out->total_moves = (int64_t)d.total_moves;
    out->total_evals = (int64_t)d.total_evals; out->none_events = (int64_t)d.none_events;
    c.last_n_eval = d.n_eval;
}

int sgo_step(sgo_ctx *x, const float *d_policy, const float *d_value, int sym_k, void *stream, sgo_status *out) {
    if (!x || !out || sym_k < 0 || sym_k > 7) { set_error("sgo_step: bad argument"); return SGO_ERR_ARG; }
    Ctx &c = x->c;
    hipStream_t st = (hipStream_t)stream;
    if (c.last_n_eval > 0 && (!d_policy || !d_value)) {
        set_error("sgo_step: the previous step listed positions to evaluate; policy/value are required");
        return SGO_ERR_STATE;
    }
    const int rc = step_enqueue(x, d_policy, d_value, sym_k, nullptr, st, true);
    if (rc != SGO_OK) return rc;
    SGO_HIP(hipStreamSynchronize(st));
    if (c.hstatus->n_leaf > 0) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, x->h.ev0, x->h.ev1) == hipSuccess) {
            x->h.adv_ms += ms; x->h.adv_launches += 1; x->h.adv_positions += c.hstatus->n_leaf;
        }
    }
    read_status(x, out);
    return SGO_OK;
}

int sgo_step_enqueue(sgo_ctx *x, const float *d_policy, const float *d_value, const int32_t *d_sym_k, void *stream) {
    if (!x || !d_policy || !d_value || !d_sym_k) { set_error("sgo_step_enqueue: bad argument"); return SGO_ERR_ARG; }
    return step_enqueue(x, d_policy, d_value, 0, d_sym_k, (hipStream_t)stream, false);
}

int sgo_step_status(sgo_ctx *x, sgo_status *out) {
    if (!x || !out) { set_error("sgo_step_status: bad argument"); return SGO_ERR_ARG; }
    read_status(x, out);
    return SGO_OK;
}

int sgo_eval_list(sgo_ctx *x, const uint32_t **d_records, const int32_t **d_index, const int32_t **d_models) {
    if (!x) { set_error("sgo_eval_list: bad argument"); return SGO_ERR_ARG; }
    if (d_records) *d_records = x->c.pos;
    if (d_index) *d_index = x->c.evalIdx;
    if (d_models) *d_models = x->c.evalModel;
    return x->c.G * x->c.E;
}

int sgo_eval_models(sgo_ctx *x, int cap, int32_t *models) {
    if (!x || !models || cap < 0) { set_error("sgo_eval_models: bad argument"); return SGO_ERR_ARG; }
    Ctx &c = x->c;
    const int n = c.last_n_eval;
    if (n > cap) { set_error("sgo_eval_models: caller buffer too small"); return SGO_ERR_ARG; }
    if (n > 0) {
        SGO_HIP(hipMemcpyAsync(models, c.evalModel, sizeof(int32_t) * n, hipMemcpyDeviceToHost, x->h.last_stream));
        SGO_HIP(hipStreamSynchronize(x->h.last_stream));
    }
    return n;
}

int sgo_collect(sgo_ctx *x, int sym_k, int layout, int dtype, void *d_nn_in, void *stream) {
    if (!x || !d_nn_in) { set_error("sgo_collect: bad argument"); return SGO_ERR_ARG; }
    Ctx &c = x->c;
    if (sym_k < 0 || sym_k > 7 || layout < 0 || layout > 2 || dtype < 0 || dtype > 1) { set_error("sgo_collect: bad argument"); return SGO_ERR_ARG; }
    return launch_nn_pack(c.S, c.last_n_eval, c.pos, c.evalIdx, sym_k, layout, dtype, d_nn_in, (hipStream_t)stream);
}

int sgo_drain_records(sgo_ctx *x, int cap, sgo_move_record *recs, uint32_t *packed, double *policy) {
    if (!x || cap < 0) { set_error("sgo_drain_records: bad argument"); return SGO_ERR_ARG; }
    Ctx &c = x->c;
    // sgo_step has synchronised its stream before returning, so the records (and the count in the status it returned)
    // are complete; everything here is queued behind that stream and waited for once
    hipStream_t st = x->h.last_stream;
    int n = c.hstatus->n_records;
    if (n > c.rec_cap) n = c.rec_cap;
    if (n > cap) { set_error("sgo_drain_records: caller buffer too small"); return SGO_ERR_ARG; }
    if (n > 0) {
        if (recs) SGO_HIP(hipMemcpyAsync(recs, c.recs, sizeof(sgo_move_record) * n, hipMemcpyDeviceToHost, st));
        if (packed) SGO_HIP(hipMemcpyAsync(packed, c.recPacked, sizeof(uint32_t) * (size_t)n * c.RW, hipMemcpyDeviceToHost, st));
        if (policy) SGO_HIP(hipMemcpyAsync(policy, c.recPolicy, sizeof(double) * (size_t)n * c.A, hipMemcpyDeviceToHost, st));
    }
    SGO_HIP(hipMemsetAsync(&c.counters->rec_count, 0, sizeof(int32_t), st));
    SGO_HIP(hipStreamSynchronize(st));
    c.hstatus->n_records = 0;
    return n;
}

int sgo_game_results(sgo_ctx *x, int n, const int32_t *slots, sgo_game_result *out) {
    if (!x || n < 0 || !out) { set_error("sgo_game_results: bad argument"); return SGO_ERR_ARG; }
    Ctx &c = x->c;
    std::vector<GameState> all(c.G);
    SGO_HIP(hipMemcpyAsync(all.data(), c.gs, sizeof(GameState) * c.G, hipMemcpyDeviceToHost, x->h.last_stream));
    SGO_HIP(hipStreamSynchronize(x->h.last_stream));
    for (int i = 0; i < n; i++) {
        int g = slots ? slots[i] : i;
        if (g < 0 || g >= c.G) { set_error("sgo_game_results: slot out of range"); return SGO_ERR_ARG; }
        const GameState &s = all[g];
        out[i].winner = s.winner; out[i].black = s.black; out[i].white = s.white; out[i].end_reason = s.end_reason;
        out[i].n_moves = s.n_moves; out[i].last_player = s.last_player; out[i].done = (s.phase == PH_DONE) ? 1 : 0;
        out[i].first_model = s.first_model;
        out[i].blocks_high_water = c.L - s.min_free;
        if (s.error) out[i].done = s.error;
    }
    return SGO_OK;
}

int sgo_set_halt(sgo_ctx *x, int slot, int move_n) {
    if (!x || slot < 0 || slot >= x->c.G) { set_error("sgo_set_halt: bad argument"); return SGO_ERR_ARG; }
    Ctx &c = x->c;
    SGO_HIP(hipDeviceSynchronize());
    GameState s;
    SGO_HIP(hipMemcpy(&s, c.gs + slot, sizeof s, hipMemcpyDeviceToHost));
    s.halt_at = move_n;
    SGO_HIP(hipMemcpy(c.gs + slot, &s, sizeof s, hipMemcpyHostToDevice));
    return SGO_OK;
}

// ---- introspection (host side walks a snapshot of one game's blocks)
struct Snap {
    GameState s;
    std::vector<float> P, W, Q;
    std::vector<int32_t> N, B;
    std::vector<uint8_t> busy;
    std::vector<uint32_t> legal;
    std::vector<double> p64;
    std::vector<int32_t> ovf;     // the game's row of the overflow map
};
// physical block behind local id `blk` of game g, given the game's row of the overflow map
static size_t host_phys(const Ctx &c, int g, int blk, const std::vector<int32_t> &ovf) {
    if (blk < c.cap) return (size_t)g * c.cap + blk;
    return (size_t)c.G * c.cap + (size_t)ovf[blk - c.cap];
}
static int ovf_row(Ctx &c, int g, std::vector<int32_t> &ovf) {
    ovf.assign((size_t)c.ovf_cap, -1);
    if (c.ovf_cap > 0)
        SGO_HIP(hipMemcpy(ovf.data(), c.ovfMap + (size_t)g * c.ovf_cap, sizeof(int32_t) * c.ovf_cap, hipMemcpyDeviceToHost));
    return SGO_OK;
}
static int snapshot(Ctx &c, int g, Snap &sn) {
    SGO_HIP(hipDeviceSynchronize());
    SGO_HIP(hipMemcpy(&sn.s, c.gs + g, sizeof(GameState), hipMemcpyDeviceToHost));
    if (sn.s.error) { set_error("this slot's game failed (its tree was abandoned and its shared blocks released)"); return SGO_ERR_STATE; }
    CK(ovf_row(c, g, sn.ovf));
    int hi = 0;                                            // local ids [0, cap + hi) may hold blocks
    for (int j = 0; j < c.ovf_cap; j++)
        if (sn.ovf[j] >= 0) hi = j + 1;
    const size_t nb = (size_t)c.cap + hi, ns = nb * c.APAD;
    sn.P.resize(ns); sn.W.resize(ns); sn.Q.resize(ns); sn.N.resize(ns); sn.B.resize(ns); sn.busy.resize(ns);
    sn.legal.resize(nb * c.NW); sn.p64.resize(c.APAD);
    // the private region in one piece, then every backed overflow block on its own
    auto pull = [&](size_t dst_blk, size_t src_blk, size_t n_blk) -> int {
        const size_t k = n_blk * c.APAD, d = dst_blk * c.APAD, o = src_blk * c.APAD;
        SGO_HIP(hipMemcpy(sn.P.data() + d, c.cP + o, sizeof(float) * k, hipMemcpyDeviceToHost));
        SGO_HIP(hipMemcpy(sn.W.data() + d, c.cW + o, sizeof(float) * k, hipMemcpyDeviceToHost));
        SGO_HIP(hipMemcpy(sn.Q.data() + d, c.cQ + o, sizeof(float) * k, hipMemcpyDeviceToHost));
        SGO_HIP(hipMemcpy(sn.N.data() + d, c.cN + o, sizeof(int32_t) * k, hipMemcpyDeviceToHost));
        SGO_HIP(hipMemcpy(sn.B.data() + d, c.cB + o, sizeof(int32_t) * k, hipMemcpyDeviceToHost));
        SGO_HIP(hipMemcpy(sn.busy.data() + d, c.cBusy + o, k, hipMemcpyDeviceToHost));
        SGO_HIP(hipMemcpy(sn.legal.data() + dst_blk * c.NW, c.legal + src_blk * c.NW, sizeof(uint32_t) * n_blk * c.NW, hipMemcpyDeviceToHost));
        return SGO_OK;
    };
    CK(pull(0, (size_t)g * c.cap, c.cap));
    for (int j = 0; j < hi; j++)
        if (sn.ovf[j] >= 0) CK(pull((size_t)c.cap + j, (size_t)c.G * c.cap + sn.ovf[j], 1));
    SGO_HIP(hipMemcpy(sn.p64.data(), c.rootP64 + (size_t)g * c.APAD, sizeof(double) * c.APAD, hipMemcpyDeviceToHost));
    return SGO_OK;
}
static bool snap_exists(const Ctx &c, const Snap &sn, int blk, int i) {
    return (sn.legal[(size_t)blk * c.NW + (i >> 5)] >> (i & 31)) & 1u;
}

int sgo_root_table(sgo_ctx *x, int slot, int32_t *N, float *W, float *Q, double *P, int8_t *EX, int32_t *root_count,
                   float *root_value) {
    if (!x || slot < 0 || slot >= x->c.G) { set_error("sgo_root_table: bad argument"); return SGO_ERR_ARG; }
    Ctx &c = x->c;
    Snap sn;
    CK(snapshot(c, slot, sn));
    const int rb = sn.s.root_blk;
    std::vector<int32_t> bslot(1);
    SGO_HIP(hipMemcpy(bslot.data(), c.bSlot + host_phys(c, slot, rb, sn.ovf), sizeof(int32_t), hipMemcpyDeviceToHost));
    const bool expanded = bslot[0] != -2;
    for (int a = 0; a < c.A; a++) {
        const size_t o = (size_t)rb * c.APAD + a;
        bool ex = expanded && snap_exists(c, sn, rb, a);
        if (N) N[a] = ex ? sn.N[o] : 0;
        if (W) W[a] = ex ? sn.W[o] : 0;
        if (Q) Q[a] = ex ? sn.Q[o] : 0;
        if (P) P[a] = ex ? (sn.s.root_f64 ? sn.p64[a] : (double)sn.P[o]) : 0;
        if (EX) EX[a] = ex ? 1 : 0;
    }
    if (root_count) *root_count = sn.s.root_count;
    if (root_value) *root_value = sn.s.root_value;
    return SGO_OK;
}

static void ser_rec(const Ctx &c, const Snap &sn, int blk, bool f64, uint8_t *buf, int64_t cap, int64_t &off, int64_t &nn,
                    int64_t &ne, int depth = -1) {
    const int rec = depth >= 0 ? 40 : 32;   // depth >= 0: extended 40-byte records (+ i depth, i pad) for sgo_tree_dump
    for (int a = 0; a < c.A; a++) {
        if (!snap_exists(c, sn, blk, a)) continue;
        const size_t o = (size_t)blk * c.APAD + a;
        const int32_t cb = sn.B[o];
        if (buf && off + rec <= cap) {
            int32_t i32;
            double p = f64 ? sn.p64[a] : (double)sn.P[o];
            if (depth >= 0) { i32 = depth; memcpy(buf + off + 32, &i32, 4); i32 = 0; memcpy(buf + off + 36, &i32, 4); }
            i32 = a; memcpy(buf + off, &i32, 4);
            i32 = sn.N[o]; memcpy(buf + off + 4, &i32, 4);
            memcpy(buf + off + 8, &sn.W[o], 4);
            memcpy(buf + off + 12, &sn.Q[o], 4);
            memcpy(buf + off + 16, &p, 8);
            i32 = sn.busy[o]; memcpy(buf + off + 24, &i32, 4);
            i32 = cb >= 0 ? 1 : 0; memcpy(buf + off + 28, &i32, 4);
        }
        off += rec;
        nn++;
        if (cb >= 0) { ne++; ser_rec(c, sn, cb, false, buf, cap, off, nn, ne, depth >= 0 ? depth + 1 : -1); }
    }
}

int64_t sgo_tree_serialize(sgo_ctx *x, int slot, uint8_t *buf, int64_t cap, int64_t *n_nodes, int64_t *n_expanded) {
    if (!x || slot < 0 || slot >= x->c.G) { set_error("sgo_tree_serialize: bad argument"); return SGO_ERR_ARG; }
    Ctx &c = x->c;
    Snap sn;
    CK(snapshot(c, slot, sn));
    int32_t bslot = 0;
    SGO_HIP(hipMemcpy(&bslot, c.bSlot + host_phys(c, slot, sn.s.root_blk, sn.ovf), sizeof(int32_t), hipMemcpyDeviceToHost));
    int64_t off = 0, nn = 0, ne = 0;
    if (bslot != -2) ser_rec(c, sn, sn.s.root_blk, sn.s.root_f64 != 0, buf, cap, off, nn, ne);
    if (n_nodes) *n_nodes = nn;
    if (n_expanded) *n_expanded = ne;
    return off;
}

int64_t sgo_tree_dump(sgo_ctx *x, int slot, uint8_t *buf, int64_t cap, int64_t *n_nodes) {
    if (!x || slot < 0 || slot >= x->c.G) { set_error("sgo_tree_dump: bad argument"); return SGO_ERR_ARG; }
    Ctx &c = x->c;
    Snap sn;
    CK(snapshot(c, slot, sn));
    int32_t bslot = 0;
    SGO_HIP(hipMemcpy(&bslot, c.bSlot + host_phys(c, slot, sn.s.root_blk, sn.ovf), sizeof(int32_t), hipMemcpyDeviceToHost));
    int64_t off = 0, nn = 0, ne = 0;
    if (bslot != -2) ser_rec(c, sn, sn.s.root_blk, sn.s.root_f64 != 0, buf, cap, off, nn, ne, 0);
    if (n_nodes) *n_nodes = nn;
    return off;
}

int sgo_game_board(sgo_ctx *x, int slot, int32_t *board17) {
    if (!x || slot < 0 || slot >= x->c.G || !board17) { set_error("sgo_game_board: bad argument"); return SGO_ERR_ARG; }
    Ctx &c = x->c;
    SGO_HIP(hipDeviceSynchronize());
    GameState s;
    SGO_HIP(hipMemcpy(&s, c.gs + slot, sizeof s, hipMemcpyDeviceToHost));
    int32_t *d = nullptr;
    const size_t bsz = sizeof(int32_t) * (size_t)c.S * c.S * 17;
    SGO_HIP(hipMalloc((void **)&d, bsz));
    std::vector<int32_t> ovf;
    CK(ovf_row(c, slot, ovf));
    int r = sgo_unpack_dev(c.S, 1, c.pos + host_phys(c, slot, s.root_blk, ovf) * c.RW, d, nullptr);
    if (r == SGO_OK) {
        hipError_t e = hipMemcpy(board17, d, bsz, hipMemcpyDeviceToHost);
        if (e != hipSuccess) r = hip_fail(e, "hipMemcpy", __FILE__, __LINE__);
    }
    (void)hipFree(d);
    return r;
}

/* Diagnostic: cycles per phase of k_search summed over games and calls (zeros unless built with -DSGO_KSEARCH_PROFILE):
 * [0] consuming evaluations (expand), [2] selection, [7] the round's back-propagation, [3] the move step, [4] wave-calls. */
int sgo_debug_counters(sgo_ctx *x, unsigned long long *out, int n) {
    if (!x || !out || n < 0) return SGO_ERR_ARG;
    Counters h;
    SGO_HIP(hipMemcpy(&h, x->c.counters, sizeof h, hipMemcpyDeviceToHost));
    for (int i = 0; i < n && i < 8; i++) out[i] = h.dbg[i];
    return SGO_OK;
}

int sgo_advance_timing(sgo_ctx *x, double *total_ms, int64_t *launches, int64_t *positions) {
    if (!x) { set_error("sgo_advance_timing: bad argument"); return SGO_ERR_ARG; }
    if (total_ms) *total_ms = x->h.adv_ms;
    if (launches) *launches = x->h.adv_launches;
    if (positions) *positions = x->h.adv_positions;
    x->h.adv_ms = 0; x->h.adv_launches = 0; x->h.adv_positions = 0;
    return SGO_OK;
}

}  // extern "C"
