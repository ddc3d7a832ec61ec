/*
 * sgo_oracle.c -- CPU restatement of the sejonggo self-play hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker / the timed CPU comparator.  The product path (sejonggo_amd + libsgo_hip.so) never
 * links, imports or falls back to it.
 *
 * What it restates (file:line are relative to the reference, drsagitn/sejonggo @ v0):
 *   rules      play.py:31-34 index2coord, :106-112 get_real_board, :159-180 capture_group,
 *              :182-217 take_stones, :219-224 swap_player, :226-242 make_play, :71-104 legal_moves,
 *              :244-292 color_board/_get_points/get_winner, :295-299 game_init
 *   symmetry   symmetry.py:12-42 LUT construction, :45-114 transforms, :117-132 SYMMETRIES
 *   tree       play.py:308-352 selectors, :376-421 new_tree/new_subtree,
 *              tree_util.py:4-32, nomodel_self_play.py:40-56 back_propagation,
 *              :59-82 async_simulate2, :114-140 select_play, :142-271 play_game_async,
 *              simulation_workers.py:42-54 basic_tasks2
 *
 * Arithmetic regime: the reference as it runs under numpy >= 2 (NEP 50) in the golden-vector
 * container: node value / mean_value / PUCT score in float32, except at a root whose priors were
 * mixed with Dirichlet noise (float64 priors => float64 score).  See DESIGN.md "float regime".
 *
 * Parity is PINNED: tests/test_oracle_golden.py checks every function here against
 * the tests/golden npz fixtures, which tests/golden/gen_golden.py produced by running the Python reference.
 *
 * Board tensor = the reference's: int32 [S][S][17] (NHWC, batch dim dropped), plane 2k = to-play
 * side's stones k plies ago, 2k+1 = opponent's, plane 16 = to-play colour (+1 black / -1 white).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MAXS 19
#define MAXN (MAXS * MAXS)
#define MAXA (MAXN + 1)
#define NPL 17
#define BD(b, S, y, x, c) ((b)[(((y) * (S)) + (x)) * NPL + (c)])

#define ORA_ERR_OCCUPIED (-101)
#define ORA_ERR_RANGE (-102)

/* ------------------------------------------------------------------ rules ------------------ */

/* play.py:295-299 */
void ora_game_init(int S, int32_t *b) {
    memset(b, 0, sizeof(int32_t) * S * S * NPL);
    for (int i = 0; i < S * S; i++) b[i * NPL + 16] = 1;
}

/* play.py:106-112: absolute colours, black=+1 white=-1 */
static void real_board(int S, const int32_t *b, int8_t *rb) {
    int player = b[16];
    for (int i = 0; i < S * S; i++) {
        int d = b[i * NPL + 0] - b[i * NPL + 1];
        rb[i] = (int8_t)(player == 1 ? d : -d);
    }
}
void ora_get_real_board(int S, const int32_t *b, int8_t *rb) { real_board(S, b, rb); }

/* play.py:219-224 */
int ora_swap_player(int S, int32_t *b) {
    int player = b[16];
    for (int i = 0; i < S * S; i++) {
        int32_t *p = b + i * NPL;
        for (int k = 0; k < 16; k += 2) {
            int32_t t = p[k];
            p[k] = p[k + 1];
            p[k + 1] = t;
        }
    }
    player = (player == 1) ? -1 : 1;
    for (int i = 0; i < S * S; i++) b[i * NPL + 16] = player;
    return player;
}

/* play.py:159-180.  Recursive DFS in the reference's neighbour order; returns 0 ("None") as soon as
 * an empty neighbour is met, else the number of members written to grp (x,y pairs, visit order). */
static const int DX[4] = {1, -1, 0, 0};
static const int DY[4] = {0, 0, 1, -1};

static int cg_rec(int S, const int8_t *rb, int x, int y, int16_t *grp, int *n) {
    int c = rb[y * S + x];
    for (int d = 0; d < 4; d++) {
        int nx = x + DX[d], ny = y + DY[d];
        int seen = 0;
        for (int i = 0; i < *n; i++)
            if (grp[2 * i] == nx && grp[2 * i + 1] == ny) { seen = 1; break; }
        if (seen) continue;
        if (!(0 <= nx && nx < S && 0 <= ny && ny < S)) continue;
        int dc = rb[ny * S + nx];
        if (dc == 0) return 0;
        else if (dc == c) {
            grp[2 * (*n)] = (int16_t)nx;
            grp[2 * (*n) + 1] = (int16_t)ny;
            (*n)++;
            if (!cg_rec(S, rb, nx, ny, grp, n)) return 0;
        }
    }
    return 1;
}

int ora_capture_group(int S, const int8_t *rb, int x, int y, int16_t *grp) {
    int n = 1;
    grp[0] = (int16_t)x;
    grp[1] = (int16_t)y;
    if (!cg_rec(S, rb, x, y, grp, &n)) return 0;
    return n;
}

/* play.py:182-217 */
static void take_stones(int S, int x, int y, int32_t *b) {
    int8_t rb[MAXN];
    int16_t grp[2 * MAXN];
    real_board(S, b, rb);
    int player = (b[16] == 1) ? 1 : -1;
    for (int d = 0; d < 4; d++) {
        int nx = x + DX[d], ny = y + DY[d];
        if (!(0 <= nx && nx < S && 0 <= ny && ny < S)) continue;
        if (rb[ny * S + nx] == 0) continue;
        if (rb[ny * S + nx] == player) continue;
        int n = ora_capture_group(S, rb, nx, ny, grp);
        for (int i = 0; i < n; i++) {
            int gx = grp[2 * i], gy = grp[2 * i + 1];
            BD(b, S, gy, gx, 1) = 0;
            rb[gy * S + gx] = 0;
        }
    }
    for (int d = 0; d < 5; d++) {
        int nx = x + (d < 4 ? DX[d] : 0), ny = y + (d < 4 ? DY[d] : 0);
        if (!(0 <= nx && nx < S && 0 <= ny && ny < S)) continue;
        if (rb[ny * S + nx] == 0) continue;
        if (rb[ny * S + nx] != player) continue;
        int n = ora_capture_group(S, rb, nx, ny, grp);
        for (int i = 0; i < n; i++) {
            int gx = grp[2 * i], gy = grp[2 * i + 1];
            BD(b, S, gy, gx, 0) = 0; /* suicide is executed, not rejected */
            rb[gy * S + gx] = 0;
        }
    }
}

/* play.py:226-242.  color: 0 = None.  Returns the player who moved (+1/-1) or a negative error
 * where the reference would raise (assert on an occupied point, IndexError out of range). */
int ora_make_play(int S, int32_t *b, int x, int y, int color) {
    int player;
    if (y != S && !(0 <= x && x < S && 0 <= y && y < S)) return ORA_ERR_RANGE;
    if (color != 0 && color != b[16]) player = ora_swap_player(S, b);
    else player = b[16];
    for (int i = 0; i < S * S; i++) { /* board[...,2:16] = board[...,0:14] (overlap-safe copy) */
        int32_t *p = b + i * NPL;
        for (int k = 15; k >= 2; k--) p[k] = p[k - 2];
    }
    if (y != S) {
        if (BD(b, S, y, x, 1) != 0 || BD(b, S, y, x, 0) != 0) return ORA_ERR_OCCUPIED;
        BD(b, S, y, x, 0) = 1;
        take_stones(S, x, y, b);
    }
    ora_swap_player(S, b);
    return player;
}

/* play.py:45-56 get_surrounding: up, right, down, left */
static int surrounding(int S, int x, int y, int *ox, int *oy) {
    int n = 0;
    if (y - 1 >= 0) { ox[n] = x; oy[n] = y - 1; n++; }
    if (x + 1 < S) { ox[n] = x + 1; oy[n] = y; n++; }
    if (y + 1 < S) { ox[n] = x; oy[n] = y + 1; n++; }
    if (x - 1 >= 0) { ox[n] = x - 1; oy[n] = y; n++; }
    return n;
}

/* play.py:71-104.  mask[a] = 1 => illegal; mask[S*S] (pass) = 0. */
void ora_legal_moves(int S, const int32_t *b, uint8_t *mask) {
    int N = S * S;
    int8_t rb[MAXN], cb[MAXN];
    int16_t grp[2 * MAXN];
    int ko_cnt = 0;
    for (int i = 0; i < N; i++) {
        mask[i] = (b[i * NPL + 0] != 0) || (b[i * NPL + 1] != 0);
        if (b[i * NPL + 2] - b[i * NPL + 0] == 1) ko_cnt++;
    }
    if (ko_cnt == 1)
        for (int i = 0; i < N; i++)
            if (b[i * NPL + 2] - b[i * NPL + 0] == 1) mask[i] = 1;
    int player = b[16];
    real_board(S, b, rb);
    for (int index = 0; index < N; index++) {
        if (mask[index] != 0) continue;
        int col = index % S, row = index / S; /* col, row = index2coord(index) */
        memcpy(cb, rb, N);
        cb[row * S + col] = (int8_t)player;
        int capture_others = 0;
        int sx[4], sy[4];
        /* get_surrounding(row, col) yields (rs, cs) pairs; the board is square so the bounds agree */
        int ns = surrounding(S, row, col, sx, sy);
        for (int k = 0; k < ns; k++) {
            int rs = sx[k], cs = sy[k];
            if (player != rb[rs * S + cs] && ora_capture_group(S, cb, cs, rs, grp)) {
                capture_others = 1;
                break;
            }
        }
        if (capture_others) continue;
        if (ora_capture_group(S, rb, col, row, grp)) mask[index] = 1;
    }
    mask[N] = 0;
}

/* play.py:244-271 */
static void color_adjoint(int S, int i, int j, int color, int8_t *bd) {
    if (i > 0 && bd[(i - 1) * S + j] == 0) { bd[(i - 1) * S + j] = (int8_t)color; color_adjoint(S, i - 1, j, color, bd); }
    if (i < S - 1 && bd[(i + 1) * S + j] == 0) { bd[(i + 1) * S + j] = (int8_t)color; color_adjoint(S, i + 1, j, color, bd); }
    if (j > 0 && bd[i * S + j - 1] == 0) { bd[i * S + j - 1] = (int8_t)color; color_adjoint(S, i, j - 1, color, bd); }
    if (j < S - 1 && bd[i * S + j + 1] == 0) { bd[i * S + j + 1] = (int8_t)color; color_adjoint(S, i, j + 1, color, bd); }
}
void ora_color_board(int S, const int8_t *rb, int color, int8_t *out) {
    memcpy(out, rb, S * S);
    for (int i = 0; i < S; i++)
        for (int j = 0; j < S; j++)
            if (out[i * S + j] == color) color_adjoint(S, i, j, color, out);
}
/* play.py:286-292: counts of total = colored(+1) + colored(-1) for values -2..2 (index v+2) */
void ora_get_points(int S, const int8_t *rb, int *counts5) {
    int8_t c1[MAXN], c2[MAXN];
    ora_color_board(S, rb, 1, c1);
    ora_color_board(S, rb, -1, c2);
    for (int k = 0; k < 5; k++) counts5[k] = 0;
    for (int i = 0; i < S * S; i++) counts5[c1[i] + c2[i] + 2]++;
}
/* play.py:274-284 */
int ora_get_winner(int S, const int32_t *b, double komi, int *black_out, double *white_out) {
    int8_t rb[MAXN];
    int cnt[5];
    real_board(S, b, rb);
    ora_get_points(S, rb, cnt);
    int black = cnt[3] + cnt[4];
    double white = (double)(cnt[1] + cnt[0]) + komi;
    if (black_out) *black_out = black;
    if (white_out) *white_out = white;
    if ((double)black > white) return 1;
    if ((double)black == white) return 0;
    return -1;
}

/* ------------------------------------------------------------------ symmetry --------------- */
/* Canonical order: 0 id, 1 left_diagonal, 2 vertical_axis, 3 horizontal_axis, 4 rot90, 5 rot180,
 * 6 rot270 (symmetry.py:117-125 SYMMETRIES), 7 right_diagonal (implemented and tested by the
 * reference, not listed in SYMMETRIES). */

static long py_round(double v) { return lrint(v); } /* round-half-even like Python; inputs are ~integers */

/* symmetry.py:12-27 */
static void rotation_indexes(int S, double angle, int32_t *lut) {
    for (int x = 0; x < S; x++)
        for (int y = 0; y < S; y++) {
            int index = x + S * y;
            double fx = x - (S - 1) / 2.0, fy = y - (S - 1) / 2.0;
            double nx = cos(angle) * fx - sin(angle) * fy;
            double ny = sin(angle) * fx + cos(angle) * fy;
            nx += (S - 1) / 2.0;
            ny += (S - 1) / 2.0;
            lut[index] = (int32_t)py_round(nx + S * ny);
        }
    lut[S * S] = S * S;
}
/* symmetry.py:29-42 */
static void axis_symmetry_indexes(int S, double angle, int32_t *lut) {
    for (int x = 0; x < S; x++)
        for (int y = 0; y < S; y++) {
            int index = x + S * y;
            double fx = x - (S - 1) / 2.0, fy = y - (S - 1) / 2.0;
            double nx = cos(2 * angle) * fx + sin(2 * angle) * fy;
            double ny = sin(2 * angle) * fx - cos(2 * angle) * fy;
            nx += (S - 1) / 2.0;
            ny += (S - 1) / 2.0;
            lut[index] = (int32_t)py_round(nx + S * ny);
        }
    lut[S * S] = S * S;
}
void ora_sym_lut(int S, int k, int32_t *lut) {
    const double pi = 3.141592653589793;
    switch (k) {
    case 0: for (int i = 0; i <= S * S; i++) lut[i] = i; break;
    case 1: axis_symmetry_indexes(S, pi / 4., lut); break;
    case 2: axis_symmetry_indexes(S, pi / 2., lut); break;
    case 3: axis_symmetry_indexes(S, 0, lut); break;
    case 4: rotation_indexes(S, pi / 2., lut); break;
    case 5: rotation_indexes(S, pi, lut); break;
    case 6: rotation_indexes(S, 3 * pi / 2, lut); break;
    case 7: axis_symmetry_indexes(S, 3 * pi / 4., lut); break;
    }
}
/* symmetry.py:45-114 board transforms, one board [S][S][17]; out[i][j] = in[si][sj] */
void ora_sym_board(int S, int k, const int32_t *in, int32_t *out) {
    for (int i = 0; i < S; i++)
        for (int j = 0; j < S; j++) {
            int si = i, sj = j;
            switch (k) {
            case 0: break;
            case 1: si = j; sj = i; break;                 /* transpose axes (0,2,1,3) */
            case 2: sj = S - 1 - j; break;                 /* board[:,:,j] = board[:,:,S-1-j] */
            case 3: si = S - 1 - i; break;                 /* board[:,i] = board[:,S-1-i] */
            case 4: si = j; sj = S - 1 - i; break;         /* np.rot90 k=1 */
            case 5: si = S - 1 - i; sj = S - 1 - j; break; /* k=2 */
            case 6: si = S - 1 - j; sj = i; break;         /* k=3 */
            case 7: si = S - 1 - j; sj = S - 1 - i; break; /* rot180(transpose) */
            }
            memcpy(out + (i * S + j) * NPL, in + (si * S + sj) * NPL, sizeof(int32_t) * NPL);
        }
}
/* reverse_*: policy[:, :] = policy[:, SWAP]  => out[a] = in[SWAP[a]] */
void ora_sym_policy_inverse(int S, int k, const float *in, float *out) {
    int32_t lut[MAXA];
    ora_sym_lut(S, k, lut);
    for (int a = 0; a <= S * S; a++) out[a] = in[lut[a]];
}

/* ------------------------------------------------------------------ tree ------------------- */

typedef struct ONode {
    int index;
    int count;
    float value;      /* np.float32 accumulate (python int 0 until first add) */
    float mean;       /* value / float(count) in float32 */
    double p;         /* prior; exact widening of a float32 unless p_f64 */
    int p_f64;        /* 1 when the prior went through the Dirichlet mix (float64 regime) */
    int vloss;        /* virtual_loss */
    struct ONode *parent;
    struct ONode **child; /* NULL <=> subtree == {} ; else A pointers, NULL where no child */
    int nchild;
} ONode;

static ONode *node_new(int index, double p, int p_f64, ONode *parent) {
    ONode *n = (ONode *)calloc(1, sizeof(ONode));
    n->index = index;
    n->p = p;
    n->p_f64 = p_f64;
    n->parent = parent;
    return n;
}
static void node_free(ONode *n, int A) {
    if (!n) return;
    if (n->child) {
        for (int a = 0; a < A; a++) node_free(n->child[a], A);
        free(n->child);
    }
    free(n);
}

/* play.py:391-421.  noise == NULL <=> add_noise False. */
static void new_subtree(int S, const float *policy, const int32_t *board, ONode *parent, const double *noise,
                        double eps) {
    int A = S * S + 1;
    uint8_t mask[MAXA];
    ora_legal_moves(S, board, mask);
    parent->child = (ONode **)calloc(A, sizeof(ONode *));
    parent->nchild = 0;
    for (int a = 0; a < A; a++) {
        if (mask[a]) continue;
        double p;
        int f64 = 0;
        if (noise) {
            /* (1 - eps) * masked_float32_array: numpy.ma wraps the python float into a 0-d float64
             * array, so the product is float64 (verified against the golden games), then + float64 */
            double t = (1.0 - eps) * (double)policy[a];
            p = t + eps * noise[a];
            f64 = 1;
        } else {
            p = (double)policy[a];
        }
        parent->child[a] = node_new(a, p, f64, parent);
        parent->nchild++;
    }
}

/* PUCT score, play.py:308-323 (Cpuct = 1).  Returns the score as double (exact widening in the
 * float32 regime) so callers can compare with python-int sentinels. */
static double puct_score(const ONode *c, double total_n) {
    if (c->p_f64) {
        double u = c->p * total_n / (1. + (double)c->count);
        return (double)c->mean + u;
    } else {
        float u = (float)c->p * (float)total_n; /* float32 * python float -> float32 */
        u = u / (float)(1. + (double)c->count);
        float v = c->mean + u;
        return (double)v;
    }
}
static double children_total_n(const ONode *node, int A) {
    long sum = 0;
    for (int a = 0; a < A; a++)
        if (node->child[a]) sum += node->child[a]->count;
    double t = sqrt((double)sum);
    if (t == 0) t = 1;
    return t;
}
/* play.py:308-323: returns action or -1 for {} */
static int top_one_with_virtual_loss(const ONode *node, int A) {
    double total_n = children_total_n(node, A);
    double max_value = -100;
    int best = -1;
    for (int a = 0; a < A; a++) {
        const ONode *c = node->child[a];
        if (!c) continue;
        if (c->vloss > 0) continue;
        double v = puct_score(c, total_n);
        if (v > max_value) { max_value = v; best = a; }
    }
    return best;
}

/* tree_util.py:4-24.  Returns leaf or NULL; moves[] / *nmoves filled. */
static ONode *find_best_leaf_virtual_loss(ONode *node, int A, int *moves, int *nmoves) {
    int n = 0;
    while (node->child != NULL) {
        int a = top_one_with_virtual_loss(node, A);
        if (a < 0) {
            if (node->parent == NULL) { *nmoves = 0; return NULL; }
            node->vloss = 2;
            node = node->parent;
            n = n > 0 ? n - 1 : 0;
            continue;
        }
        node = node->child[a];
        moves[n++] = a;
    }
    node->vloss = 2;
    *nmoves = n;
    return node;
}

/* ---- unit-level entry points over flat child tables (for tests/golden/puct.npz) ---- */
static ONode *table_node(int A, const double *P, const int32_t *N, const float *Q, const int8_t *V,
                         const int8_t *EX, int f64) {
    ONode *root = node_new(-1, 1, 0, NULL);
    root->child = (ONode **)calloc(A, sizeof(ONode *));
    for (int a = 0; a < A; a++) {
        if (!EX[a]) continue;
        ONode *c = node_new(a, P[a], f64, root);
        c->count = N[a];
        c->mean = Q[a];
        c->vloss = V[a];
        root->child[a] = c;
        root->nchild++;
    }
    return root;
}
int ora_top_one_with_virtual_loss(int A, const double *P, const int32_t *N, const float *Q, const int8_t *V,
                                  const int8_t *EX, int f64) {
    ONode *r = table_node(A, P, N, Q, V, EX, f64);
    int a = top_one_with_virtual_loss(r, A);
    node_free(r, A);
    return a;
}
/* play.py:325-336 (sync path selector; sentinel value -1, no busy exclusion) */
int ora_top_one_action(int A, const double *P, const int32_t *N, const float *Q, const int8_t *EX, int f64) {
    int8_t V[MAXA] = {0};
    ONode *r = table_node(A, P, N, Q, V, EX, f64);
    double total_n = children_total_n(r, A);
    double maxv = -1;
    int best = -1;
    for (int a = 0; a < A; a++) {
        if (!r->child[a]) continue;
        double v = puct_score(r->child[a], total_n);
        if (v > maxv) { maxv = v; best = a; }
    }
    node_free(r, A);
    return best;
}
/* play.py:338-352: stable descending insertion, truncated to top_n */
int ora_top_n_actions(int A, const double *P, const int32_t *N, const float *Q, const int8_t *EX, int f64,
                      int top_n, int32_t *out) {
    int8_t V[MAXA] = {0};
    ONode *r = table_node(A, P, N, Q, V, EX, f64);
    double total_n = children_total_n(r, A);
    double vals[MAXA + 1];
    int acts[MAXA + 1];
    int len = 0;
    for (int a = 0; a < A; a++) {
        if (!r->child[a]) continue;
        double v = puct_score(r->child[a], total_n);
        if (len < top_n || v > vals[len - 1]) {
            int pos = len; /* append, then stable sort descending: lands after every element >= v */
            while (pos > 0 && vals[pos - 1] < v) pos--;
            for (int i = len; i > pos; i--) { vals[i] = vals[i - 1]; acts[i] = acts[i - 1]; }
            vals[pos] = v;
            acts[pos] = a;
            len++;
        }
        if (len > top_n) len--;
    }
    for (int i = 0; i < len; i++) out[i] = acts[i];
    node_free(r, A);
    return len;
}

/* ------------------------------------------------------------------ async self-play game --- */

#define MAXE 64
#define PH_ROOT 0  /* waiting for the root evaluation of the current position */
#define PH_LEAF 1  /* waiting for evaluations of launched leaves */
#define PH_DONE 2

typedef struct {
    ONode *leaf;
    int moves[2 * MAXN + 8];
    int nmoves;
    int32_t *board; /* leaf position (root board + replayed moves) */
    int evaluated;
    float *policy;
    float value;
} Pending;

typedef struct {
    int move_n, player, action, x, y;
    float value;
    int32_t *board;
    double *policy;
} MoveRec;

typedef struct OraGame {
    int S, A;
    double komi;
    int sims, energy, stop_exploration, num_moves;
    int self_play;
    double dir_eps;
    float resign; /* NaN = None */
    int has_resign;
    /* draws */
    const double *uniforms; int n_uniforms, i_uniform;
    const double *noises; int n_noises, i_noise;
    /* game state (play_game_async locals) */
    int32_t *board;
    int player;
    ONode *tree;
    float value, last_value;
    int has_value;
    int skipped_last, temperature, move_n;
    int end_reason; /* 0 PLAYED ALL MOVES, 1 resign, 2 BOTH_PASSED */
    /* search state */
    int phase;
    int rounds_left;
    int e_left, pre_bp;
    int original_player;
    Pending fifo[MAXE * 2];
    int fifo_head, fifo_tail;
    int need_bp; /* selection stopped on "no best leaf": back-propagate ONE result, then go on */
    /* outputs */
    MoveRec *recs; int n_recs, cap_recs;
    int winner, black_points; double white_points;
    long n_predict, n_root_predict, none_events;
    int error;
    int halt_at; /* test hook: stop right before the move choice of move halt_at (error 5) */
} OraGame;

static void pending_clear(Pending *p) {
    free(p->board);
    free(p->policy);
    memset(p, 0, sizeof(*p));
}

OraGame *ora_game_new(int S, double komi, int sims, int energy, int stop_exploration, int num_moves,
                      int self_play, double dir_eps) {
    if (S < 2 || S > MAXS || energy < 1 || energy > MAXE) return NULL;
    OraGame *g = (OraGame *)calloc(1, sizeof(OraGame));
    g->S = S; g->A = S * S + 1; g->komi = komi; g->sims = sims; g->energy = energy;
    g->stop_exploration = stop_exploration;
    g->num_moves = num_moves < 0 ? S * S * 2 : num_moves;
    g->self_play = self_play; g->dir_eps = dir_eps;
    g->board = (int32_t *)malloc(sizeof(int32_t) * S * S * NPL);
    ora_game_init(S, g->board);
    g->player = 1;
    g->temperature = 1;
    g->halt_at = -1;
    g->phase = PH_ROOT;
    /* loop head of move 0 (nomodel_self_play.py:161-164) */
    if (g->num_moves == 0) g->phase = PH_DONE;
    if (g->move_n == g->stop_exploration) g->temperature = 0;
    return g;
}
void ora_game_set_halt(OraGame *g, int move_n) { g->halt_at = move_n; }
/* `if resign and value <= resign` (nomodel_self_play.py:171): a threshold of 0.0 is falsy in Python = never resign */
void ora_game_set_resign(OraGame *g, int has, float thr) { g->has_resign = has && thr != 0.0f; g->resign = thr; }
void ora_game_set_draws(OraGame *g, const double *uniforms, int n_u, const double *noises, int n_n) {
    g->uniforms = uniforms; g->n_uniforms = n_u; g->i_uniform = 0;
    g->noises = noises; g->n_noises = n_n; g->i_noise = 0;
}
void ora_game_free(OraGame *g) {
    if (!g) return;
    for (int i = 0; i < MAXE * 2; i++) pending_clear(&g->fifo[i]);
    node_free(g->tree, g->A);
    for (int i = 0; i < g->n_recs; i++) { free(g->recs[i].board); free(g->recs[i].policy); }
    free(g->recs);
    free(g->board);
    free(g);
}

static void finish_game(OraGame *g) {
    g->winner = ora_get_winner(g->S, g->board, g->komi, &g->black_points, &g->white_points);
    g->phase = PH_DONE;
}

/* nomodel_self_play.py:40-56 with the graft of simulation_workers.py:42-54's result */
static void back_propagation(OraGame *g, Pending *r) {
    ONode *leaf = r->leaf;
    /* basic_tasks2: subtree, v, count/value/mean of the (pickled copy of the) leaf */
    new_subtree(g->S, r->policy, r->board, leaf, NULL, 0);
    float v = (r->board[16] == g->original_player) ? r->value : -r->value;
    leaf->count += 1;
    leaf->value += v;
    leaf->mean = leaf->value / (float)leaf->count;
    /* back_propagation */
    leaf->vloss = 0;
    ONode *cp = leaf->parent;
    while (1) {
        cp->count += 1;
        cp->value += leaf->value;
        cp->mean = cp->value / (float)cp->count;
        cp->vloss = 0;
        if (cp->parent) cp = cp->parent;
        else break;
    }
}

static Pending *fifo_push(OraGame *g) {
    Pending *p = &g->fifo[g->fifo_tail % (MAXE * 2)];
    g->fifo_tail++;
    return p;
}
static int fifo_size(const OraGame *g) { return g->fifo_tail - g->fifo_head; }
static Pending *fifo_front(OraGame *g) { return &g->fifo[g->fifo_head % (MAXE * 2)]; }
static void fifo_pop(OraGame *g) {
    pending_clear(fifo_front(g));
    g->fifo_head++;
}
static int fifo_unevaluated(const OraGame *g) {
    int n = 0;
    for (int i = g->fifo_head; i < g->fifo_tail; i++)
        if (!g->fifo[i % (MAXE * 2)].evaluated) n++;
    return n;
}

static void choose_and_play(OraGame *g);

/* async_simulate2 (nomodel_self_play.py:59-82), selection part.  Runs until energy is spent or a
 * result must be awaited.  Returns with phase PH_LEAF if evaluations are needed. */
static void run_search(OraGame *g) {
    int S = g->S, A = g->A;
    for (;;) {
        if (g->rounds_left == 0) { choose_and_play(g); return; }
        /* one async_simulate2 call */
        if (g->e_left < 0) { /* round not started */
            if (g->tree->child == NULL) { g->rounds_left--; continue; } /* subtree == {} -> return */
            g->e_left = g->energy;
            g->pre_bp = 0;
        }
        while (g->e_left > 0) {
            int moves[2 * MAXN + 8], nm = 0;
            ONode *leaf = find_best_leaf_virtual_loss(g->tree, A, moves, &nm);
            if (leaf != NULL && leaf->count > 0) { g->e_left--; g->pre_bp++; continue; }
            if (leaf == NULL) {
                g->none_events++;
                if (fifo_size(g) == 0) { g->error = 1; g->phase = PH_DONE; return; } /* reference would block forever */
                if (!fifo_front(g)->evaluated) { g->need_bp = 1; g->phase = PH_LEAF; return; }
                back_propagation(g, fifo_front(g));
                fifo_pop(g);
                g->pre_bp++;
                continue;
            }
            leaf->parent = leaf->parent; /* best_leaf['parent'] = None only severs the pickled copy */
            Pending *p = fifo_push(g);
            p->leaf = leaf;
            p->nmoves = nm;
            memcpy(p->moves, moves, sizeof(int) * nm);
            p->board = (int32_t *)malloc(sizeof(int32_t) * S * S * NPL);
            memcpy(p->board, g->board, sizeof(int32_t) * S * S * NPL);
            for (int i = 0; i < nm; i++) { /* basic_tasks2: replay from the root position */
                int m = moves[i];
                int y = m / S, x = m - S * y;
                ora_make_play(S, p->board, x, y, 0);
            }
            p->evaluated = 0;
            g->e_left--;
        }
        if (fifo_unevaluated(g) > 0) { g->phase = PH_LEAF; return; }
        /* for i in range(ENERGY - pre_bp): get(); back_propagation */
        int nbp = g->energy - g->pre_bp;
        for (int i = 0; i < nbp; i++) {
            if (fifo_size(g) == 0) { g->error = 2; g->phase = PH_DONE; return; }
            back_propagation(g, fifo_front(g));
            fifo_pop(g);
        }
        g->e_left = -1;
        g->rounds_left--;
    }
}

/* select_play tail (nomodel_self_play.py:125-138) + play_game_async body after it (:180-216) */
static void choose_and_play(OraGame *g) {
    int S = g->S, A = g->A;
    ONode *root = g->tree;
    int selected = -1;
    if (g->halt_at == g->move_n) { g->error = 5; g->phase = PH_DONE; return; }
    if (g->temperature == 1) {
        long total_n = 0;
        for (int a = 0; a < A; a++) if (root->child[a]) total_n += root->child[a]->count;
        int mv[MAXA]; double cdf[MAXA]; int n = 0;
        double acc = 0;
        for (int a = 0; a < A; a++) {
            ONode *c = root->child[a];
            if (!c || !c->count) continue;
            double p = (double)c->count / (double)total_n;
            acc += p; /* np.cumsum: sequential float64 */
            mv[n] = a; cdf[n] = acc; n++;
        }
        if (n == 0 || g->i_uniform >= g->n_uniforms) { g->error = 3; g->phase = PH_DONE; return; }
        double u = g->uniforms[g->i_uniform++];
        double last = cdf[n - 1];
        int idx = n; /* searchsorted(cdf/last, u, side='right') */
        for (int i = 0; i < n; i++) if (cdf[i] / last > u) { idx = i; break; }
        if (idx >= n) idx = n - 1;
        selected = mv[idx];
    } else {
        /* max((count, mean_value, a)) : lexicographic, ties -> higher mean, then higher index */
        int bc = -1; float bm = 0; int ba = -1;
        for (int a = 0; a < A; a++) {
            ONode *c = root->child[a];
            if (!c) continue;
            if (ba < 0 || c->count > bc || (c->count == bc && (c->mean > bm || (c->mean == bm && a > ba)))) {
                bc = c->count; bm = c->mean; ba = a;
            }
        }
        selected = ba;
    }
    int y = selected / S, x = selected - S * y;
    /* move_data */
    if (g->n_recs == g->cap_recs) {
        g->cap_recs = g->cap_recs ? g->cap_recs * 2 : 64;
        g->recs = (MoveRec *)realloc(g->recs, sizeof(MoveRec) * g->cap_recs);
    }
    MoveRec *r = &g->recs[g->n_recs++];
    r->move_n = g->move_n; r->player = g->player; r->action = selected; r->x = x; r->y = y;
    r->value = g->value;
    r->board = (int32_t *)malloc(sizeof(int32_t) * S * S * NPL);
    memcpy(r->board, g->board, sizeof(int32_t) * S * S * NPL);
    r->policy = (double *)calloc(A, sizeof(double));
    for (int a = 0; a < A; a++) if (root->child[a]) r->policy[a] = root->child[a]->p;

    if (g->skipped_last && y == S) { g->end_reason = 2; finish_game(g); return; }
    g->skipped_last = (y == S);
    /* re-root (self-play: mcts_tree and other_mcts are the same object) */
    ONode *nr = root->child[selected];
    root->child[selected] = NULL;
    node_free(root, A);
    nr->parent = NULL;
    g->tree = nr;
    g->player = ora_make_play(S, g->board, x, y, 0); /* board, player = make_play(...): the MOVER */
    /* next iteration of `for move_n in range(num_moves)` */
    g->move_n++;
    if (g->move_n >= g->num_moves) { g->end_reason = 0; finish_game(g); return; }
    g->last_value = g->value;
    if (g->move_n == g->stop_exploration) g->temperature = 0;
    g->phase = PH_ROOT;
}

int ora_game_phase(const OraGame *g) { return g->phase; }
int ora_game_error(const OraGame *g) { return g->error; }

/* Boards that need a network evaluation now.  Returns their number; boards_out gets n x [S][S][17]. */
int ora_game_pending(OraGame *g, int32_t *boards_out) {
    int sz = g->S * g->S * NPL;
    if (g->phase == PH_ROOT) {
        if (boards_out) memcpy(boards_out, g->board, sizeof(int32_t) * sz);
        return 1;
    }
    if (g->phase == PH_LEAF) {
        int n = 0;
        for (int i = g->fifo_head; i < g->fifo_tail; i++) {
            Pending *p = &g->fifo[i % (MAXE * 2)];
            if (p->evaluated) continue;
            if (boards_out) memcpy(boards_out + (size_t)n * sz, p->board, sizeof(int32_t) * sz);
            n++;
        }
        return n;
    }
    return 0;
}

/* Results for exactly the boards ora_game_pending() listed, same order. */
void ora_game_submit(OraGame *g, const float *policies, const float *values) {
    int S = g->S, A = g->A;
    if (g->phase == PH_ROOT) {
        g->n_predict++; g->n_root_predict++;
        g->value = values[0];
        g->has_value = 1;
        if (g->has_resign && g->value <= g->resign) { g->end_reason = 1; finish_game(g); return; }
        if (!g->tree || !g->tree->child) {
            node_free(g->tree, A);
            g->tree = node_new(-1, 1, 0, NULL);
            const double *noise = NULL;
            if (g->self_play) {
                if (g->i_noise >= g->n_noises) { g->error = 4; g->phase = PH_DONE; return; }
                noise = g->noises + (size_t)(g->i_noise++) * A;
            }
            new_subtree(S, policies, g->board, g->tree, noise, g->dir_eps);
        }
        g->rounds_left = g->sims / g->energy; /* int(MCTS_SIMULATIONS / ENERGY) */
        g->e_left = -1;
        g->original_player = g->board[16];
        run_search(g);
        return;
    }
    if (g->phase == PH_LEAF) {
        int n = 0;
        for (int i = g->fifo_head; i < g->fifo_tail; i++) {
            Pending *p = &g->fifo[i % (MAXE * 2)];
            if (p->evaluated) continue;
            p->policy = (float *)malloc(sizeof(float) * A);
            memcpy(p->policy, policies + (size_t)n * A, sizeof(float) * A);
            p->value = values[n];
            p->evaluated = 1;
            g->n_predict++;
            n++;
        }
        if (g->need_bp) {
            g->need_bp = 0;
            back_propagation(g, fifo_front(g));
            fifo_pop(g);
            g->pre_bp++;
        }
        run_search(g);
    }
}

/* ---- read-back ---- */
int ora_game_n_moves(const OraGame *g) { return g->n_recs; }
void ora_game_move(const OraGame *g, int i, int *action, int *player, float *value, int32_t *board, double *policy) {
    const MoveRec *r = &g->recs[i];
    if (action) *action = r->action;
    if (player) *player = r->player;
    if (value) *value = r->value;
    if (board) memcpy(board, r->board, sizeof(int32_t) * g->S * g->S * NPL);
    if (policy) memcpy(policy, r->policy, sizeof(double) * g->A);
}
void ora_game_result(const OraGame *g, int *winner, int *black, double *white, int *end_reason, int *last_player) {
    if (winner) *winner = g->winner;
    if (black) *black = g->black_points;
    if (white) *white = g->white_points;
    if (end_reason) *end_reason = g->end_reason;
    if (last_player) *last_player = g->player;
}
void ora_game_counters(const OraGame *g, long *n_predict, long *n_root, long *none_events) {
    if (n_predict) *n_predict = g->n_predict;
    if (n_root) *n_root = g->n_root_predict;
    if (none_events) *none_events = g->none_events;
}
void ora_game_board(const OraGame *g, int32_t *board) { memcpy(board, g->board, sizeof(int32_t) * g->S * g->S * NPL); }
int ora_game_move_n(const OraGame *g) { return g->move_n; }

/* root child table of the CURRENT tree */
void ora_game_root_table(const OraGame *g, int32_t *N, float *W, float *Q, double *P, int8_t *EX, int32_t *root_count,
                         float *root_value) {
    for (int a = 0; a < g->A; a++) {
        const ONode *c = (g->tree && g->tree->child) ? g->tree->child[a] : NULL;
        N[a] = c ? c->count : 0;
        W[a] = c ? c->value : 0;
        Q[a] = c ? c->mean : 0;
        P[a] = c ? c->p : 0;
        EX[a] = c ? 1 : 0;
    }
    if (root_count) *root_count = g->tree ? g->tree->count : 0;
    if (root_value) *root_value = g->tree ? g->tree->value : 0;
}

/* Canonical serialisation (same record as gen_golden.py:_tree_hash): pre-order, ascending action,
 * per child  <i action, i count, f value, f mean, d p, i vloss, i expanded>  = 32 bytes. */
static size_t ser_rec(const ONode *n, int A, uint8_t *buf, size_t cap, size_t off, long *nn, long *ne) {
    for (int a = 0; a < A; a++) {
        const ONode *c = n->child[a];
        if (!c) continue;
        if (buf && off + 32 <= cap) {
            int32_t i32;
            i32 = a; memcpy(buf + off, &i32, 4);
            i32 = c->count; memcpy(buf + off + 4, &i32, 4);
            memcpy(buf + off + 8, &c->value, 4);
            memcpy(buf + off + 12, &c->mean, 4);
            memcpy(buf + off + 16, &c->p, 8);
            i32 = c->vloss; memcpy(buf + off + 24, &i32, 4);
            i32 = c->child ? 1 : 0; memcpy(buf + off + 28, &i32, 4);
        }
        off += 32;
        (*nn)++;
        if (c->child) { (*ne)++; off = ser_rec(c, A, buf, cap, off, nn, ne); }
    }
    return off;
}
size_t ora_game_tree_serialize(const OraGame *g, uint8_t *buf, size_t cap, long *n_nodes, long *n_expanded) {
    long nn = 0, ne = 0;
    size_t off = 0;
    if (g->tree && g->tree->child) off = ser_rec(g->tree, g->A, buf, cap, 0, &nn, &ne);
    if (n_nodes) *n_nodes = nn;
    if (n_expanded) *n_expanded = ne;
    return off;
}
