/*
 * sgo.h -- C ABI of libsgo_hip.so, the MI355X (gfx950) self-play hot path of sejonggo.
 *
 * The reference (drsagitn/sejonggo) is pure Python and has no FFI; its boundary for this path is a
 * set of Python module symbols.  This header is the C ABI placed underneath those
 * symbols: plain pointers and sizes, no torch types, `int` return 0 = ok / negative = error (text via
 * sgo_last_error()).  Each entry point cites the reference interface it replaces (file:line relative
 * to the reference repo).  INTEGRATION.md shows the ctypes binding a maintainer adds on the
 * reference side.
 *
 * Conventions
 *   board17   int32 [n][S][S][17]  the reference's board tensor (play.py:295-299), NHWC, plane 2k =
 *             to-play side's stones k plies ago, 2k+1 = opponent's, plane 16 = to-play colour (+1/-1).
 *   action    a = y*S + x, pass = S*S                                   (play.py:31-37)
 *   packed    uint32 [n][sgo_packed_words(S)]  16 bit-planes of ceil(S*S/32) words, ABSOLUTE colours: bit a of plane
 *             2k = black stone at a, k plies ago; plane 2k+1 = white (board17's planes are relative to the side to
 *             move: relative plane c = absolute plane c ^ [white to play]).  The to-play flag is the top bit of the
 *             last word of plane 0 (1 = white to play).  19x19: 192 words = 768 B.
 *   legal     uint32 [n][sgo_plane_words(S)]   bit a = 1 <=> action a is LEGAL (pass bit always 1).
 *   *_dev     arguments are DEVICE pointers; `stream` is a hipStream_t passed as void*.
 *   host entry points (no _dev suffix) copy caller HOST buffers to the GPU, run the same kernels and
 *   copy back; they exist for drop-in use and parity tests, not for throughput.
 *   Supported board sizes: 5, 7, 9, 13, 19.
 *
 * There is no CPU fallback anywhere in this library: without a HIP device every call fails.
 */
#ifndef SGO_H
#define SGO_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SGO_OK 0
#define SGO_ERR_ARG (-1)          /* bad size / null pointer / unsupported board size */
#define SGO_ERR_HIP (-2)          /* HIP runtime error, see sgo_last_error() */
#define SGO_ERR_UNSUPPORTED (-3)  /* no hand-written kernel for this shape; the caller takes another route */
#define SGO_ERR_OCCUPIED (-101)   /* play.py:233-234 assert: stone on an occupied point */
#define SGO_ERR_RANGE (-102)      /* coordinates outside the board (IndexError in the reference) */
#define SGO_ERR_CAPACITY (-201)   /* a game's tree-block pool is exhausted */
#define SGO_ERR_DRAWS (-202)      /* ran out of injected random draws */
#define SGO_ERR_STATE (-203)      /* call sequence violated */

/* Bumped whenever a signature or a struct layout below changes.  A binding compares it with sgo_version() of the
 * library it loaded and refuses a mismatch (sejonggo_amd/_lib.py load(); INTEGRATION.md §B does the same).
 *   1: round 1.   2: sgo_start_games(+stream), sgo_game_result.first_model (40 bytes), sgo_config.two_model,
 *   sgo_conv_backend removed.   3: this round's additions (see the "half-populations" and "packed stem" sections). */
#define SGO_ABI_VERSION 3

const char *sgo_last_error(void);
int sgo_version(void);            /* SGO_ABI_VERSION the library was built with */
int sgo_device_count(void);
int sgo_set_device(int device_id);

/* ---- geometry ---------------------------------------------------------------------------------- */
int sgo_plane_words(int S);   /* ceil(S*S/32); also the number of words of a legal bitset */
int sgo_packed_words(int S);  /* words per packed position record */
int sgo_apad(int S);          /* child slots per tree block (= 32 * plane_words) */

/* ---- stateless rules on HOST buffers (drop-in for play.py) -------------------------------------- */
/* play.py:295-299 game_init */
int sgo_game_init(int S, int n, int32_t *board17);
/* play.py:226-242 make_play (+ :182-217 take_stones, :159-180 capture_group, :219-224 swap_player).
 * colors[i] = 0 means "color=None".  movers[i] receives the player who moved; status[i] is SGO_OK or
 * SGO_ERR_OCCUPIED / SGO_ERR_RANGE (board i is then left untouched).  Returns SGO_OK if the batch ran. */
int sgo_make_play(int S, int n, int32_t *board17, const int32_t *xs, const int32_t *ys, const int32_t *colors,
                  int32_t *movers, int32_t *status);
/* play.py:71-104 legal_moves: mask[n][S*S+1], 1 = ILLEGAL (the reference's convention), pass = 0 */
int sgo_legal_moves(int S, int n, const int32_t *board17, uint8_t *mask);
/* play.py:274-292 get_winner/_get_points: winner +1/0/-1, black points, white points (incl. komi) */
int sgo_get_winner(int S, int n, const int32_t *board17, double komi, int32_t *winner, int32_t *black, double *white);
/* play.py:182-217 take_stones on board tensors in place (planes 0/1 only): removes the opponent groups next to (x, y)
 * that have no liberty, then the own groups among (x, y) and its four neighbours that have none (suicide executed). */
int sgo_take_stones(int S, int n, int32_t *board17, const int32_t *xs, const int32_t *ys);
/* Group / territory queries on plain boards (play.py:159-180 capture_group, :55-69 get_liberties, :244-271 color_board).
 * cells int8 [n][S][S]: +1 black, -1 white, 0 empty, any other value = wall (neither stone nor liberty).
 * mode 0: member[i] = the seed (xs[i], ys[i]) plus the stones of colour colors[i] connected to it; liberty[i] = empty
 *         points next to a member.  mode 1: member[i] = empty points connected through empty points to a stone of
 *         colour colors[i] (color_board's fill); liberty is zeroed.  member / liberty: uint8 [n][S][S]. */
int sgo_board_query(int S, int n, int mode, const int8_t *cells, const int32_t *xs, const int32_t *ys, const int32_t *colors,
                    uint8_t *member, uint8_t *liberty);
/* symmetry.py:45-114 board transforms; k: 0 id, 1 left_diagonal, 2 vertical_axis, 3 horizontal_axis,
 * 4 rotation_90, 5 rotation_180, 6 rotation_270 (order of symmetry.SYMMETRIES, :117-125), 7 right_diagonal */
int sgo_sym_apply(int S, int k, int n, const int32_t *in17, int32_t *out17);
/* symmetry.py reverse_*: out[i][a] = in[i][SWAP_k[a]] */
int sgo_sym_invert_policy(int S, int k, int n, const float *in, float *out);
/* symmetry.py:12-42 the SWAP table itself (A entries) */
int sgo_sym_lut(int S, int k, int32_t *lut);

/* ---- device-resident batch kernels (the data-parallel hot path) --------------------------------- */
int sgo_pack_dev(int S, int n, const int32_t *d_board17, uint32_t *d_packed, void *stream);
int sgo_unpack_dev(int S, int n, const uint32_t *d_packed, int32_t *d_board17, void *stream);
/* board_advance: for i<n  out[out_idx?out_idx[i]:i] = make_play(in[in_idx?in_idx[i]:i], moves[i]) and
 * legal[...] = legal bits of the new position.  moves[i] = action index; colors may be NULL (None);
 * status may be NULL.  in and out may alias record-for-record (in place). */
int sgo_advance_legal_dev(int S, int n, const uint32_t *d_in, const int32_t *d_in_idx, const int32_t *d_moves,
                          const int32_t *d_colors, uint32_t *d_out, const int32_t *d_out_idx, uint32_t *d_legal,
                          int32_t *d_status, void *stream);
/* Kernel form behind the non-aliasing board_advance launches (sgo_advance_legal_dev with disjoint dense in / out, the
 * engine's leaf step): 0 = history stream + one lane per position (two launches), 1 = the same two in one launch,
 * 2 = one half-wavefront per position (row per lane; csrc/sgo_rows.hpp), -1 = by batch size (default: 2 up to 32 768
 * positions, 1 up to 65 536, 0 above).  Returns the previous mode; other values only query.  All forms are bit-identical. */
int sgo_advance_mode(int mode);
int sgo_legal_dev(int S, int n, const uint32_t *d_packed, const int32_t *d_idx, uint32_t *d_legal, void *stream);
/* d_result[i] = {winner, black, white_stones_and_territory (without komi)} as 3 x int32 */
int sgo_score_dev(int S, int n, const uint32_t *d_packed, const int32_t *d_idx, double komi, int32_t *d_result,
                  void *stream);
/* nn_input_pack (+ fused sym_apply): network input for positions d_packed[d_idx[i]], transformed by
 * symmetry k.  layout 0: NHWC [n][S][S][17], 1: NCHW [n][17][S][S], 2: NHWC with the channels zero-padded
 * to 32 [n][S][S][32] (MFMA-friendly K for the stem convolution); dtype 0: fp16, 1: fp32. */
int sgo_nn_pack_dev(int S, int n, const uint32_t *d_packed, const int32_t *d_idx, int k, int layout, int dtype,
                    void *d_out, void *stream);

/* Fused convolution epilogue of the resident net (model.py:37-46 BatchNorm folded into the conv, then
 * Activation('relu') / Add()): out = relu(x + bias[c] (+ skip)) on NHWC fp16, in place allowed.
 * n_elems and channels must be multiples of 8. */
int sgo_bias_act_dev(long n_elems, int channels, const void *d_x, const void *d_bias, const void *d_skip, void *d_out,
                     void *stream);

/* The 3x3 convolutions of the resident net with the epilogue fused: y = relu(conv3x3(x, w) + bias[k] (+ skip)),
 * stride 1, pad 0 or 1, NHWC fp16 (x [n][h][w][c], w [k][3][3][c] = PyTorch channels_last weight storage,
 * y / skip [n][ho][wo][k]), fp32 accumulation on MFMA.  Dispatches to the two hand-written kernels below; any other
 * shape returns SGO_ERR_UNSUPPORTED (net.FusedInferenceNet then runs that layer through the framework's convolution and
 * sgo_bias_act_dev). */
int sgo_conv3x3_bias_act_dev(int n, int h, int w, int c, int k, int pad, const void *d_x, const void *d_w,
                             const void *d_bias, const void *d_skip, void *d_y, void *stream);

/* The hand-written CDNA4 kernel for the tower shape (c = k = 256, pad 1, w <= 19; csrc/sgo_conv4w.hpp, sgo_conv8w.hpp).  Same layouts. */
int sgo_conv3x3_tower_dev(int n, int h, int w, const void *d_x, const void *d_w, const void *d_bias, const void *d_skip,
                          void *d_y, void *stream);
/* The same convolution fed from a filter bank in MFMA-FRAGMENT ORDER (csrc/sgo_conv4r.hpp, k_conv4r: the weights go L2 -> registers,
 * one 1-KB contiguous load per fragment, and never touch the LDS; the resident net's route since round 3).  A bank is
 * sgo_conv3x3_tower_packed_bytes() bytes, 16-byte aligned, and is written ONCE per layer from the OHWI weights d_w
 * [256][3][3][256] fp16 by sgo_conv3x3_tower_prepack_dev (again after every weight update); x / bias / skip / y as above.
 * Results equal sgo_conv3x3_tower_dev's bit for bit (same MFMA order per output). */
long sgo_conv3x3_tower_packed_bytes(void);
int sgo_conv3x3_tower_prepack_dev(const void *d_w, void *d_wp, void *stream);
int sgo_conv3x3_tower_packed_dev(int n, int h, int w, const void *d_x, const void *d_wp, const void *d_bias, const void *d_skip,
                                 void *d_y, void *stream);
/* Schedule variant of k_conv4r (A/B builds with -DSGO_CONV4W_VARIANTS; 1 = the product).  Returns the previous one; negative = query. */
int sgo_conv_packed_variant(int v);
/* The hand-written CDNA4 kernel for the stem (c = 32: the 17 input planes zero-padded, k = 256, pad 0, no skip;
 * csrc/sgo_stem.hpp; model.py:57-60).  x [n][h][w][32] is layout 2 of sgo_nn_pack_dev: the route of callers that hold board
 * TENSORS (put_predict_request); the self-play engine feeds the net through sgo_stem_packed_dev below instead. */
int sgo_conv3x3_stem_dev(int n, int h, int w, const void *d_x, const void *d_w, const void *d_bias, void *d_y, void *stream);
/* The stem straight from PACKED POSITION RECORDS (csrc/sgo_stem_packed.hpp; model.py:57-60 + symmetry.py:127-132): row i of the
 * output is relu(conv3x3_valid(planes of sym_k(record d_index[i])) + bias), y [n][S-2][S-2][256] fp16.  The 16 stone planes are
 * expanded from the record's bit-planes in LDS (relative to the side to move, symmetry on the gather side); the colour plane
 * (+-1 over the whole board, 'valid' convolution) is the per-position constant c * d_wcol[k] added to the bias.
 * d_w10: [256][10][16] fp16 -- taps 0..8 (dy * 3 + dx) of the 16 stone planes, tap 9 zero; d_bias fp16 [256];
 * d_wcol float [256] = sum over the 9 taps of the colour plane's weights.  d_index NULL = records 0..n-1; d_sym_k (device
 * int, optional) overrides sym_k when the kernel runs.  No network-input tensor exists on this route. */
int sgo_stem_packed_dev(int S, int n, const uint32_t *d_records, const int32_t *d_index, int sym_k, const int32_t *d_sym_k,
                        const void *d_w10, const void *d_bias, const float *d_wcol, void *d_y, void *stream);
/* Which hand-written kernel sgo_conv3x3_tower_dev launches: 1 = k_conv4w (csrc/sgo_conv4w.hpp: two 256-thread workgroups per
 * CU, 256 pixels x 128 channels each; the default), 0 = k_conv8w (csrc/sgo_conv8w.hpp: one 512-thread workgroup per CU,
 * 256 pixels x 256 channels).  Same results bit for bit.  Returns the previous choice; other values only query. */
int sgo_conv_tower_kernel(int mode);
/* Tile order of the tower kernel's launches: 1 = every XCD walks a contiguous range of pixel tiles (default: the halo rows
 * a tile shares with its neighbour are then in that XCD's L2), 0 = identity.  Returns the previous mode; other values query. */
int sgo_conv_tile_order(int mode);
/* Test hook: cap the samples per launch of sgo_conv3x3_tower_dev / _stem_dev (0 = no cap) so that the slice loop, which
 * otherwise needs tensors beyond 2^31 bytes, can be exercised on small inputs.  Returns the previous cap; negative = query. */
long sgo_conv_tower_slice_cap(long cap);

/* ---- self-play engine: virtual-loss PUCT + game loop, many games resident on one GPU ------------ */
/* Replaces nomodel_self_play.py:59-82 async_simulate2, :114-140 select_play, :142-271 play_game_async,
 * tree_util.py:4-32, play.py:308-323/376-421, simulation_workers.py:42-54 basic_tasks2 and the request
 * side of predicting_queue_worker.py:40-102.  One ctx per GPU; not thread-safe. */
typedef struct sgo_ctx sgo_ctx;

typedef struct sgo_config {
    int32_t size;             /* conf['SIZE'] */
    int32_t n_games;          /* concurrent game slots */
    int32_t sims;             /* conf['MCTS_SIMULATIONS'] */
    int32_t energy;           /* conf['ENERGY'] (<= 64) */
    int32_t stop_exploration; /* conf['STOP_EXPLORATION'] */
    int32_t num_moves;        /* play_game_async(num_moves); <0 => 2*S*S */
    int32_t blocks_per_game;  /* PRIVATE tree blocks per game; <=0 => 8*sims + 128 (and the default shared pool, below) */
    int32_t self_play;        /* add Dirichlet noise when a tree is created (play.py:400-403) */
    double komi;              /* conf['KOMI'] */
    double dirichlet_epsilon; /* conf['DIRICHLET_EPSILON'] */
    int32_t device_id;
    int32_t two_model;        /* 1: evaluation games between two nets (evaluate_worker.py:137): one tree per player, the tree of
                                 the side not to move follows the move when it holds it (nomodel_self_play.py:203-218);
                                 requires self_play = 0.  Every row of a step's evaluation list belongs to one model:
                                 sgo_eval_models() */
    int32_t shared_blocks;    /* tree blocks SHARED by all games of the context: a game whose tree outgrows its private blocks
                                 takes more from here, one at a time, and gives them back when a move prunes its tree.
                                 > 0: that many; < 0: the default (2*sims per game, at least 12*sims + 128); 0: the default when
                                 blocks_per_game <= 0, none otherwise (a fixed per-game pool).  All bounded by 60 % of the free
                                 device memory and, per game, by the id space of k_search's LDS work queue (~38 000 blocks). */
    int32_t reserved;
} sgo_config;

typedef struct sgo_status {
    int32_t n_eval;       /* positions waiting for a network evaluation after this step */
    int32_t n_records;    /* move records waiting in the record buffer */
    int32_t n_active;     /* game slots still playing */
    int32_t n_done;       /* game slots finished and not yet restarted */
    int32_t error;        /* first error raised by any game (SGO_ERR_*) or 0 */
    int32_t error_game;
    int64_t total_moves;  /* moves played since ctx creation */
    int64_t total_evals;  /* network evaluations consumed since ctx creation */
    int64_t none_events;  /* "No best leaf" events (nomodel_self_play.py:70-75) */
} sgo_status;

/* one record per move played = the reference's move_data (nomodel_self_play.py:187-194) */
typedef struct sgo_move_record {
    int32_t game;      /* slot */
    int32_t game_seq;  /* how many games this slot had finished before this one */
    int32_t move_n;
    int32_t action;
    int32_t player;    /* 'player' exactly as the reference records it */
    float value;       /* raw network value at the root */
} sgo_move_record;

typedef struct sgo_game_result {
    int32_t winner;      /* +1 black, -1 white, 0 draw (play.py:274-284) */
    int32_t black;       /* black points */
    double white;        /* white points incl. komi */
    int32_t end_reason;  /* 0 PLAYED ALL MOVES, 1 resign, 2 BOTH_PASSED */
    int32_t n_moves;
    int32_t last_player; /* 'player' when the loop ended (used for "X+R") */
    int32_t done;
    int32_t first_model; /* two_model games: which model moved first = plays black (0 = model1, 1 = model2) */
    int32_t blocks_high_water; /* most tree blocks the game ever held at once (private + shared) */
} sgo_game_result;

sgo_ctx *sgo_ctx_create(const sgo_config *cfg);
void sgo_ctx_destroy(sgo_ctx *ctx);
int sgo_blocks_per_game(sgo_ctx *ctx);   /* the private tree blocks per game this context was created with */
/* out[0..n): {private blocks per game, local ids per game (private + overflow ids), shared pool blocks, shared blocks free now,
 * fewest shared blocks ever free, games}.  n <= 6.  Synchronises the device. */
int sgo_pool_info(sgo_ctx *ctx, int64_t *out, int n);
/* (Re)start game slots.  noise: [n][A] float64 Dirichlet draws (np.random.dirichlet stand-in, consumed
 * when a tree is created); uniforms: [n][n_uniforms] float64 in [0,1) consumed one per sampled move
 * (np.random.choice stand-in); resign: [n] thresholds, NaN or 0 = None.  HOST pointers; they are copied before the call
 * returns.  The batch reaches the device in ONE host-to-device copy followed by one kernel, both queued on `stream`
 * (use the stream the steps run on): no device-wide synchronisation. */
int sgo_start_games(sgo_ctx *ctx, int n, const int32_t *slots, const double *noise, const double *uniforms,
                    int n_uniforms, const float *resign, void *stream);
/* (Re)start slots of a two_model context: no Dirichlet noise (self_play is off), one resign threshold per model
 * (nomodel_self_play.py:170: `resign_model1 if current == model1 else resign_model2`), and who moves first
 * (first_model[i] = 0: model1 plays black; play.py:301-306 choose_first_player is the caller's coin). */
int sgo_start_games2(sgo_ctx *ctx, int n, const int32_t *slots, const double *uniforms, int n_uniforms,
                     const float *resign_model1, const float *resign_model2, const int32_t *first_model, void *stream);
/* Which model (0 / 1) must evaluate each row of the evaluation list of the last step (all 0 unless two_model).  HOST buffer
 * models[cap]; returns the number of rows. */
int sgo_eval_models(sgo_ctx *ctx, int cap, int32_t *models);
/* One engine step.  Consumes the evaluations of the positions listed by the previous step
 * (d_policy [n_eval][A] float32, d_value [n_eval] float32, produced from inputs transformed by
 * symmetry sym_k; NULL on the first call), back-propagates, selects the next leaves, plays moves whose
 * search is complete, computes the new leaf positions, and reports what must be evaluated next.
 * Synchronises `stream` once to return `st`. */
int sgo_step(sgo_ctx *ctx, const float *d_policy, const float *d_value, int sym_k, void *stream, sgo_status *st);
/* The same step in two halves, for launch chains that must not wait for the host (hipGraph capture, two half-populations
 * alternating on two streams): sgo_step_enqueue queues k_search / k_compact / board_advance and the copy of the status words to
 * pinned host memory on `stream` and returns at once; the symmetry the consumed evaluations were produced under is read from
 * DEVICE memory (*d_sym_k, 0..7) when the kernel runs, so one captured chain serves every symmetry.  d_policy / d_value must
 * stay valid until the chain has run (rows beyond the listed count are ignored).  After the caller has synchronised the stream
 * (or an event behind the enqueue), sgo_step_status returns what that step reported.  No board_advance timing in this form. */
int sgo_step_enqueue(sgo_ctx *ctx, const float *d_policy, const float *d_value, const int32_t *d_sym_k, void *stream);
int sgo_step_status(sgo_ctx *ctx, sgo_status *st);
/* Where the evaluation list of the last step lives ON THE DEVICE, for consumers that read packed records directly
 * (sgo_stem_packed_dev): *d_records = the context's record array (sgo_packed_words(S) words per record), *d_index = the
 * record index of every row of the list (n_eval of them valid, in the order results are expected), *d_models = which model
 * evaluates each row (two_model contexts).  The pointers stay valid for the life of the context; the contents change with
 * every step.  Returns the capacity of the list (n_games * energy). */
int sgo_eval_list(sgo_ctx *ctx, const uint32_t **d_records, const int32_t **d_index, const int32_t **d_models);
/* Network input for the positions listed by the last sgo_step (same order as the results expected). */
int sgo_collect(sgo_ctx *ctx, int sym_k, int layout, int dtype, void *d_nn_in, void *stream);
/* Move records produced so far (HOST buffers): recs[cap], boards packed [cap][packed_words],
 * policy targets [cap][A] float64.  Returns the number written (>=0) and clears the buffer. */
int sgo_drain_records(sgo_ctx *ctx, int cap, sgo_move_record *recs, uint32_t *packed, double *policy);
int sgo_game_results(sgo_ctx *ctx, int n, const int32_t *slots, sgo_game_result *out);
/* Introspection for parity tests: root child table of a slot's current tree and the canonical
 * serialisation of the whole tree (32-byte records, see oracle/sgo_oracle.c ora_game_tree_serialize). */
int sgo_root_table(sgo_ctx *ctx, int slot, int32_t *N, float *W, float *Q, double *P, int8_t *EX, int32_t *root_count,
                   float *root_value);
int64_t sgo_tree_serialize(sgo_ctx *ctx, int slot, uint8_t *buf, int64_t cap, int64_t *n_nodes, int64_t *n_expanded);
/* Same walk with 40-byte records <i action, i count, f value, f mean, d p, i vloss, i expanded, i depth, i pad>, from
 * which a host rebuilds the reference's nested dict nodes (play.py:376-421) -- see engine.SelfPlayEngine.tree_dict. */
int64_t sgo_tree_dump(sgo_ctx *ctx, int slot, uint8_t *buf, int64_t cap, int64_t *n_nodes);
int sgo_game_board(sgo_ctx *ctx, int slot, int32_t *board17);
/* test hook: stop slot right before the move choice of move_n == k (phase becomes done, error 0) */
int sgo_set_halt(sgo_ctx *ctx, int slot, int move_n);
/* Diagnostic: cycles per phase of k_search, summed over games and calls (zeros unless the library was built with
 * -DSGO_KSEARCH_PROFILE): [0] consuming evaluations, [2] selection, [7] round back-propagation, [3] move step, [4] wave-calls. */
int sgo_debug_counters(sgo_ctx *ctx, unsigned long long *out, int n);

/* average duration (ms) and launch count of the board_advance kernel inside sgo_step since the last
 * call (HIP events on the step's stream); used by bench.py for the roofline object */
int sgo_advance_timing(sgo_ctx *ctx, double *total_ms, int64_t *launches, int64_t *positions);

#ifdef __cplusplus
}
#endif
#endif
