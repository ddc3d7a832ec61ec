"""Host driver of the MI355X self-play engine (libsgo_hip.so, include/sgo.h "self-play engine").

`SelfPlayEngine` keeps `n_games` games resident on one GPU and advances all of them in lock-free
steps: collect the positions that need a network evaluation -> one batched forward pass of the
resident policy/value net -> `sgo_step` (back-propagation, virtual-loss PUCT selection, move choice,
re-rooting, board_advance for the new leaves, all on the GPU).  It is what sits underneath the
reference-shaped entry points in nomodel_self_play.py / selfplay_worker.py of this package, and it
replaces, for many games at once, the reference's process zoo around one game
(nomodel_self_play.py:142-271 play_game_async + simulation_workers.py Pool + predicting_queue_worker.py).

The net is any object with the reference's model contract (model.py:57,80,90): `.name` and
`.predict_on_batch(X[n,S,S,17]) -> [policy [n,S*S+1] (float32), value [n,1] (float32)]`, here on CUDA
tensors.  PyTorch is plumbing (device memory, stream, the net itself); the tree and the rules are HIP.
"""
import ctypes as C
import random as pyrandom

import numpy as np

from . import _lib
from .conf import conf

END_REASONS = {0: "PLAYED ALL MOVES", 1: "resign", 2: "BOTH_PASSED"}


def unpack_positions(packed, size):
    """packed uint32 [n, RW] -> the reference's board tensor int32 [n, S, S, 17] (host-side format conversion for
    move records; the hot path never leaves the packed form).  Record planes are absolute (2k black, 2k+1 white,
    k plies ago) with the to-play bit in the top bit of plane 0's last word; the tensor's planes are relative to the
    side to move (include/sgo.h)."""
    packed = np.ascontiguousarray(packed, dtype=np.uint32)
    n = packed.shape[0]
    N = size * size
    NW = (N + 31) // 32
    planes = packed[:, :16 * NW].reshape(n, 16, NW)
    white = ((planes[:, 0, NW - 1] >> np.uint32(31)) & np.uint32(1)).astype(bool)
    bits = np.unpackbits(planes.view(np.uint8).reshape(n, 16, NW * 4), axis=2, bitorder="little")[:, :, :N]
    rel = np.where(white[:, None, None], bits[:, np.arange(16) ^ 1, :], bits)       # relative plane c = absolute c ^ white
    boards = np.zeros((n, size, size, 17), dtype=np.int32)
    boards[:, :, :, :16] = rel.transpose(0, 2, 1).reshape(n, size, size, 16)
    boards[:, :, :, 16] = np.where(white, -1, 1).astype(np.int32)[:, None, None]
    return boards


class MoveRecord(dict):
    """One move_data dict of the reference (nomodel_self_play.py:187-194: board, policy, value, move, move_n, player).
    The position is kept in its packed 768-byte form; `record['board']` expands it to the reference's int32
    [1,S,S,17] tensor on access (a resident 19x19 board tensor is 24.5 KB -- times ~300 moves times 1 024 games in
    flight that would be 7.5 GB of host memory for data that is written out once)."""
    __slots__ = ("size",)

    def __missing__(self, key):
        if key == 'board':
            return unpack_positions(self['packed'][None], self.size)
        raise KeyError(key)


class SelfPlayEngine(object):
    def __init__(self, net, size=None, n_games=None, sims=None, energy=None, stop_exploration=None, num_moves=None,
                 komi=None, self_play=True, dirichlet_alpha=None, dirichlet_epsilon=None, blocks_per_game=0,
                 device=0, symmetry="random1", layout="nhwc", dtype="fp16", seed=0, raise_on_error=True, packed=True,
                 net2=None, graph=False, stream=None, shared_blocks=0):
        """net2: a second net turns the slots into two-model EVALUATION games (evaluate_worker.py:137: model1 = `net`,
        model2 = `net2`, one tree per player, no Dirichlet noise); start them with start_eval_games."""
        import torch
        self.torch = torch
        self.lib = _lib.require_gpu()
        self.net = net
        self.net2 = net2
        self.two_model = net2 is not None
        if self.two_model:
            self_play = False
        self.S = size or conf['SIZE']
        self.A = self.S * self.S + 1
        self.G = n_games or conf['GAMES_PER_GPU']
        self.sims = conf['MCTS_SIMULATIONS'] if sims is None else sims
        self.E = conf['ENERGY'] if energy is None else energy
        self.stop_exploration = conf['STOP_EXPLORATION'] if stop_exploration is None else stop_exploration
        self.num_moves = num_moves
        self.max_moves = 2 * self.S * self.S if num_moves is None else num_moves
        self.komi = conf['KOMI'] if komi is None else komi
        self.alpha = conf['DIRICHLET_ALPHA'] if dirichlet_alpha is None else dirichlet_alpha
        self.symmetry = symmetry
        assert symmetry in ("identity", "random1", "avg8") or symmetry in range(8)  # int k: always that symmetry
        if layout == "nhwc" and getattr(net, "in_channels", 17) == 32:
            layout = "nhwc32"          # the fused inference net takes the channel-padded input
        self.layout = {"nhwc": 0, "nchw": 1, "nhwc32": 2}[layout]
        self.dtype = {"fp16": 0, "fp32": 1}[dtype]
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        cfg = _lib.Config(size=self.S, n_games=self.G, sims=self.sims, energy=self.E,
                          stop_exploration=self.stop_exploration, num_moves=-1 if num_moves is None else num_moves,
                          blocks_per_game=blocks_per_game, self_play=1 if self_play else 0, komi=self.komi,
                          dirichlet_epsilon=conf['DIRICHLET_EPSILON'] if dirichlet_epsilon is None else dirichlet_epsilon,
                          device_id=device, two_model=1 if self.two_model else 0, shared_blocks=int(shared_blocks))
        self.ctx = C.c_void_p(self.lib.sgo_ctx_create(C.byref(cfg)))
        if not self.ctx:
            raise _lib.SgoError("sgo_ctx_create failed: %s" % self.lib.sgo_last_error().decode())
        self.RW = self.lib.sgo_packed_words(self.S)
        self.rng = np.random.RandomState(seed)
        self.pyrng = pyrandom.Random(seed)
        max_eval = self.G * self.E
        tdt = torch.float16 if self.dtype == 0 else torch.float32
        shape = {0: (max_eval, self.S, self.S, 17), 1: (max_eval, 17, self.S, self.S), 2: (max_eval, self.S, self.S, 32)}[self.layout]
        # The resident net reads the evaluation list as PACKED RECORDS (net.FusedInferenceNet.predict_packed -> sgo_stem_packed_dev:
        # the stem kernel expands the bit-planes in LDS), so no network-input tensor exists on that route; nets that take board
        # tensors (stub nets, plain torch modules, fp32) get them from sgo_collect (k_nn_pack) into nn_in.
        nets = [net] + ([net2] if net2 is not None else [])
        self.packed = bool(packed) and self.dtype == 0 and all(getattr(m, "packed_ok", False) for m in nets)
        self.nn_in = None if self.packed else torch.zeros(shape, dtype=tdt, device=self.device)
        rec, idx, mod = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _lib.check(self.lib.sgo_eval_list(self.ctx, C.byref(rec), C.byref(idx), C.byref(mod)), "sgo_eval_list")
        self._rec_ptr, self._idx_ptr = rec.value, idx.value
        self.status = _lib.Status()
        self._policy = None
        self._value = None
        self._luts = None
        self.records = {}          # slot -> list of move dicts of the game in progress
        self.game_ids = {}         # slot -> caller-supplied id
        self.n_steps = 0
        self.n_net_calls = 0
        self.n_net_positions = 0
        self.n_model_positions = [0, 0]    # two-model games: positions evaluated by model1 / model2
        self._primed = False
        self.raise_on_error = raise_on_error   # False: a failing slot (e.g. block pool exhausted) is left to the caller
        # Captured rounds (hipGraph): one replay = stem from the listed records -> tower -> heads -> k_search -> k_compact ->
        # board_advance -> status copy, with no host launch in between; graphs are captured lazily per padded batch size and
        # share one memory pool.  Needs the packed route, one model and one forward pass per list.
        self.graph = bool(graph) and self.packed and not self.two_model and symmetry != "avg8"
        self.stream = stream               # a torch stream of this engine's own (DualEngine); None = the caller's current stream
        self._graphs, self._graph_pool = {}, None
        self._pol_static = self._val_static = self._kdev = None
        self._in_flight = False
        self.n_graph_replays = 0
        self._done_ev = None
        self.round_timeout_s = float(conf.get('ROUND_TIMEOUT_S', 300.0))

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.sgo_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ game slots
    def start_games(self, slots, noises=None, uniforms=None, resign=None, ids=None):
        """(Re)start game slots.  Draws default to numpy's own generators (np.random.dirichlet /
        random_sample -- the same distributions the reference draws from, play.py:401, and
        nomodel_self_play.py:135 through np.random.choice); tests inject recorded draws instead."""
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        n = len(slots)
        if n == 0:
            return
        if noises is None:
            noises = self.rng.dirichlet([self.alpha] * self.A, size=n)
        noises = np.ascontiguousarray(noises, dtype=np.float64).reshape(n, self.A)
        if uniforms is None:
            uniforms = self.rng.random_sample((n, max(1, self.max_moves)))
        uniforms = np.ascontiguousarray(uniforms, dtype=np.float64).reshape(n, -1)
        res = None
        if resign is not None:
            # `if resign and value <= resign` (nomodel_self_play.py:171): None AND 0.0 mean "never resign"
            res = np.array([np.nan if not r else r for r in resign], dtype=np.float32)
        with self._on_stream():
            _lib.check(self.lib.sgo_start_games(self.ctx, C.c_int(n), _lib.ptr(slots), _lib.ptr(noises), _lib.ptr(uniforms),
                                                C.c_int(uniforms.shape[1]), _lib.ptr(res), _lib.stream_ptr()), "sgo_start_games")
        for i, s in enumerate(slots):
            self.records[int(s)] = []
            self.game_ids[int(s)] = None if ids is None else ids[i]

    def start_eval_games(self, slots, first_model=None, uniforms=None, resign_model1=None, resign_model2=None, ids=None):
        """(Re)start two-model game slots.  first_model[i] = 0: model1 moves first (plays black).  Default: the reference's
        coin, play.choose_first_player (one draw of Python's `random` per game, model1 first below .5)."""
        from . import play
        slots = np.ascontiguousarray(slots, dtype=np.int32)
        n = len(slots)
        if n == 0:
            return
        if first_model is None:
            first_model = [0 if play.choose_first_player(0, 1)[0] == 0 else 1 for _ in range(n)]
        first_model = np.ascontiguousarray(first_model, dtype=np.int32)
        if uniforms is None:
            uniforms = self.rng.random_sample((n, max(1, self.max_moves)))
        uniforms = np.ascontiguousarray(uniforms, dtype=np.float64).reshape(n, -1)

        def thr(r):
            if r is None:
                return None
            return np.array([np.nan if not v else v for v in (r if np.ndim(r) else [r] * n)], dtype=np.float32)

        r1, r2 = thr(resign_model1), thr(resign_model2)
        with self._on_stream():
            _lib.check(self.lib.sgo_start_games2(self.ctx, C.c_int(n), _lib.ptr(slots), _lib.ptr(uniforms), C.c_int(uniforms.shape[1]),
                                                 _lib.ptr(r1), _lib.ptr(r2), _lib.ptr(first_model), _lib.stream_ptr()), "sgo_start_games2")
        for i, s in enumerate(slots):
            self.records[int(s)] = []
            self.game_ids[int(s)] = None if ids is None else ids[i]

    # ------------------------------------------------------------------ one engine step
    def _lut(self, k):
        torch = self.torch
        if self._luts is None:
            from .symmetry import sym_lut
            self._luts = [torch.from_numpy(sym_lut(self.S, kk).astype(np.int64)).to(self.device) for kk in range(8)]
        return self._luts[k]

    def _draw_k(self):
        if self.symmetry == "identity":
            return 0
        if self.symmetry == "random1":
            return self.pyrng.randrange(7)  # choice(SYMMETRIES), symmetry.py:128
        return int(self.symmetry)

    def _index_view(self):
        """The device-side evaluation list (record indices, int32 [G * E]) as a torch tensor over the engine's own memory."""
        if getattr(self, "_idx_t", None) is None:
            class _Mem(object):
                pass
            m = _Mem()
            m.__cuda_array_interface__ = {"shape": (self.G * self.E,), "typestr": "<i4", "data": (self._idx_ptr, False), "version": 2}
            self._idx_t = self.torch.as_tensor(m, device=self.device)
        return self._idx_t

    def _forward(self, n, k):
        torch = self.torch
        self.n_net_calls += 1
        self.n_net_positions += n
        if self.packed:
            if not self.two_model:
                p, v = self.net.predict_packed(self._rec_ptr, self._idx_ptr, n, k)
                return p.to(torch.float32), v.to(torch.float32).reshape(n)
            x = None
        else:
            _lib.check(self.lib.sgo_collect(self.ctx, C.c_int(k), C.c_int(self.layout), C.c_int(self.dtype),
                                            _lib.ptr(self.nn_in), _lib.stream_ptr()), "sgo_collect")
            x = self.nn_in[:n]
            if self.layout == 1:
                x = x.permute(0, 2, 3, 1)  # present the reference's NHWC view
            if not self.two_model:
                p, v = self.net.predict_on_batch(x)
                return p.to(torch.float32), v.to(torch.float32).reshape(n)
        # two-model games: every row of the list belongs to the model that is to move in its game
        ids = np.zeros(n, dtype=np.int32)
        _lib.check(self.lib.sgo_eval_models(self.ctx, C.c_int(n), _lib.ptr(ids)), "sgo_eval_models")
        pol = torch.empty((n, self.A), dtype=torch.float32, device=self.device)
        val = torch.empty((n,), dtype=torch.float32, device=self.device)
        for m, net in ((0, self.net), (1, self.net2)):
            idx = np.flatnonzero(ids == m)
            if len(idx) == 0:
                continue
            self.n_model_positions[m] += len(idx)
            if len(idx) == n:
                p, v = net.predict_packed(self._rec_ptr, self._idx_ptr, n, k) if self.packed else net.predict_on_batch(x)
                pol.copy_(p)
                val.copy_(v.reshape(n))
                continue
            it = torch.from_numpy(idx).to(self.device)
            if self.packed:
                sub = self._index_view()[:n].index_select(0, it).contiguous()       # this model's rows of the list
                p, v = net.predict_packed(self._rec_ptr, sub.data_ptr(), len(idx), k)
            else:
                p, v = net.predict_on_batch(x.index_select(0, it).contiguous())
            pol.index_copy_(0, it, p.to(torch.float32))
            val.index_copy_(0, it, v.to(torch.float32).reshape(-1))
        return pol, val

    def _on_stream(self):
        import contextlib
        return self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _bucket(self, n):
        """Batch sizes the captured rounds exist for: n rounded up to 1/32 of the largest list (rows beyond n are evaluated on
        stale list entries and ignored by k_search: at most 3 % wasted network work, and lockstep games hit G and G * E exactly)."""
        full = self.G * self.E
        q = max(64, full // 32)
        return min(full, -(-n // q) * q)

    def _capture(self, nb):
        torch = self.torch
        if self._kdev is None:
            self._kdev = torch.zeros(1, dtype=torch.int32, device=self.device)
            self._pol_static = torch.zeros((self.G * self.E, self.A), dtype=torch.float32, device=self.device)
            self._val_static = torch.zeros((self.G * self.E,), dtype=torch.float32, device=self.device)
            self._graph_pool = torch.cuda.graph_pool_handle()
            # capture on the engine's own stream where it has one: every extra HIP stream is another candidate for sharing a
            # hardware queue with the stepping streams (see sejonggo_amd/__init__.py on GPU_MAX_HW_QUEUES)
            self._cap_stream = self.stream if self.stream is not None else torch.cuda.Stream(device=self.device)
        kd = self._kdev.data_ptr()
        # library handles / workspaces of this batch size come into being outside the capture
        self.net.predict_packed(self._rec_ptr, self._idx_ptr, nb, 0, kd)
        torch.cuda.current_stream().synchronize()
        g = torch.cuda.CUDAGraph()
        # thread_local: other threads of the process (RCCL's watchdog polling its events, writer threads) may keep calling
        # the runtime while this thread captures
        with torch.cuda.graph(g, pool=self._graph_pool, stream=self._cap_stream, capture_error_mode="thread_local"):
            p, v = self.net.predict_packed(self._rec_ptr, self._idx_ptr, nb, 0, kd)
            self._pol_static[:nb].copy_(p)
            self._val_static[:nb].copy_(v.reshape(nb))
            _lib.check(self.lib.sgo_step_enqueue(self.ctx, self._pol_static.data_ptr(), self._val_static.data_ptr(), kd,
                                                 _lib.stream_ptr()), "sgo_step_enqueue")
        self._graphs[nb] = g
        return g

    def launch(self):
        """First half of a step: queue the evaluation of the listed positions and the engine step that consumes it, without
        waiting (captured-round mode); in the other modes the whole step runs here.  finish() completes it."""
        n = self.status.n_eval if self._primed else 0
        if not (self.graph and n > 0):
            self._step_eager()
            self._in_flight = False
            return
        with self._on_stream():
            nb = self._bucket(n)
            g = self._graphs.get(nb) or self._capture(nb)
            self._kdev.fill_(self._draw_k())       # one symmetry per list (symmetry.py:127-132), read by the stem and by k_search
            self.n_net_calls += 1
            self.n_net_positions += n
            g.replay()
            self.n_graph_replays += 1
            if self._done_ev is None:
                self._done_ev = self.torch.cuda.Event()
            self._done_ev.record()
        self._in_flight = True

    def finish(self):
        """Second half: wait for the queued round, read what it reported."""
        if self._in_flight:
            # wait by polling the round's event, with a deadline: a round that never completes becomes an error the caller
            # sees, not a process that sits in hipStreamSynchronize for ever (see DualEngine.MAX_ROUND_PIXELS)
            import time
            ev, t0 = self._done_ev, time.perf_counter()
            while not ev.query():
                dt = time.perf_counter() - t0
                if dt > 2e-3:
                    time.sleep(5e-5)
                if dt > self.round_timeout_s:
                    raise _lib.SgoError("a captured round did not complete within %.0f s (stream stalled)" % self.round_timeout_s)
            _lib.check(self.lib.sgo_step_status(self.ctx, C.byref(self.status)), "sgo_step_status")
            self._in_flight = False
            self.n_steps += 1
        if self.status.error and self.raise_on_error:
            raise _lib.SgoError("game slot %d failed with error %d (%s)" % (
                self.status.error_game, self.status.error,
                {-201: "tree-block pool exhausted: raise blocks_per_game", -202: "ran out of injected random draws",
                 -203: "engine state error"}.get(self.status.error, "?")))
        return self.status

    def step(self):
        """One engine step; returns the status struct."""
        self.launch()
        return self.finish()

    def _step_eager(self):
        with self._on_stream():
            self._step_eager_body()

    def _step_eager_body(self):
        torch = self.torch
        n = self.status.n_eval if self._primed else 0
        if n > 0:
            if self.symmetry == "avg8":
                pol = None
                val = None
                for k in range(8):
                    p, v = self._forward(n, k)
                    p = p.index_select(1, self._lut(k))   # reverse_*: policy[:, SWAP]
                    pol = p if pol is None else pol + p
                    val = v if val is None else val + v
                pol = (pol / 8.0).contiguous()
                val = (val / 8.0).contiguous()
                k_used = 0
            else:
                k_used = self._draw_k()
                pol, val = self._forward(n, k_used)
                pol = pol.contiguous()
                val = val.contiguous()
            self._policy, self._value = pol, val  # keep alive until the step has consumed them
            pp, vp = _lib.ptr(pol), _lib.ptr(val)
        else:
            pp, vp, k_used = None, None, 0
        _lib.check(self.lib.sgo_step(self.ctx, pp, vp, C.c_int(k_used), _lib.stream_ptr(), C.byref(self.status)), "sgo_step")
        self._primed = True
        self.n_steps += 1

    # ------------------------------------------------------------------ records / results
    def drain(self):
        """Moves recorded since the last drain, appended to the per-slot game records."""
        n = self.status.n_records
        if n <= 0:
            return 0
        cap = 2 * self.G + 16
        recs = np.zeros(cap, dtype=_lib.MOVE_RECORD_DTYPE)
        packed = np.zeros((cap, self.RW), dtype=np.uint32)
        policy = np.zeros((cap, self.A), dtype=np.float64)
        n = _lib.check(self.lib.sgo_drain_records(self.ctx, C.c_int(cap), _lib.ptr(recs), _lib.ptr(packed), _lib.ptr(policy)),
                       "sgo_drain_records")
        self.status.n_records = 0
        if n == 0:
            return 0
        order = np.lexsort((recs["move_n"][:n], recs["game"][:n])).tolist()
        # columns as Python lists once (per-element numpy scalar access is the slow part of this loop); the rows of a batch
        # stay views of ONE copy of its policy / record arrays
        games, mvn, acts = recs["game"][:n].tolist(), recs["move_n"][:n].tolist(), recs["action"][:n].tolist()
        players, seqs, vals = recs["player"][:n].tolist(), recs["game_seq"][:n].tolist(), recs["value"][:n].copy()
        pol, pk = policy[:n].copy(), packed[:n].copy()
        S = self.S
        for i in order:
            a = acts[i]
            rec = MoveRecord(policy=pol[i], value=vals[i], move=(a % S, a // S), move_n=mvn[i], player=players[i], packed=pk[i],
                             action=a, game_seq=seqs[i])
            rec.size = S
            self.records.setdefault(games[i], []).append(rec)
        return n

    def results(self, slots=None):
        slots = np.arange(self.G, dtype=np.int32) if slots is None else np.ascontiguousarray(slots, dtype=np.int32)
        out = np.zeros(len(slots), dtype=_lib.GAME_RESULT_DTYPE)
        _lib.check(self.lib.sgo_game_results(self.ctx, C.c_int(len(slots)), _lib.ptr(slots), _lib.ptr(out)), "sgo_game_results")
        return out

    def game_data(self, slot, result, model_name=None):
        """The reference's game_data dict (nomodel_self_play.py:261-270) for a finished slot."""
        if self.two_model:
            return self._eval_game_data(slot, result)
        name = model_name or getattr(self.net, "name", "model")
        winner = int(result["winner"])
        player_string = {1: "B", 0: "D", -1: "W"}
        if int(result["end_reason"]) == 1:
            winner_string = "%s+R" % player_string[int(result["last_player"])]
        else:
            winner_string = "%s+%s" % (player_string[winner], abs(int(result["black"]) - float(result["white"])))
        return {
            'moves': self.records.get(int(slot), []),
            'modelB_name': name, 'modelW_name': name,
            'winner': {1: 1, -1: 0, 0: None}[winner],
            'winner_model': None if winner == 0 else name,
            'result': winner_string,
            'resign_model1': None, 'resign_model2': None,
            'end_reason': END_REASONS[int(result["end_reason"])],
            'black_points': int(result["black"]), 'white_points': float(result["white"]),
            'slot': int(slot), 'id': self.game_ids.get(int(slot)),
            'blocks_high_water': int(result["blocks_high_water"]),
        }

    def _eval_game_data(self, slot, result):
        """game_data of a two-model game (nomodel_self_play.py:227-270): model names by colour, winner_model with the
        reference's rule -- right while model1 plays black, the loser's name otherwise (:247) -- behind COMPAT_WINNER_MODEL."""
        names = (getattr(self.net, "name", "model1"), getattr(self.net2, "name", "model2"))
        model1_black = int(result["first_model"]) == 0
        nameB, nameW = (names[0], names[1]) if model1_black else (names[1], names[0])
        winner = int(result["winner"])
        tag = {1: "B", 0: "D", -1: "W"}
        if int(result["end_reason"]) == 1:
            winner_string = "%s+R" % tag[int(result["last_player"])]
        else:
            winner_string = "%s+%s" % (tag[winner], abs(int(result["black"]) - float(result["white"])))
        if winner == 0:
            winner_model = None
        elif conf.get('COMPAT_WINNER_MODEL', True):
            winner_model = nameB if (winner == 1) == model1_black else nameW
        else:
            winner_model = nameB if winner == 1 else nameW
        return {
            'moves': self.records.get(int(slot), []),
            'modelB_name': nameB, 'modelW_name': nameW,
            'winner': {1: 1, -1: 0, 0: None}[winner], 'winner_model': winner_model, 'result': winner_string,
            'resign_model1': None, 'resign_model2': None,
            'end_reason': END_REASONS[int(result["end_reason"])],
            'black_points': int(result["black"]), 'white_points': float(result["white"]),
            'slot': int(slot), 'id': self.game_ids.get(int(slot)), 'first_model': int(result["first_model"]),
        }

    def run(self, max_steps=None):
        """Steps until no game is active; returns finished game_data dicts (slots are not restarted)."""
        steps = 0
        while True:
            st = self.step()
            if st.n_records >= self.G:
                self.drain()
            steps += 1
            if st.n_active == 0 or (max_steps is not None and steps >= max_steps):
                break
        self.drain()
        res = self.results()
        out = []
        for s in range(self.G):
            if res[s]["done"] == 1:
                out.append(self.game_data(s, res[s]))
        return out

    # ------------------------------------------------------------------ introspection (parity tests)
    def root_table(self, slot):
        A = self.A
        N = np.zeros(A, np.int32); W = np.zeros(A, np.float32); Q = np.zeros(A, np.float32)
        P = np.zeros(A, np.float64); EX = np.zeros(A, np.int8)
        rc, rv = C.c_int32(0), C.c_float(0)
        _lib.check(self.lib.sgo_root_table(self.ctx, C.c_int(slot), _lib.ptr(N), _lib.ptr(W), _lib.ptr(Q), _lib.ptr(P),
                                           _lib.ptr(EX), C.byref(rc), C.byref(rv)), "sgo_root_table")
        return {"N": N, "W": W, "Q": Q, "P": P, "EX": EX, "root_count": rc.value, "root_value": np.float32(rv.value)}

    def tree_serialize(self, slot):
        nn, ne = C.c_int64(0), C.c_int64(0)
        sz = _lib.check(self.lib.sgo_tree_serialize(self.ctx, C.c_int(slot), None, C.c_int64(0), C.byref(nn), C.byref(ne)),
                        "sgo_tree_serialize")
        buf = np.zeros(max(1, sz), dtype=np.uint8)
        _lib.check(self.lib.sgo_tree_serialize(self.ctx, C.c_int(slot), _lib.ptr(buf), C.c_int64(sz), C.byref(nn), C.byref(ne)),
                   "sgo_tree_serialize")
        return buf[:sz], nn.value, ne.value

    def tree_dict(self, slot):
        """The slot's device tree as the reference's nested dict nodes (play.py:376-421: index, count, value,
        mean_value, p, subtree, parent, virtual_loss) -- a read-only snapshot for inspection, GTP-style front-ends
        and tests; the search itself never leaves the GPU."""
        nn = C.c_int64(0)
        sz = _lib.check(self.lib.sgo_tree_dump(self.ctx, C.c_int(slot), None, C.c_int64(0), C.byref(nn)), "sgo_tree_dump")
        buf = np.zeros(max(1, sz), dtype=np.uint8)
        _lib.check(self.lib.sgo_tree_dump(self.ctx, C.c_int(slot), _lib.ptr(buf), C.c_int64(sz), C.byref(nn)), "sgo_tree_dump")
        recs = np.frombuffer(buf[:sz].tobytes(), dtype=np.dtype([("a", "<i4"), ("n", "<i4"), ("w", "<f4"), ("q", "<f4"),
                                                                 ("p", "<f8"), ("vl", "<i4"), ("ex", "<i4"), ("d", "<i4"),
                                                                 ("pad", "<i4")]))
        t = self.root_table(slot)
        root = {'index': -1, 'count': int(t["root_count"]), 'value': t["root_value"],
                'mean_value': (t["root_value"] / np.float32(t["root_count"])) if t["root_count"] else 0, 'p': 1,
                'subtree': {}, 'parent': None, 'virtual_loss': 0}
        stack = [root]
        for r in recs:
            del stack[int(r["d"]) + 1:]
            parent = stack[-1]
            node = {'index': int(r["a"]), 'count': int(r["n"]), 'value': np.float32(r["w"]), 'mean_value': np.float32(r["q"]),
                    'p': np.float64(r["p"]), 'subtree': {}, 'parent': parent, 'virtual_loss': int(r["vl"])}
            parent['subtree'][int(r["a"])] = node
            stack.append(node)
        return root

    def board(self, slot):
        b = np.zeros((1, self.S, self.S, 17), dtype=np.int32)
        _lib.check(self.lib.sgo_game_board(self.ctx, C.c_int(slot), _lib.ptr(b)), "sgo_game_board")
        return b

    def pool_info(self):
        """Tree-block accounting of the context: private blocks per game, local ids per game, shared pool size, shared blocks
        free now, fewest ever free."""
        out = (C.c_int64 * 6)()
        _lib.check(self.lib.sgo_pool_info(self.ctx, out, 6), "sgo_pool_info")
        return {"private_per_game": int(out[0]), "ids_per_game": int(out[1]), "shared_blocks": int(out[2]),
                "shared_free": int(out[3]), "shared_free_low_water": int(out[4]), "games": int(out[5])}

    def set_halt(self, slot, move_n):
        _lib.check(self.lib.sgo_set_halt(self.ctx, C.c_int(slot), C.c_int(move_n)), "sgo_set_halt")

    def advance_timing(self):
        ms, n, p = C.c_double(0), C.c_int64(0), C.c_int64(0)
        _lib.check(self.lib.sgo_advance_timing(self.ctx, C.byref(ms), C.byref(n), C.byref(p)), "sgo_advance_timing")
        return ms.value, n.value, p.value


class _SumStatus(object):
    """The status of a DualEngine step: counts summed over the halves, the first error with its slot in the whole population."""
    __slots__ = ("n_eval", "n_records", "n_active", "n_done", "error", "error_game", "total_moves", "total_evals", "none_events")

    def __init__(self):
        for k in self.__slots__:
            setattr(self, k, 0)


class DualEngine(object):
    """The resident games as TWO half-populations on two HIP streams, alternating (ping-pong): while the MFMA-bound tower of
    one half's evaluation list runs, the other half's k_search / k_compact / board_advance (one wavefront per game, latency- and
    HBM-bound) run beside it on the idle VALU side, and the host prepares one half while the GPU works on the other.  Each half
    is a SelfPlayEngine with captured rounds, so the host issues one graph launch per half and round.  Games are independent
    state machines: a game's moves, trees and records are what the single-context engine produces for the same draws.
    Same surface as SelfPlayEngine for the drivers (start_games / step / drain / results / game_data / records / status)."""

    # Largest round (leaf pixels = games x energy x tower points) the two-stream form is allowed at.  With LONG kernels on both
    # streams (tower launches of >= ~0.5 ms: 19x19 with >= 512 games) the run stalls intermittently after a few captured
    # rounds -- the GPU stays responsive, no kernel of ours waits on another, 4 or 8 hardware queues alike
    # (tools/debug_dual.py dual 19 512 20 400 60; gpurun_out/r03ae..ah); with the short kernels of small boards it has run tens
    # of thousands of rounds.  Until that is understood the form is refused where it has nothing to gain anyway (a 19x19 round
    # is one 40-85 ms tower pass: 21.9 ms per round with either engine at 256 games).
    MAX_ROUND_PIXELS = 400000

    def __init__(self, net, n_games=None, seed=0, device=0, allow_large=False, **kw):
        import torch
        G = n_games or conf['GAMES_PER_GPU']
        if G < 2:
            raise ValueError("DualEngine needs at least two games")
        S_, E_ = kw.get('size') or conf['SIZE'], kw.get('energy') or conf['ENERGY']
        if not allow_large and G * E_ * (S_ - 2) * (S_ - 2) >= self.MAX_ROUND_PIXELS:
            raise ValueError("DualEngine: a round of %d x %d leaves on a %dx%d board is too large for the two-stream form (see "
                             "DualEngine.MAX_ROUND_PIXELS); use SelfPlayEngine" % (G, E_, S_, S_))
        kw.pop("stream", None)
        kw.setdefault("graph", True)
        sizes = [G - G // 2, G // 2]
        torch.cuda.set_device(torch.device("cuda", device))
        self.halves = [SelfPlayEngine(net, n_games=sizes[i], seed=seed + 7919 * i, device=device,
                                      stream=torch.cuda.Stream(device=torch.device("cuda", device)), **kw) for i in range(2)]
        self.offsets = [0, sizes[0]]
        self.G, self.S, self.A, self.E = G, self.halves[0].S, self.halves[0].A, self.halves[0].E
        self.sims, self.max_moves, self.net = self.halves[0].sims, self.halves[0].max_moves, net
        self.packed, self.graph, self.two_model = self.halves[0].packed, self.halves[0].graph, False
        self.status = _SumStatus()
        self.records = _RecordView(self)
        self._started = False

    # -- slots -----------------------------------------------------------------------------------------------------
    def _split(self, slots):
        slots = np.ascontiguousarray(slots, dtype=np.int64)
        which = (slots >= self.offsets[1]).astype(np.int64)
        return [(np.flatnonzero(which == h), slots[which == h] - self.offsets[h]) for h in range(2)]

    def start_games(self, slots, noises=None, uniforms=None, resign=None, ids=None):
        for h, (pos, local) in enumerate(self._split(slots)):
            if len(local) == 0:
                continue
            pick = lambda a: None if a is None else [a[int(i)] for i in pos]
            self.halves[h].start_games(local, noises=None if noises is None else np.asarray(noises)[pos],
                                       uniforms=None if uniforms is None else np.asarray(uniforms)[pos],
                                       resign=pick(resign), ids=pick(ids))

    def _half_of(self, slot):
        h = 1 if slot >= self.offsets[1] else 0
        return self.halves[h], int(slot) - self.offsets[h]

    # -- stepping --------------------------------------------------------------------------------------------------
    def step(self):
        """Every half advances one round.  Order per half: finish the round queued by the previous call, queue the next one --
        so while the host handles one half, the other half's round is running, and both rounds overlap on the GPU."""
        if not self._started:
            for e in self.halves:
                e.launch()
            self._started = True
        import os
        if os.environ.get("SGO_DUAL_SERIAL") == "1":      # diagnostic: the halves' rounds strictly one after the other
            for e in self.halves:
                e.finish()
            for e in self.halves:
                e.launch()
                e.finish()
                e._in_flight = False
            return self._sum_status()
        for e in self.halves:
            e.finish()
            e.launch()
        return self._sum_status()

    def sync(self):
        """Complete the rounds in flight (before reading records / results of a population that is about to be inspected)."""
        for e in self.halves:
            e.finish()
        self._started = False
        return self._sum_status()

    def _sum_status(self):
        st = self.status
        for k in ("n_eval", "n_records", "n_active", "n_done", "total_moves", "total_evals", "none_events"):
            setattr(st, k, sum(int(getattr(e.status, k)) for e in self.halves))
        st.error, st.error_game = 0, 0
        for h, e in enumerate(self.halves):
            if e.status.error and not st.error:
                st.error, st.error_game = int(e.status.error), int(e.status.error_game) + self.offsets[h]
        return st

    # -- records / results -----------------------------------------------------------------------------------------
    def drain(self):
        self.sync()
        return sum(e.drain() for e in self.halves)

    def results(self, slots=None):
        self.sync()
        if slots is None:
            return np.concatenate([e.results() for e in self.halves])
        slots = np.ascontiguousarray(slots, dtype=np.int64)
        out = np.zeros(len(slots), dtype=_lib.GAME_RESULT_DTYPE)
        for h, (pos, local) in enumerate(self._split(slots)):
            if len(local):
                out[pos] = self.halves[h].results(local)
        return out

    def game_data(self, slot, result, model_name=None):
        e, local = self._half_of(slot)
        gd = e.game_data(local, result, model_name)
        gd['slot'] = int(slot)
        return gd

    def run(self, max_steps=None):
        steps = 0
        while True:
            st = self.step()
            steps += 1
            if st.n_records >= self.G:
                self.drain()
            if st.n_active == 0 or (max_steps is not None and steps >= max_steps):
                break
        self.drain()
        res = self.results()
        return [self.game_data(s, res[s]) for s in range(self.G) if res[s]["done"] == 1]

    def root_table(self, slot):
        e, local = self._half_of(slot)
        return e.root_table(local)

    def tree_serialize(self, slot):
        e, local = self._half_of(slot)
        return e.tree_serialize(local)

    def board(self, slot):
        e, local = self._half_of(slot)
        return e.board(local)

    def pool_info(self):
        return [e.pool_info() for e in self.halves]

    def advance_timing(self):
        t = [e.advance_timing() for e in self.halves]
        return tuple(sum(x[i] for x in t) for i in range(3))

    @property
    def n_net_calls(self):
        return sum(e.n_net_calls for e in self.halves)

    @property
    def n_net_positions(self):
        return sum(e.n_net_positions for e in self.halves)

    @property
    def n_steps(self):
        return sum(e.n_steps for e in self.halves)

    def close(self):
        for e in self.halves:
            e.close()


class _RecordView(object):
    """engine.records of a DualEngine: slot -> move list, over the halves' own dicts."""

    def __init__(self, dual):
        self.dual = dual

    def _loc(self, slot):
        e, local = self.dual._half_of(slot)
        return e.records, local

    def __getitem__(self, slot):
        d, k = self._loc(slot)
        return d[k]

    def __setitem__(self, slot, value):
        d, k = self._loc(slot)
        d[k] = value

    def get(self, slot, default=None):
        d, k = self._loc(slot)
        return d.get(k, default)

    def setdefault(self, slot, default):
        d, k = self._loc(slot)
        return d.setdefault(k, default)
