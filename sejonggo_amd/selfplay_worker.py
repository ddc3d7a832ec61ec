"""Drop-in for the reference's selfplay_worker.py: NoModelSelfPlayWorker(process_id) and
SelfPlayWorker(gpuid, forever=False, one_game_only=-1) are multiprocessing.Process subclasses with the
reference's start()/join() life cycle, game-directory reservation (selfplay_worker.py:83-90), resign-
threshold calibration (:92-112) and zero-move clean-up (:115-118).

NoModelSelfPlayWorker -- difference in kind, not in contract: the reference runs ONE game per process (and 8 pool
workers under it); here one process drives ONE MI355X that keeps conf['GAMES_PER_GPU'] games resident, reserving the
next free game directory whenever a slot finishes.  Run one worker per GPU (conf['N_GAME_PROCESS'] = number of GPUs).

SelfPlayWorker follows selfplay_worker.py:29-58 literally: load the best model, play self_play.model_self_play (the
sync path, one game at a time, game number `one_game_only` when >= 0), and with `forever` keep reloading the best model,
sleeping conf['SLEEP_SECONDS'] while it has not changed.  conf['SELFPLAY_WORKER_ENGINE'] = 'device' swaps the sync
game loop for the many-games device engine (same files on disk, far higher throughput; not the reference's search).

Processes and the GPU.  A child forked from a process that has initialised a HIP runtime cannot use the GPU.  The
workers therefore start by fork (the reference's behaviour: conf and module state are inherited) only while the
parent is GPU-free -- which main_selfplay.main() guarantees -- and by 'spawn' otherwise, carrying a snapshot of
conf (and of a picklable model factory) into the fresh interpreter."""
import os
import sys
import time
import traceback
import multiprocessing
from multiprocessing import Process
from random import random

import numpy as np

from .conf import conf


class GameScheduler(object):
    """Host logic shared by both workers: which game number a free slot plays next, and the resign
    threshold (selfplay_worker.py:82-112).  Pure bookkeeping; no compute."""

    def __init__(self, self_play_dir, model_name, n_games, resignation_percent, allowed_error, rand=random, only_game=None):
        self.dir, self.model_name, self.n_games = self_play_dir, model_name, n_games
        self.only_game = only_game     # self_play.py:303-304: `one_game_only` plays exactly that game number
        self.resignation_percent, self.allowed_error = resignation_percent, allowed_error
        self.rand = rand
        self.next_game = 0
        self.current_resign = None
        self.min_values = []

    def reserve(self):
        """Next game number whose directory could be created, or None when range(n_games) is exhausted."""
        while self.next_game < self.n_games:
            g = self.next_game
            self.next_game += 1
            if self.only_game is not None and g != self.only_game:
                continue
            directory = os.path.join(self.dir, self.model_name, "game_%05d" % g)
            if os.path.isdir(directory):
                continue
            try:
                os.makedirs(directory)
            except Exception:
                continue
            return g
        return None

    def pick_resign(self):
        return self.current_resign if self.rand() > self.resignation_percent else None

    def finished(self, game_data, resign):
        """selfplay_worker.py:100-112: only no-resign games calibrate the threshold; the list is kept in
        arrival order (not sorted), as the reference does."""
        if resign is None and game_data['moves']:
            winner = game_data['winner']
            vals = [m['value'] for m in game_data['moves'][::2]] if winner == 1 else [m['value'] for m in game_data['moves'][1::2]]
            if vals:
                self.min_values.append(min(vals))
                idx = int(self.allowed_error * len(self.min_values))
                if idx > 0:
                    self.current_resign = self.min_values[idx]

    def discard(self, game_no):
        try:
            os.rmdir(os.path.join(self.dir, self.model_name, "game_%05d" % game_no))
        except OSError:
            pass


def run_selfplay(gpu_id, model_indicator="BEST_SYM", n_games=None, games_per_gpu=None, on_game=None, max_steps=None,
                 only_game=None, engine_kwargs=None, stats=None):
    """The worker body, callable in-process (tests, bench) as well as from the Process subclasses.

    The stepping thread only steps the engine, restarts finished slots (one batched sgo_start_games per step) and
    hands finished games to conf['WRITER_THREADS'] writer threads, which expand the packed positions and write the
    sample files; `stats` (a dict) receives wall-clock totals per activity."""
    import time
    from concurrent.futures import ThreadPoolExecutor
    from .engine import SelfPlayEngine
    from .predicting_queue_worker import init_predicting_workers, get_model, put_name_request
    from .sgfsave import save_self_play_data
    init_predicting_workers([gpu_id])
    net = get_model(model_indicator, gpu_id)
    model_name = put_name_request(model_indicator)
    n_games = conf['N_GAMES'] if n_games is None else n_games
    G = min(games_per_gpu or conf['GAMES_PER_GPU'], max(1, n_games if only_game is None else 1))
    sched = GameScheduler(conf['SELF_PLAY_DIR'], model_name, n_games, conf['RESIGNATION_PERCENT'],
                          conf['RESIGNATION_ALLOWED_ERROR'], only_game=only_game)
    sym = conf.get('SYMMETRY_MODE', 'random1') if model_indicator.endswith("_SYM") else "identity"
    kw = dict(size=conf['SIZE'], n_games=G, sims=conf['MCTS_SIMULATIONS'], energy=conf['ENERGY'],
              stop_exploration=conf['STOP_EXPLORATION'], komi=conf['KOMI'], self_play=True, symmetry=sym,
              device=gpu_id, seed=gpu_id, raise_on_error=False, blocks_per_game=int(conf.get('BLOCKS_PER_GAME', 0) or 0),
              shared_blocks=int(conf.get('SHARED_BLOCKS', 0) or 0))
    kw.update(engine_kwargs or {})
    # conf['ENGINE_HALVES'] = 2: two half-populations alternating on two streams, every round a captured launch chain
    # (engine.DualEngine); conf['ENGINE_GRAPH']: captured rounds on one population.  Both pay on small boards / shallow nets,
    # where a round is launch-bound; at 19x19 with the 20-block net a round is one 85-ms tower and neither matters.
    halves = int(conf.get('ENGINE_HALVES', 0) or 0)
    if halves == 0:
        # auto: the two-stream form pays where a round is short (the tower's launch does not fill the chip for long): below
        # 200 000 leaf pixels per round (9x9 / 256 games: 100 000; 19x19 / 1 024 games: 2.4 million), and is refused above
        # DualEngine.MAX_ROUND_PIXELS, where long kernels on two streams have stalled the run
        t = max(1, kw['size'] - 2)
        halves = 2 if (getattr(net, "packed_ok", False) and kw['n_games'] >= 2 and kw['n_games'] * kw['energy'] * t * t < 200000) else 1
    if halves == 2 and kw['n_games'] >= 2:
        from .engine import DualEngine
        eng = DualEngine(net, **kw)            # refuses large rounds (DualEngine.MAX_ROUND_PIXELS): an explicit setting fails loudly
    else:
        eng = SelfPlayEngine(net, graph=bool(conf.get('ENGINE_GRAPH', False)), **kw)
    slot_game, slot_resign = {}, {}
    t = {"step": 0.0, "turnover": 0.0, "writer_wait": 0.0, "steps": 0, "moves": 0, "games": 0, "files": 0}
    # Finished games leave the stepping thread at once.  Default: conf['WRITER_THREADS'] threads of this process (libhdf5
    # is not thread-safe, so the file writes themselves are serialised on its lock: ~4-8 k files/s on the GPU box, plenty at
    # 19x19 / 400 sims).  conf['WRITER_PROCESSES'] > 0 hands the games to that many 'spawn'ed torch-free writer processes
    # instead, each with its own libhdf5 -- for small boards, where one GPU produces more positions than one lock writes.
    n_proc = int(conf.get('WRITER_PROCESSES', 0) or 0)
    if n_proc > 0 and on_game is None and not conf.get('SGF_ENABLED'):
        from concurrent.futures import ProcessPoolExecutor
        from .sgfsave import write_packed_game
        writers = ProcessPoolExecutor(max_workers=n_proc, mp_context=multiprocessing.get_context('spawn'))
        conf_items = {k: conf[k] for k in ('SELF_PLAY_DIR', 'COMPAT_Z', 'WRITE_NPZ_TWIN') if k in conf}
    else:
        n_proc = 0
        writers = ThreadPoolExecutor(max_workers=max(1, int(conf.get('WRITER_THREADS', 2))))
    pending = []

    def write(g, gd):
        save_self_play_data(model_name, g, gd)
        if on_game is not None:
            on_game(g, gd)
        return len(gd['moves'])

    def submit(g, gd):
        if n_proc:
            mv = gd['moves']
            return writers.submit(write_packed_game, conf_items, model_name, g, conf['SIZE'],
                                  np.stack([m['packed'] for m in mv]), np.stack([np.asarray(m['policy'], dtype=np.float32) for m in mv]),
                                  np.array([m['player'] for m in mv]), np.array([m['move_n'] for m in mv]), gd['winner'])
        return writers.submit(write, g, gd)

    def reap(block=False):
        keep = []
        for f in pending:
            if block or f.done():
                t["files"] += f.result()          # re-raises a writer's exception in the stepping thread
            else:
                keep.append(f)
        pending[:] = keep

    def fill(slots):
        start, res, ids = [], [], []
        for s in slots:
            g = sched.reserve()
            if g is None:
                continue
            r = sched.pick_resign()
            slot_game[s], slot_resign[s] = g, r
            start.append(s); res.append(r); ids.append(g)
        if start:
            eng.start_games(start, resign=res, ids=ids)
        return len(start)

    played = 0
    idle = 0          # finished slots that could not be refilled (no game numbers left)
    t_loop0 = time.perf_counter()
    try:
        active = fill(range(G))
        idle = G - active
        steps = 0
        while active > 0:
            t0 = time.perf_counter()
            st = eng.step()
            steps += 1
            if st.n_records >= G:
                eng.drain()
            t1 = time.perf_counter()
            t["step"] += t1 - t0
            if st.n_done > idle or (st.error and st.error_game in slot_game):
                eng.drain()
                res = eng.results()
                free = []
                for s in list(slot_game):
                    if res[s]["done"] < 0:
                        # the slot failed (typically SGO_ERR_CAPACITY: its tree outgrew blocks_per_game): give the game
                        # number back, drop what was recorded, and let the slot start a fresh game -- loudly
                        print("self-play slot %d (game %d) failed with engine error %d; game discarded" % (
                            s, slot_game[s], res[s]["done"]), file=sys.stderr)
                        sched.discard(slot_game.pop(s))
                        slot_resign.pop(s, None)
                        eng.records[s] = []
                        free.append(s)
                        active -= 1
                        continue
                    if res[s]["done"] != 1:
                        continue
                    gd = eng.game_data(s, res[s], model_name)
                    eng.records[s] = []        # the finished game owns its move list now
                    gd['resign_model1'] = gd['resign_model2'] = slot_resign[s]
                    g = slot_game.pop(s)
                    sched.finished(gd, slot_resign.pop(s))
                    if len(gd['moves']) == 0:
                        sched.discard(g)
                    else:
                        pending.append(submit(g, gd))
                        played += 1
                        t["moves"] += len(gd['moves'])
                    free.append(s)
                    active -= 1
                refilled = fill(free)
                active += refilled
                idle += len(free) - refilled
                reap()
                t["turnover"] += time.perf_counter() - t1
            if max_steps is not None and steps >= max_steps:
                break
        t["steps"] = steps
    finally:
        t0 = time.perf_counter()
        try:
            reap(block=True)
        finally:
            writers.shutdown(wait=True)
            t["writer_wait"] = time.perf_counter() - t0
            t["loop_s"] = time.perf_counter() - t_loop0   # first restart batch .. last sample file on disk
            t["games"] = played
            t["net_calls"], t["net_positions"] = eng.n_net_calls, eng.n_net_positions
            if stats is not None:
                stats.update(t)
            eng.close()
    return played


class _GpuWorker(Process):
    """multiprocessing.Process whose child may use the GPU whatever the parent did before start()."""

    def start(self):
        from . import predicting_queue_worker as pq
        self._conf_snapshot = dict(conf)
        self._factory_snapshot = pq._factory
        return Process.start(self)

    @staticmethod
    def _Popen(process_obj):
        from . import _lib
        method = 'spawn' if _lib.gpu_runtime_initialised() else None
        return multiprocessing.get_context(method).Process._Popen(process_obj)

    def _enter_child(self):
        """Under 'spawn' the interpreter is fresh: restore what a fork would have inherited."""
        from . import predicting_queue_worker as pq
        snap = getattr(self, '_conf_snapshot', None)
        if snap is not None:
            conf.update(snap)
        if pq._factory is None and getattr(self, '_factory_snapshot', None) is not None:
            pq.set_model_factory(self._factory_snapshot)
        # several workers share the host's cores (one per GPU): one OpenMP thread per core in EACH of them makes the small
        # host-side tensor ops of a step crawl (distributed.launch_ranks measures the same effect)
        workers = max(1, int(conf.get('N_GAME_PROCESS', 1)))
        if workers > 1:
            import torch
            torch.set_num_threads(max(1, (os.cpu_count() or 1) // workers))


class NoModelSelfPlayWorker(_GpuWorker):
    def __init__(self, process_id):
        Process.__init__(self, name='SelfPlayProcessor')
        self._process_id = process_id

    def run(self):
        try:
            self._enter_child()
            gpus = conf['GPUs']
            run_selfplay(gpus[self._process_id % len(gpus)], "BEST_SYM")
        except Exception as e:  # the reference prints and carries on (selfplay_worker.py:126-130)
            print("EXCEPTION in NoModelSelfPlayWorker!!!: %s" % e)
            traceback.print_exc(file=sys.stdout)


class SelfPlayWorker(_GpuWorker):
    def __init__(self, gpuid, forever=False, one_game_only=-1):
        Process.__init__(self, name='SelfPlayProcessor')
        self._gpuid = gpuid
        self._forever = forever
        self._one_game_only = one_game_only

    def _play(self, model):
        if conf.get('SELFPLAY_WORKER_ENGINE', 'sync') == 'device':
            n = conf['N_GAMES']
            only = self._one_game_only if self._one_game_only >= 0 else None
            return run_selfplay(self._gpuid, "BEST", n_games=n, only_game=only)
        from .self_play import model_self_play
        from .predicting_queue_worker import _NumpyNet
        # the sync game loop is host code on numpy boards (like the reference's, which talks to Keras): the resident
        # net is presented through the numpy face of the model contract
        return model_self_play(_NumpyNet(model), one_game_only=self._one_game_only)

    def run(self):
        """selfplay_worker.py:29-58."""
        try:
            self._enter_child()
            from .predicting_queue_worker import init_predicting_workers, destroy_predicting_workers, get_model
            from .simulation_workers import init_simulation_workers, destroy_simulation_workers
            if conf.get('THREAD_SIMULATION', True):
                init_simulation_workers()
            init_predicting_workers([self._gpuid])
            name = ""
            model = get_model("BEST", self._gpuid)
            while True:
                if model.name != name:
                    name = model.name
                    self._play(model)
                else:
                    print("No new best model")
                    if self._forever:
                        time.sleep(conf.get('SLEEP_SECONDS', 120))
                if not self._forever:
                    break
                destroy_predicting_workers([self._gpuid])       # drop the resident copy: reload from MODEL_DIR
                init_predicting_workers([self._gpuid])
                model = get_model("BEST", self._gpuid)
            destroy_simulation_workers()
        except Exception as e:
            print("EXCEPTION in SelfPlayWorker!!!: %s" % e)
            traceback.print_exc(file=sys.stdout)
