"""Drop-in for the asynchronous half of the reference's evaluate_worker.py: NoModelEvaluateWorker(gpuid, task)
(evaluate_worker.py:97-157) -- EVALUATE_N_GAMES games of the best model against the latest one,
play_game_async("BEST_SYM", "LATEST_SYM", ENERGY, stop_exploration=0), every game saved as training data
(save_game_data(..., game_name="eval_game")) and its result recorded as a file named after the winner under
EVAL_DIR/<latest>/game_%03d/ (save_eval_game), which evaluator.promote_best_model later counts.

On MI355X the games of one worker run CONCURRENTLY as two-model slots of the device engine (engine.SelfPlayEngine with
net2: two trees per slot inside k_search, every evaluation row tagged with the model to move), conf['GAMES_PER_GPU'] at a
time.  Who plays black is the reference's coin per game (play.choose_first_player).

conf['COMPAT_LATEST_SYM'] (default on) reproduces predicting_queue_worker.py:92: LATEST_SYM requests are answered by the BEST
model, i.e. the reference's evaluation games are best-vs-best under the latest model's name.  Switch it off to evaluate the
latest model for real."""
import os
import sys
import traceback
from multiprocessing import Process

from .conf import conf
from .selfplay_worker import _GpuWorker


def run_evaluation(gpu_id, n_games=None, games_per_gpu=None, engine_kwargs=None):
    """The worker body.  Returns (wins of the latest model, games played)."""
    from .engine import SelfPlayEngine
    from .evaluator import save_eval_game
    from .predicting_queue_worker import init_predicting_workers, get_model, put_name_request
    from .sgfsave import save_game_data
    init_predicting_workers([gpu_id])
    best_name, latest_name = put_name_request("BEST_NAME"), put_name_request("LATEST_NAME")
    if latest_name == best_name:
        print("BEST MODEL and LAST MODEL are the same!! Quitting")
        return 0, 0
    n_games = conf['EVALUATE_N_GAMES'] if n_games is None else n_games
    if conf.get('COMPAT_LATEST_SYM', True):
        print("note: COMPAT_LATEST_SYM is on -- LATEST_SYM requests are answered by the best model "
              "(predicting_queue_worker.py:92); the games below are %s against itself, recorded under %s" % (best_name, latest_name),
              file=sys.stderr)
    net1, net2 = get_model("BEST_SYM", gpu_id), get_model("LATEST_SYM", gpu_id)

    class _Named(object):          # the engine labels colours by the nets' names: the latest model's name, whatever answers
        def __init__(self, net, name):
            self.net, self.name = net, name
            self.in_channels = getattr(net, "in_channels", 17)

        def predict_on_batch(self, X):
            return self.net.predict_on_batch(X)

    G = min(games_per_gpu or conf['GAMES_PER_GPU'], max(1, n_games))
    kw = dict(size=conf['SIZE'], n_games=G, sims=conf['MCTS_SIMULATIONS'], energy=conf['ENERGY'], stop_exploration=0,
              komi=conf['KOMI'], symmetry=conf.get('SYMMETRY_MODE', 'random1'), device=gpu_id, seed=gpu_id, raise_on_error=False)
    kw.update(engine_kwargs or {})
    eng = SelfPlayEngine(_Named(net1, best_name), net2=_Named(net2, latest_name), **kw)
    next_game = [0]

    def reserve():
        """evaluate_worker.py:123-130: the next game number whose directory could be created."""
        while next_game[0] < n_games:
            g = next_game[0]
            next_game[0] += 1
            d = os.path.join(conf['EVAL_DIR'], latest_name, "game_%03d" % g)
            if os.path.isdir(d):
                continue
            try:
                os.makedirs(d)
            except Exception:
                continue
            return g
        return None

    slot_game = {}

    def fill(slots):
        start = []
        for s in slots:
            g = reserve()
            if g is None:
                continue
            slot_game[s] = g
            start.append(s)
        if start:
            eng.start_eval_games(start, ids=[slot_game[s] for s in start])
        return len(start)

    wins = total = 0
    try:
        active = fill(range(G))
        idle = G - active
        while active > 0:
            st = eng.step()
            if st.n_records >= G:
                eng.drain()
            if st.n_done > idle or (st.error and st.error_game in slot_game):
                eng.drain()
                res = eng.results()
                free = []
                for s in list(slot_game):
                    if res[s]["done"] == 0:
                        continue
                    g = slot_game.pop(s)
                    free.append(s)
                    active -= 1
                    if res[s]["done"] < 0:
                        print("evaluation slot %d (game %d) failed with engine error %d; game dropped" % (s, g, res[s]["done"]),
                              file=sys.stderr)
                        eng.records[s] = []
                        continue
                    gd = eng.game_data(s, res[s])
                    eng.records[s] = []
                    if gd['winner_model'] == latest_name:
                        wins += 1
                    total += 1
                    save_game_data(latest_name, g, gd, game_name="eval_game")
                    save_eval_game(latest_name, g, gd['winner_model'])
                refilled = fill(free)
                active += refilled
                idle += len(free) - refilled
    finally:
        eng.close()
    return wins, total


class NoModelEvaluateWorker(_GpuWorker):
    def __init__(self, gpuid, task="evaluate"):
        Process.__init__(self, name='EvaluateProcessor')
        self._gpuid = gpuid
        self._task = task

    def run(self):
        if self._task == "promote_best_model":
            from .evaluator import promote_best_model
            self._enter_child()
            promote_best_model()
            return
        try:
            self._enter_child()
            run_evaluation(self._gpuid)
        except Exception as e:
            print("EXCEPTION IN NO MODEL EVALUATION WORKER!!!: %s" % e)
            traceback.print_exc()
