"""Drop-in for the reference's nomodel_self_play.py entry points on MI355X.

play_game_async (nomodel_self_play.py:142-271) keeps its signature and its game_data result; the body is
one game slot of the device engine (engine.SelfPlayEngine: async_simulate2 :59-82, back_propagation :40-56,
select_play :114-140 all run in k_search on the GPU).  Many-game throughput comes from
selfplay_worker.NoModelSelfPlayWorker, which keeps conf['GAMES_PER_GPU'] games resident at once."""
import numpy as np

from . import _lib
from .conf import conf
from .predicting_queue_worker import get_model, put_name_request


def play_game_async(model1_indicator, model2_indicator, energy, stop_exploration, process_id, self_play=False,
                    num_moves=None, resign_model1=None, resign_model2=None, seed=None):
    if model1_indicator != model2_indicator:
        raise NotImplementedError("two-model evaluation games (evaluate_worker.py:137) are SURVEY.md §8f row 3; "
                                  "the MI355X engine currently plays single-model self-play games")
    from .engine import SelfPlayEngine
    net = get_model(model1_indicator)
    sym = "random1" if model1_indicator.endswith("_SYM") else "identity"
    eng = SelfPlayEngine(net, size=conf['SIZE'], n_games=1, sims=conf['MCTS_SIMULATIONS'], energy=energy,
                         stop_exploration=stop_exploration, num_moves=num_moves, komi=conf['KOMI'], self_play=self_play,
                         symmetry=sym, seed=(process_id if seed is None else seed))
    try:
        eng.start_games([0], resign=[resign_model1] if resign_model1 is not None else None)
        games = eng.run()
        if not games:
            raise _lib.SgoError("engine finished without a game")
        gd = games[0]
    finally:
        eng.close()
    name = put_name_request(model1_indicator)
    gd['modelB_name'] = gd['modelW_name'] = name
    gd['winner_model'] = None if gd['winner'] is None else name
    gd['resign_model1'], gd['resign_model2'] = resign_model1, resign_model2
    return gd


def select_play(board, energy, mcts_tree, temperature, model_indicator, gpuid):
    raise NotImplementedError("select_play on a host dict tree is served by the GTP row (SURVEY.md §8f row 4); "
                              "the device engine owns its trees -- use play_game_async / SelfPlayEngine")


def back_propagation(result, node):
    """nomodel_self_play.py:40-56 on host dict trees: graft the evaluated leaf at `moves`, then add its value to
    every ancestor up to the root (same value at every level: no minimax sign flip), clearing busy flags."""
    from .tree_util import get_node_by_moves
    leaf, moves = result
    parent = get_node_by_moves(node, moves[:-1])
    leaf['virtual_loss'] = 0
    parent['subtree'][moves[-1]].pop('parent', None)   # detach the placeholder
    parent['subtree'][moves[-1]] = leaf
    leaf['parent'] = parent
    while parent is not None:
        parent['count'] += 1
        parent['value'] += leaf['value']
        parent['mean_value'] = parent['value'] / float(parent['count'])
        parent['virtual_loss'] = 0
        parent = parent['parent'] if parent['parent'] else None
