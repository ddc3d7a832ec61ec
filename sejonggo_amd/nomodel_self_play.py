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
    if model1_indicator != model2_indicator or not self_play:
        # two different nets / evaluation games: separate tree per player -> the host-tree form of the same loop
        return play_game_host(model1_indicator, model2_indicator, energy, stop_exploration, process_id, self_play=self_play,
                              num_moves=num_moves, resign_model1=resign_model1, resign_model2=resign_model2)
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


def _evaluate(model_indicator, boards):
    """One batched network call for a list of [1,S,S,17] boards -> list of (policy, value)."""
    from .predicting_queue_worker import predict_batch
    pol, val = predict_batch(model_indicator, np.concatenate(boards))
    return [(pol[i], val[i][0]) for i in range(len(boards))]


def async_simulate2(node, board, model_indicator, energy, original_player, process_id):
    """nomodel_self_play.py:59-82 on a host dict tree: pick `energy` distinct leaves by busy-flag exclusion, evaluate
    them (here: ONE batched forward pass on the GPU instead of a process pool), back-propagate in launch order.
    When no leaf is selectable the oldest pending result is back-propagated first, as the reference does."""
    from .play import make_play, index2coord, new_subtree
    from .tree_util import find_best_leaf_virtual_loss
    if node['subtree'] == {}:
        return
    size = board.shape[-2]
    pending, done = [], []          # launched leaves awaiting evaluation / evaluated results in launch order

    def flush():
        if pending:
            for (leaf, moves, b), (policy, value) in zip(pending, _evaluate(model_indicator, [p[2] for p in pending])):
                new_leaf = dict(leaf, parent=None)
                new_leaf['subtree'] = new_subtree(policy, b, new_leaf)
                v = value if b[0, 0, 0, -1] == original_player else -value
                new_leaf['count'] += 1
                new_leaf['value'] += v
                new_leaf['mean_value'] = new_leaf['value'] / float(new_leaf['count'])
                done.append((new_leaf, moves))
            del pending[:]

    pre_bp = 0
    total = energy
    while energy > 0:
        leaf, moves = find_best_leaf_virtual_loss(node)
        if leaf is not None and leaf['count'] > 0:
            energy -= 1
            pre_bp += 1
            continue
        if leaf is None:
            flush()
            back_propagation(done.pop(0), node)
            pre_bp += 1
            continue
        b = np.copy(board)
        for m in moves:
            x, y = index2coord(m, size)
            make_play(x, y, b)
        pending.append((leaf, moves, b))
        energy -= 1
    flush()
    for _ in range(total - pre_bp):
        back_propagation(done.pop(0), node)


def select_play(board, energy, mcts_tree, temperature, model_indicator, gpuid):
    """nomodel_self_play.py:114-140 on a host dict tree (GTP-style single-position use)."""
    for _ in range(int(conf['MCTS_SIMULATIONS'] / conf['ENERGY'])):
        async_simulate2(mcts_tree, np.copy(board), model_indicator, energy, board[0, 0, 0, -1], gpuid)
    children = mcts_tree['subtree']
    if temperature == 1:
        total_n = sum(c['count'] for c in children.values())
        moves = [m for m, c in children.items() if c['count']]
        return np.random.choice(moves, size=1, p=[children[m]['count'] / float(total_n) for m in moves])[0]
    return max((c['count'], c['mean_value'], a) for a, c in children.items())[2]


def play_game_host(model1_indicator, model2_indicator, energy, stop_exploration, process_id, self_play=False, num_moves=None,
                   resign_model1=None, resign_model2=None):
    """play_game_async (nomodel_self_play.py:142-271) with host dict trees: the general form that also covers two
    different models (evaluate_worker.py:137: best vs latest, separate tree per player).  Rules, symmetries and
    the nets run on the GPU; one game at a time -- throughput self-play uses the device engine instead."""
    from . import play
    from ._game_loop import play_loop
    from .predicting_queue_worker import put_predict_request
    first, second = play.choose_first_player(model1_indicator, model2_indicator)      # one draw of `random`, play.py:301-306
    swap = first != model1_indicator
    r_first, r_second = (resign_model2, resign_model1) if swap else (resign_model1, resign_model2)

    def choose(board, tree, temperature, indicator):
        # looked up on the module so that a wrapped select_play (tests, instrumentation) is honoured
        return globals()["select_play"](board, energy, tree, temperature, indicator, process_id)

    return play_loop(conf['SIZE'], first, second, lambda ind, board: put_predict_request(ind, board, response_now=True),
                     choose, put_name_request, stop_exploration, self_play=self_play, num_moves=num_moves,
                     resign_first=r_first, resign_second=r_second, first_is_model1=not swap, async_winner_rule=True)


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
