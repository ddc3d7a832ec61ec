"""Drop-in for the reference's sync self-play path (self_play.py): simulate (:28-120), mcts_decision (:123-152),
select_play (:154-162), play_game (:164-290).  This is the path the reference's own MCTS unit tests target
(test/tests.py:684-1068) and that SelfPlayWorker / evaluator use with an in-process model.  Trees are host dicts
(see play.py); every rules call (make_play, legal_moves through new_subtree, get_winner) and every symmetry
transform runs on the MI355X through libsgo_hip.so.  Many-game throughput is the business of
engine.SelfPlayEngine, not of this module."""
import numpy as np

from .conf import conf
from .play import index2coord, make_play, new_subtree, top_n_actions, top_one_action
from .symmetry import random_symmetry_predict


def _backup(leaf, value):
    node = leaf
    while node is not None:
        node['count'] += 1
        node['value'] += value
        node['mean_value'] = node['value'] / float(node['count'])
        node = node['parent'] if node['parent'] else None


def simulate(node, board, model, mcts_batch_size, original_player):
    """One batch of leaves below `node`: if the best child is a leaf, the top `mcts_batch_size` children are each
    followed down (by top_one_action) to a leaf, all leaves are evaluated in ONE predict call, expanded and backed
    up; otherwise play the best child's move and recurse."""
    size = board.shape[-2]
    picks = top_n_actions(node['subtree'], mcts_batch_size)
    best = picks[0]
    if node['subtree'][best['action']]['subtree'] != {}:
        x, y = index2coord(best['action'], size)
        make_play(x, y, board)
        return simulate(node['subtree'][best['action']], board, model, mcts_batch_size, original_player)
    boards = np.zeros((len(picks), size, size, 17), dtype=np.int32)
    for i, pick in enumerate(picks):
        b = np.copy(board)
        x, y = index2coord(pick['action'], size)
        make_play(x, y, b)
        leaf = pick['node']
        while leaf['subtree'] != {}:
            step = top_one_action(leaf['subtree'])
            leaf = step['node']
            x, y = index2coord(step['action'], size)
            make_play(x, y, b)
        pick['node'] = leaf
        boards[i] = b[0]
    policies, values = random_symmetry_predict(model, boards.astype(np.float32))   # the reference feeds float32
    for policy, v, b, pick in zip(policies, values, boards, picks):
        value = v[0] if b[0, 0, -1] == original_player else -v[0]
        leaf = pick['node']
        leaf['subtree'] = new_subtree(policy, b.reshape((1,) + b.shape), leaf)
        _backup(leaf, value)


def mcts_decision(policy, board, mcts_simulations, mcts_tree, temperature, model):
    if mcts_simulations is None:
        mcts_simulations = conf['MCTS_SIMULATIONS']
    for _ in range(int(mcts_simulations / conf['MCTS_BATCH_SIZE'])):
        simulate(mcts_tree, np.copy(board), model, conf['MCTS_BATCH_SIZE'], board[0, 0, 0, -1])
    children = mcts_tree['subtree']
    if temperature == 1:
        total_n = sum(c['count'] for c in children.values())
        moves = [m for m, c in children.items() if c['count']]
        ps = [children[m]['count'] / float(total_n) for m in moves]
        return np.random.choice(moves, size=1, p=ps)[0]
    return max((c['count'], c['mean_value'], a) for a, c in children.items())[2]


def select_play(policy, board, mcts_simulations, mcts_tree, temperature, model):
    return mcts_decision(policy, board, mcts_simulations, mcts_tree, temperature, model)


def play_game(model1, model2, mcts_simulations, stop_exploration, self_play=False, num_moves=None, resign_model1=None,
              resign_model2=None):
    """self_play.py:164-290.  Who plays black is a coin flip (choose_first_player, play.py:301-306)."""
    from . import play
    from ._game_loop import play_loop
    first, second = play.choose_first_player(model1, model2)     # always one draw of `random` (self_play.py:168)
    swap = first is not model1
    r_first, r_second = (resign_model2, resign_model1) if swap else (resign_model1, resign_model2)

    def evaluate(model, board):
        policies, values = model.predict_on_batch(board)
        return policies[0], values[0]

    def choose(board, tree, temperature, model):
        return select_play(None, board, mcts_simulations, tree, temperature, model)

    return play_loop(conf['SIZE'], first, second, evaluate, choose, lambda m: m.name, stop_exploration, self_play=self_play,
                     num_moves=num_moves, resign_first=r_first, resign_second=r_second, first_is_model1=not swap)


def _resign_bookkeeping(game_data, resign, min_values, current_resign):
    """self_play.py:318-329: only no-resign games calibrate the threshold; min_values stays in arrival order."""
    if resign is None:
        moves = game_data['moves']
        own = moves[::2] if game_data['winner'] == 1 else moves[1::2]
        min_values.append(min([m['value'] for m in own]))
        idx = int(conf['RESIGNATION_ALLOWED_ERROR'] * len(min_values))
        if idx > 0:
            current_resign = min_values[idx]
    return current_resign


def model_self_play(model, one_game_only=-1):
    """self_play.py:292-339: N_GAMES sync self-play games of `model`, skipping game numbers whose directory exists;
    `one_game_only` >= 0 plays exactly that game number."""
    import os
    from random import random
    from .sgfsave import save_self_play_data
    games_data, current_resign, min_values = [], None, []
    for game in range(conf['N_GAMES']):
        if 0 <= one_game_only and game != one_game_only:
            continue
        directory = os.path.join(conf['SELF_PLAY_DIR'], model.name, "game_%05d" % game)
        if os.path.isdir(directory):
            continue
        os.makedirs(directory)
        resign = current_resign if random() > conf['RESIGNATION_PERCENT'] else None
        game_data = play_game(model, model, conf['MCTS_SIMULATIONS'], conf['STOP_EXPLORATION'], self_play=True,
                              resign_model1=resign, resign_model2=resign)
        current_resign = _resign_bookkeeping(game_data, resign, min_values, current_resign)
        save_self_play_data(model.name, game, game_data)
        games_data.append(game_data)
        if one_game_only >= 0:
            break
    return games_data


def self_play(model, n_games, mcts_simulations):
    """self_play.py:342-379: n_games sync self-play games saved with save_game_data (GAMES_DIR)."""
    from random import random
    from .sgfsave import save_game_data
    games_data, current_resign, min_values = [], None, []
    for game in range(n_games):
        resign = current_resign if random() > conf['RESIGNATION_PERCENT'] else None
        game_data = play_game(model, model, mcts_simulations, conf['STOP_EXPLORATION'], self_play=True,
                              resign_model1=resign, resign_model2=resign)
        current_resign = _resign_bookkeeping(game_data, resign, min_values, current_resign)
        save_game_data(model.name, game, game_data)
        games_data.append(game_data)
    return games_data
