"""Drop-in for the reference's sync self-play path (self_play.py): simulate (:28-120), mcts_decision (:123-152),
select_play (:154-162), play_game (:164-290).  This is the path the reference's own MCTS unit tests target
(test/tests.py:684-1068) and that SelfPlayWorker / evaluator use with an in-process model.  Trees are host dicts
(see play.py); every rules call (make_play, legal_moves through new_subtree, get_winner) and every symmetry
transform runs on the MI355X through libsgo_hip.so.  Many-game throughput is the business of
engine.SelfPlayEngine, not of this module."""
import numpy as np

from .conf import conf
from .play import (game_init, get_winner, index2coord, make_play, new_subtree, new_tree, top_n_actions, top_one_action)
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
    size = conf['SIZE']
    board, player = game_init(size)
    moves = []
    current_model, other_model = model1, model2          # choose_first_player is a coin flip between equals in self-play
    if not self_play and np.random.random() >= .5:
        current_model, other_model = model2, model1
    model1_isblack = current_model is model1
    mcts_tree, other_mcts = None, None
    value, skipped_last, temperature, end_reason = None, False, 1, "PLAYED ALL MOVES"
    for move_n in range(size * size * 2 if num_moves is None else num_moves):
        if move_n == stop_exploration:
            temperature = 0
        policies, values = current_model.predict_on_batch(board)
        policy, value = policies[0], values[0]
        resign = resign_model1 if current_model is model1 else resign_model2
        if resign and value <= resign:
            end_reason = "resign"
            break
        if not mcts_tree or not mcts_tree['subtree']:
            mcts_tree = new_tree(policy, board, add_noise=self_play)
            if self_play:
                other_mcts = mcts_tree
        index = select_play(policy, board, mcts_simulations, mcts_tree, temperature, current_model)
        x, y = index2coord(index, size)
        policy_target = np.zeros(size * size + 1)
        for a, child in mcts_tree['subtree'].items():
            policy_target[a] = child['p']
        moves.append({'board': np.copy(board), 'policy': policy_target, 'value': value, 'move': (x, y), 'move_n': move_n,
                      'player': player})
        if skipped_last and y == size:
            end_reason = "BOTH_PASSED"
            break
        skipped_last = y == size
        if self_play or (other_mcts and index in other_mcts['subtree']):
            other_mcts = other_mcts['subtree'][index]
            other_mcts['parent'] = None
        mcts_tree = mcts_tree['subtree'][index]
        mcts_tree['parent'] = None
        board, player = make_play(x, y, board)
        current_model, other_model = other_model, current_model
        mcts_tree, other_mcts = other_mcts, mcts_tree
    winner, black_points, white_points = get_winner(board)
    tag = {1: "B", 0: "D", -1: "W"}
    result = "%s+R" % tag[player] if end_reason == "resign" else "%s+%s" % (tag[winner], abs(black_points - white_points))
    modelB, modelW = (model1, model2) if model1_isblack else (model2, model1)
    winner_model = None if winner == 0 else (model1 if (winner == 1) == model1_isblack else model2)
    return {'moves': moves, 'modelB_name': modelB.name, 'modelW_name': modelW.name, 'winner': {1: 1, -1: 0, 0: None}[winner],
            'winner_model': None if winner_model is None else winner_model.name, 'result': result,
            'resign_model1': resign_model1, 'resign_model2': resign_model2}
