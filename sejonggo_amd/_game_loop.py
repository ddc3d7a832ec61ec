"""The one game loop behind self_play.play_game (self_play.py:164-290) and the host-tree form of play_game_async
(nomodel_self_play.py:142-271): both reference functions are the same loop around different evaluate / choose
primitives, so they share this body.  Quirks kept on purpose and pinned by the goldens: policy_target = the children's
priors, `player` is what make_play returned (the previous mover) from move 1 on, a move's record is appended before
the double-pass check, trees are re-rooted with stats kept."""
import numpy as np

from .conf import conf
from .play import game_init, get_winner, index2coord, make_play, new_tree


def play_loop(size, first, second, evaluate, choose, name_of, stop_exploration, self_play=False, num_moves=None,
              resign_first=None, resign_second=None, first_is_model1=True, async_winner_rule=False):
    """first/second: the handles (model objects or indicator strings) that play black/white.
    evaluate(handle, board) -> (policy[A], value); choose(board, tree, temperature, handle) -> action index."""
    board, player = game_init(size)
    moves = []
    current, other = first, second
    mcts_tree, other_mcts = None, None
    value, skipped_last, temperature, end_reason = None, False, 1, "PLAYED ALL MOVES"
    for move_n in range(size * size * 2 if num_moves is None else num_moves):
        if move_n == stop_exploration:
            temperature = 0
        policy, value = evaluate(current, board)
        resign = resign_first if current == first else resign_second
        if resign and value <= resign:
            end_reason = "resign"
            break
        if not mcts_tree or not mcts_tree['subtree']:
            mcts_tree = new_tree(policy, board, add_noise=self_play)
            if self_play:
                other_mcts = mcts_tree
        index = choose(board, mcts_tree, temperature, current)
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
        current, other = other, current
        mcts_tree, other_mcts = other_mcts, mcts_tree
    winner, black_points, white_points = get_winner(board)
    tag = {1: "B", 0: "D", -1: "W"}
    result = "%s+R" % tag[player] if end_reason == "resign" else "%s+%s" % (tag[winner], abs(black_points - white_points))
    nameB, nameW = name_of(first), name_of(second)
    if winner == 0:
        winner_model = None
    elif async_winner_rule and conf.get('COMPAT_WINNER_MODEL', True):
        # nomodel_self_play.py:247: `modelB_name if (winner == 1) == model1_isblack else modelW_name` -- right while model1
        # plays black, the LOSER's name when model1 plays white (self_play.py:258 has the correct rule).  Reproduced behind
        # conf['COMPAT_WINNER_MODEL'] because evaluate_worker.py:141 counts wins from this field.
        winner_model = nameB if (winner == 1) == first_is_model1 else nameW
    else:
        winner_model = nameB if winner == 1 else nameW
    r1, r2 = (resign_first, resign_second) if first_is_model1 else (resign_second, resign_first)
    return {'moves': moves, 'modelB_name': nameB, 'modelW_name': nameW, 'winner': {1: 1, -1: 0, 0: None}[winner],
            'winner_model': winner_model, 'result': result, 'resign_model1': r1, 'resign_model2': r2,
            'end_reason': end_reason}
