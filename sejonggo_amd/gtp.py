"""GTP front-end with the reference's engine classes (sejonggo_nomodel.py:20-185): SejongGoEngine (play / genmove
on a persistent MCTS tree) and GTPEngine (name, version, protocol_version, list_commands, boardsize, komi, play,
genmove, clear_board, parse_command).  Single-game latency mode of the same path: the tree is a host dict tree
searched by nomodel_self_play.select_play (leaves of a round are evaluated in one batched forward pass), rules and
net run on the MI355X.  Parity: the move-coordinate text mapping is pinned by the tests; search results follow the
host async path, which is pinned by the reference's goldens."""
import string
import sys

import numpy as np

from . import __version__
from .conf import conf
from .nomodel_self_play import select_play
from .play import coord2index, game_init, index2coord, make_play, new_tree
from .predicting_queue_worker import destroy_predicting_workers, get_model, init_predicting_workers, put_predict_request

COLOR_TO_PLAYER = {'B': 1, 'W': -1, 'b': 1, 'w': -1, 'black': 1, 'white': -1}


class SejongGoEngine(object):
    def __init__(self, mcts_simulations, board, resign=None, temperature=0, add_noise=False, process_id=0):
        self.mcts_simulations = mcts_simulations
        self.resign = resign
        self.temperature = temperature
        self.board = board
        self.add_noise = add_noise
        self.mcts_tree = None
        self.move = 1
        self.process_id = process_id
        self.player = board[0, 0, 0, -1]
        self.model_indicator = "BEST"
        init_predicting_workers(conf['GPUs'][:1])

    @property
    def model(self):
        return get_model(self.model_indicator)

    def close(self):
        destroy_predicting_workers(conf['GPUs'][:1])

    def set_temperature(self, temperature):
        self.temperature = temperature

    def play(self, color, x, y, update_tree=True):
        size = self.board.shape[-2]
        index = coord2index(x, y, size)
        if update_tree:
            if self.mcts_tree and index in self.mcts_tree['subtree']:
                self.mcts_tree = self.mcts_tree['subtree'][index]
                self.mcts_tree['parent'] = None
            else:
                self.mcts_tree = None
        self.board, self.player = make_play(x, y, self.board, color)
        self.move += 1
        return self.board, self.player

    def genmove(self, color):
        size = self.board.shape[-2]
        policy, value = put_predict_request(self.model_indicator, self.board, response_now=True)
        if self.resign and value <= self.resign:
            return 0, size + 1, policy, value, self.board, self.player
        if not self.mcts_tree or not self.mcts_tree['subtree']:
            self.mcts_tree = new_tree(policy, self.board, add_noise=self.add_noise)
        keep = conf['MCTS_SIMULATIONS']
        conf['MCTS_SIMULATIONS'] = self.mcts_simulations
        try:
            index = select_play(self.board, conf['ENERGY'], self.mcts_tree, self.temperature, self.model_indicator, self.process_id)
        finally:
            conf['MCTS_SIMULATIONS'] = keep
        x, y = index2coord(index, size)
        policy_target = np.zeros(size * size + 1)
        for a, child in self.mcts_tree['subtree'].items():
            policy_target[a] = child['p']
        self.board, self.player = self.play(color, x, y)
        return x, y, policy_target, value, self.board, self.player


class GTPEngine(object):
    def __init__(self, engine=None):
        self._komi = 0
        self.size = conf['SIZE']
        self.board, self.player = game_init(self.size)
        self.sejong_engine = engine or SejongGoEngine(conf['MCTS_SIMULATIONS'], self.board)

    def name(self):
        return "SejongGo - {} - {} simulations".format(self.sejong_engine.model.name, conf['MCTS_SIMULATIONS'])

    def version(self):
        return __version__

    def protocol_version(self):
        return "2"

    def list_commands(self):
        return "\n".join(["name", "version", "protocol_version", "list_commands", "boardsize", "komi", "play", "genmove",
                          "clear_board", "quit"])

    def boardsize(self, size):
        if int(size) != self.size:
            raise Exception("The board size in configuration is {0}x{0} but GTP asked to play {1}x{1}".format(self.size, int(size)))
        return ""

    def komi(self, komi):
        self._komi = komi
        return ""

    def parse_move(self, move):
        """GTP vertex -> (x, y): letters skip 'I', rows count from the bottom (sejonggo_nomodel.py:104-120)."""
        if move.lower() == 'pass':
            return 0, self.size
        x = string.ascii_uppercase.index(move[0].upper())
        if x >= 9:
            x -= 1
        y = int(move[1:]) - 1
        return x, self.size - y - 1

    def print_move(self, x, y):
        if y >= self.size:
            return "pass" if y == self.size else "resign"
        row = self.size - y - 1
        if x >= 8:
            x += 1
        return string.ascii_uppercase[x] + str(row + 1)

    def play(self, color, move):
        x, y = self.parse_move(move)
        self.board, self.player = self.sejong_engine.play(COLOR_TO_PLAYER[color], x, y)
        return ""

    def genmove(self, color):
        x, y, _, _, self.board, self.player = self.sejong_engine.genmove(COLOR_TO_PLAYER[color])
        return self.print_move(x, y)

    def clear_board(self):
        self.board, self.player = game_init(self.size)
        self.sejong_engine.board = self.board
        self.sejong_engine.mcts_tree = None
        self.sejong_engine.move = 1
        return ""

    def quit(self):
        return ""

    def parse_command(self, line):
        tokens = line.strip().split(" ")
        if not tokens or not tokens[0]:
            return ""
        method = getattr(self, tokens[0], None)
        if method is None or tokens[0].startswith("_"):
            return "? unknown command\n\n"
        result = method(*tokens[1:])
        return "=\n\n" if not result.strip() else "= " + result + "\n\n"


def main(inp=sys.stdin, out=sys.stdout):
    engine = GTPEngine()
    for line in inp:
        for cmd in line.split("\n"):
            res = engine.parse_command(cmd)
            if res.strip():
                out.write(res)
                out.flush()
            if cmd.strip() == "quit":
                engine.sejong_engine.close()
                return


if __name__ == "__main__":
    main()
