"""Drop-in for the reference's go_game.py (go_game.py:9-31): the GoGame wrapper whose do_move places a stone through
make_play -- i.e. through libsgo_hip.so's board_advance on MI355X."""
from .conf import conf
from .play import game_init, index2coord, make_play

WHITE = -1
BLACK = +1
EMPTY = 0
RESIGN = "resign"
PASS = "pass"


class GoGame(object):
    def __init__(self, size=9, komi=7.5):
        # the reference ignores `size` for the board itself (game_init reads conf['SIZE'], go_game.py:11)
        self.board, player = game_init()
        self.current_player = player
        self.size = size
        self.ko = None
        self.komi = komi
        self.handicaps = []
        self.history = []
        self.num_black_prisoners = 0
        self.num_white_prisoners = 0
        self.is_end_of_game = False
        self.passes_white = 0
        self.passes_black = 0

    def do_move(self, action, color):
        """go_game.py:25-30: `action` is an action index (S*S = skip), RESIGN or "pass" (both no-ops here);
        color None = self.current_player (which the reference never advances)."""
        if color is None:
            color = self.current_player
        if action != RESIGN and action != "pass":
            x, y = index2coord(action, self.board.shape[-2])
            make_play(x, y, self.board, color)


class IllegalMove(Exception):
    pass
