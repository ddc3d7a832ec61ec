"""GoGame: a board held on behalf of a caller, moved through libsgo_hip.so's board_advance.

API row: the reference's go_game.py:9-31 (constructor arguments, attribute names, `do_move(action, color)`,
`IllegalMove`).  No caller exists in the reference; the class is kept because `north_star` names the file.
Behaviour that follows the reference: the board has conf['SIZE'] points a side whatever `size` says
(go_game.py:11 calls game_init() without it), `current_player` is set once and never advanced, and
RESIGN / "pass" strings leave the board untouched (the PASS *move* is the action index S*S).
"""
from . import play

WHITE, EMPTY, BLACK = -1, 0, +1
RESIGN, PASS = "resign", "pass"

_BOOKKEEPING = dict(ko=None, num_black_prisoners=0, num_white_prisoners=0, is_end_of_game=False,
                    passes_white=0, passes_black=0)


class IllegalMove(Exception):
    """Declared by the reference (go_game.py:32), raised nowhere in it; occupied points are make_play's assert."""


class GoGame(object):
    def __init__(self, size=9, komi=7.5):
        self.size, self.komi = size, komi
        self.board, self.current_player = play.game_init()
        self.handicaps, self.history = [], []
        for field, start in _BOOKKEEPING.items():      # fields the reference declares and never updates
            setattr(self, field, start)

    def do_move(self, action, color):
        """Place `color`'s stone (None: the colour of the first mover) at action index `action` on the device;
        returns the mover's colour, or None for the two string no-ops."""
        if action in (RESIGN, PASS):
            return None
        side = self.board.shape[-2]
        x, y = play.index2coord(action, side)
        _, mover = play.make_play(x, y, self.board, self.current_player if color is None else color)
        return mover
