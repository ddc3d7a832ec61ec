"""Sample writer with the reference's layout (sgfsave.py:49-79 save_self_play_data):
SELF_PLAY_DIR/<model>/game_%05d/move_%03d/sample.h5 with datasets board f32 (1,S,S,17), policy_target f32
(S*S+1), value_target f32 ().  Writers, in order of preference: h5py (exactly the reference's calls); h5lite, this
package's ctypes binding to the HDF5 C library (what h5py wraps; h5py is not installed in this image, libhdf5 is); the
pure-Python hdf5_min.py (superblock v2 / object header v2, verified readable by libhdf5: tests/test_hdf5.py), which
also keeps a sample.npz twin while conf['WRITE_NPZ_TWIN'] is set.  value_target keeps the reference's rule
`1 if winner == player else -1` (sgfsave.py:56) when conf['COMPAT_Z'] is set (default), and the corrected
outcome-from-mover's-view otherwise."""
import os

import numpy as np

from .conf import conf

try:
    import h5py  # noqa: F401
    HAVE_H5 = True
except Exception:
    HAVE_H5 = False


def _h5lite():
    """The HDF5 C library through this package's ctypes binding, or None."""
    from . import h5lite
    return h5lite if h5lite.available() else None


def value_target(winner, player, move_n, compat=None):
    compat = conf.get('COMPAT_Z', True) if compat is None else compat
    if compat:
        return 1 if winner == player else -1          # sgfsave.py:56, quirks included
    if winner is None:
        return 0
    mover_is_black = (move_n % 2 == 0)
    black_won = (winner == 1)
    return 1 if mover_is_black == black_won else -1


def _write_sample(directory, move_data, winner):
    vt = value_target(winner, move_data['player'], move_data['move_n'])
    _write_sample_arrays(directory, np.asarray(move_data['board'], dtype=np.float32), np.asarray(move_data['policy'], dtype=np.float32),
                         np.array(vt, dtype=np.float32))


def _write_sample_arrays(directory, board, pol, val):
    """sample.h5 of one position: board f32 (1,S,S,17), policy_target f32 (S*S+1), value_target f32 ()."""
    lite = None if HAVE_H5 else _h5lite()
    if HAVE_H5 or lite is not None:
        if HAVE_H5:
            import h5py
            session = h5py.File(os.path.join(directory, 'sample.h5'), 'w')
        else:
            session = lite.open(os.path.join(directory, 'sample.h5'), 'w')    # holds the library lock: writer THREADS call this
        with session as f:                                                     # the same three calls as sgfsave.py:76-78
            f.create_dataset('board', data=board, dtype=np.float32)
            f.create_dataset('policy_target', data=pol, dtype=np.float32)
            f.create_dataset('value_target', data=val, dtype=np.float32)
    else:
        from .hdf5_min import write_datasets
        write_datasets(os.path.join(directory, 'sample.h5'),
                       {'board': board, 'policy_target': pol, 'value_target': val})
        if conf.get('WRITE_NPZ_TWIN', True):
            np.savez(os.path.join(directory, 'sample.npz'), board=board, policy_target=pol, value_target=val)


def _make_move_dir(root, model_name, pattern, game_no, move_n):
    """sgfsave.py:23-33 / :59-73: a move directory that already exists means another writer owns that game number;
    the reference then bumps the game number until the directory can be created (and keeps the bumped number for the
    rest of the game)."""
    while True:
        directory = os.path.join(root, model_name, pattern % game_no, "move_%03d" % move_n)
        try:
            os.makedirs(directory)
            return directory, game_no
        except FileExistsError:          # only "somebody owns this number": a full or read-only disk must surface, not spin
            game_no += 1


def save_self_play_data(model_name, game_no, game_data):
    """sgfsave.py:49-79."""
    winner = game_data['winner']
    for move_data in game_data['moves']:
        directory, game_no = _make_move_dir(conf['SELF_PLAY_DIR'], model_name, "game_%05d", game_no, move_data['move_n'])
        _write_sample(directory, move_data, winner)
    if conf.get('SGF_ENABLED'):
        save_game_sgf(model_name, game_no, game_data)


def write_packed_game(conf_items, model_name, game_no, size, packed, policies, players, move_ns, winner):
    """The sample files of one self-play game from its PACKED move records (768-byte positions; what a writer PROCESS
    receives: a few hundred KB per game instead of 24.5 KB board tensors per move).  Top-level and torch-free so that a
    'spawn'ed writer process can import and run it; `conf_items` carries the caller's conf entries the writer reads."""
    from .engine import unpack_positions
    if conf_items:
        conf.update(conf_items)
    boards = unpack_positions(packed, size)
    for i in range(len(move_ns)):
        directory, game_no = _make_move_dir(conf['SELF_PLAY_DIR'], model_name, "game_%05d", game_no, int(move_ns[i]))
        vt = value_target(winner, int(players[i]), int(move_ns[i]))
        _write_sample_arrays(directory, boards[i:i + 1].astype(np.float32), np.asarray(policies[i], dtype=np.float32),
                             np.array(vt, dtype=np.float32))
    return len(move_ns)


def save_file(model_name, game_n, move_data, winner, game_name="game"):
    """sgfsave.py:16-38: one move of an evaluation / plain game under conf['GAMES_DIR']."""
    directory, _ = _make_move_dir(conf['GAMES_DIR'], model_name, game_name + "_%03d", game_n, move_data['move_n'])
    _write_sample(directory, move_data, winner)


def save_game_data(model_name, game_n, game_data, game_name="game"):
    """sgfsave.py:41-46."""
    winner = game_data['winner']
    for move_data in game_data['moves']:
        save_file(model_name, game_n, move_data, winner, game_name)
    if conf.get('SGF_ENABLED'):
        save_game_sgf(model_name, game_n, game_data)


def _sgf_point(x, y, size):
    """SGF coordinates of the reference's (x, y); its files come out of sgfmill with row 0 at the top, so the point
    written for (x, y) is column x, row y counted from the top (sgfsave.py:141: `(SIZE - 1 - y, x)` in sgfmill's
    bottom-up rows)."""
    if y == size:
        return ""
    return "abcdefghijklmnopqrs"[x] + "abcdefghijklmnopqrs"[y]


def save_game_sgf(model_name, game_n, game_data):
    """sgfsave.py:129-167 without sgfmill: the same properties (PB, PW, KM, RE, one B[]/W[] node per move with the
    value comment) serialised directly.  Parity unpinned: sgfmill is absent here, so the byte-level layout of its
    serialiser (line folding, property order beyond FF/GM/SZ) is not reproduced."""
    size = conf['SIZE']
    from .play import get_real_board

    def esc(t):
        return t.replace("\\", "\\\\").replace("]", "\\]")

    out = ["(;FF[4]GM[1]SZ[%d]PB[%s]PW[%s]KM[%s]RE[%s]" % (size, esc(game_data['modelB_name']), esc(game_data['modelW_name']),
                                                          conf['KOMI'], esc(game_data['result']))]
    moves = game_data['moves']
    for move_data in moves:
        color = 'B' if move_data['player'] == 1 else 'W'
        x, y = move_data['move']
        nxt = moves[(move_data['move_n'] + 1) % len(moves)]['board']
        comment = "Value %s\n %s" % (move_data['value'], get_real_board(nxt))
        out.append(";%s[%s]C[%s]" % (color, _sgf_point(x, y, size), esc(comment)))
    out.append(")\n")
    os.makedirs(os.path.join(conf['GAMES_DIR'], model_name), exist_ok=True)
    filename = os.path.join(conf['GAMES_DIR'], model_name, "game_%03d.sgf" % game_n)
    while os.path.isfile(filename):
        game_n += 1
        filename = os.path.join(conf['GAMES_DIR'], model_name, "game_%03d.sgf" % game_n)
    with open(filename, "wb") as f:
        f.write("".join(out).encode("utf-8"))
    return filename


def convert_npz_to_h5(root):
    import h5py
    n = 0
    for d, _, files in os.walk(root):
        if 'sample.npz' in files:
            z = np.load(os.path.join(d, 'sample.npz'))
            with h5py.File(os.path.join(d, 'sample.h5'), 'w') as f:
                for k in ('board', 'policy_target', 'value_target'):
                    f.create_dataset(k, data=z[k], dtype=np.float32)
            n += 1
    return n


def _short_games(model_self_play_dir, min_move):
    for game_dir in sorted(os.listdir(model_self_play_dir)):
        real_path = os.path.join(model_self_play_dir, game_dir)
        try:
            n = len(os.listdir(real_path))
        except OSError:
            continue
        if n < min_move:
            yield game_dir, real_path, n


def clean_up(self_play_dir, min_move):
    """sgfsave.py:83-96: remove every self-play game with fewer than `min_move` moves; returns how many."""
    import shutil
    total = 0
    for model_dir in os.listdir(self_play_dir):
        for _, real_path, _n in list(_short_games(os.path.join(self_play_dir, model_dir), min_move)):
            shutil.rmtree(real_path)
            total += 1
    return total


def statistic_by_model(model_self_play_dir, min_move):
    """sgfsave.py:114-127: {number of moves: [game directories]} for the games shorter than min_move."""
    stat = {i: [] for i in range(min_move)}
    for game_dir, _, n in _short_games(model_self_play_dir, min_move):
        stat[n].append(game_dir)
    return stat


def statistic_all_model(self_play_dir, min_move):
    """sgfsave.py:99-112."""
    stat = {i: [] for i in range(min_move)}
    for model_dir in os.listdir(self_play_dir):
        for game_dir, _, n in _short_games(os.path.join(self_play_dir, model_dir), min_move):
            stat[n].append(game_dir)
    return stat
