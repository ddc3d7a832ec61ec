"""Sample writer with the reference's layout (sgfsave.py:49-79 save_self_play_data):
SELF_PLAY_DIR/<model>/game_%05d/move_%03d/sample.h5 with datasets board f32 (1,S,S,17), policy_target f32
(S*S+1), value_target f32 ().  With h5py the file is written through it, exactly like the reference.  h5py is not
installed in this image (probed): then sample.h5 is produced by the spec-following minimal writer hdf5_min.py
(superblock v2 / object header v2; not yet checked against libhdf5) and, while conf['WRITE_NPZ_TWIN'] is set, the
same three arrays are also stored as sample.npz next to it (`convert_npz_to_h5` re-writes sample.h5 through h5py
wherever that exists).  value_target keeps the reference's rule
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


def value_target(winner, player, move_n, compat=None):
    compat = conf.get('COMPAT_Z', True) if compat is None else compat
    if compat:
        return 1 if winner == player else -1          # sgfsave.py:56, quirks included
    if winner is None:
        return 0
    mover_is_black = (move_n % 2 == 0)
    black_won = (winner == 1)
    return 1 if mover_is_black == black_won else -1


def save_self_play_data(model_name, game_no, game_data):
    winner = game_data['winner']
    for move_data in game_data['moves']:
        vt = value_target(winner, move_data['player'], move_data['move_n'])
        directory = os.path.join(conf['SELF_PLAY_DIR'], model_name, "game_%05d" % game_no, "move_%03d" % move_data['move_n'])
        os.makedirs(directory, exist_ok=True)
        board = np.asarray(move_data['board'], dtype=np.float32)
        pol = np.asarray(move_data['policy'], dtype=np.float32)
        val = np.array(vt, dtype=np.float32)
        if HAVE_H5:
            import h5py
            with h5py.File(os.path.join(directory, 'sample.h5'), 'w') as f:
                f.create_dataset('board', data=board, dtype=np.float32)
                f.create_dataset('policy_target', data=pol, dtype=np.float32)
                f.create_dataset('value_target', data=val, dtype=np.float32)
        else:
            from .hdf5_min import write_datasets
            write_datasets(os.path.join(directory, 'sample.h5'),
                           {'board': board, 'policy_target': pol, 'value_target': val})
            if conf.get('WRITE_NPZ_TWIN', True):
                np.savez(os.path.join(directory, 'sample.npz'), board=board, policy_target=pol, value_target=val)


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
