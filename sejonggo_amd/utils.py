"""The two helpers main_selfplay.py imports from the reference's utils.py (utils.py:117,147): directory
initialisation and the sweep of empty reserved game directories.  (The reference's utils imports TensorFlow
at module top, utils.py:6; nothing of that is needed on this path.)"""
import os

from .conf import conf


def init_directories():
    for k in ('MODEL_DIR', 'SELF_PLAY_DIR'):
        os.makedirs(conf[k], exist_ok=True)


def clean_up_empty():
    root = conf['SELF_PLAY_DIR']
    if not os.path.isdir(root):
        return
    for model in os.listdir(root):
        mdir = os.path.join(root, model)
        if not os.path.isdir(mdir):
            continue
        for game in os.listdir(mdir):
            gdir = os.path.join(mdir, game)
            if os.path.isdir(gdir) and not os.listdir(gdir):
                os.rmdir(gdir)
