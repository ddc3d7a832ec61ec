"""The two helpers main_selfplay.py imports from the reference's utils.py (utils.py:117,147): directory
initialisation and the sweep of empty reserved game directories.  (The reference's utils imports TensorFlow
at module top, utils.py:6; nothing of that is needed on this path.)"""
import os

from .conf import conf


def init_directories():
    """utils.py:117-145 (the conf-named directories; the reference also makes a bare ./logs)."""
    for d in (conf['MODEL_DIR'], conf['LOG_DIR'], conf['EVAL_DIR'], conf['SELF_PLAY_DIR'],
              os.path.join(conf['SELF_PLAY_DIR'], "KGS"), conf['TMP_DIR']):
        try:
            os.makedirs(d)
        except OSError:
            pass


def clean_up_empty():
    """utils.py:147-160: remove reserved-but-empty game directories under EVAL_DIR and SELF_PLAY_DIR."""
    try:
        for folder in (conf['EVAL_DIR'], conf['SELF_PLAY_DIR']):
            for _dir in os.listdir(folder):
                dir_path = os.path.join(folder, _dir)
                if not os.path.isdir(dir_path):
                    continue
                for d in os.listdir(dir_path):
                    d_path = os.path.join(dir_path, d)
                    if os.path.isdir(d_path) and len(os.listdir(d_path)) == 0:
                        print("Clean up empty dir", d_path)
                        os.rmdir(d_path)
    except Exception as e:
        print("EXCEPTION WHILE CLEANING FOLDERS!!")
        print(e)


def _colour(code):
    return lambda prt: print("\033[%dm %s\033[00m" % (code, prt))


prRed, prGreen, prYellow, prLightPurple, prPurple, prCyan, prLightGray, prBlack = (_colour(c) for c in range(91, 99))
