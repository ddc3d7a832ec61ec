"""Model loading with the reference's names (model.py:125-163 load_best_model / load_latest_model /
load_model_by_name, :97-122 create_initial_model).

A model is a net.PolicyValueNet on the CPU (the worker that owns a GPU moves it there, see
predicting_queue_worker.get_model).  Files under conf['MODEL_DIR']:

* `<stem>.pt`  -- a torch checkpoint {"state_dict", "name", "size", "n_blocks", "channels"} written by
  `save_model` (read with weights_only=True);
* `<stem>.h5`  -- the reference's Keras file (model.py:147-157): read through keras_import.load_keras_h5 (h5py, or the
  HDF5 C library through h5lite).  With neither available an existing .h5 is an ERROR (never a silent random-init
  fallback: a worker would otherwise write self-play data from random weights under the wrong model directory).

`load_best_model` follows model.py:147-157: conf['BEST_MODEL'] if present, otherwise the initial model `model_1`
is created (random init), saved as `model_1.pt` and as the best model, with a loud warning -- the reference does the
same through create_initial_model(name="model_1").  `load_latest_model` follows model.py:125-144: the file whose
name ends in the highest `_<n>`.  `model_name(kind)` returns a model's name from file metadata only; it never
builds a network and never touches a GPU (main_selfplay.main() asks for the name in the parent before it forks)."""
import os
import sys

import torch

from .conf import conf
from .net import PolicyValueNet

INITIAL_MODEL_NAME = "model_1"      # model.py:155


def _dir():
    return conf['MODEL_DIR']


def _best_stem():
    return os.path.splitext(conf['BEST_MODEL'])[0]


def _find(stem):
    """Path of `<stem>.pt` or `<stem>.h5` under MODEL_DIR, or None.  When both exist the one written LAST wins (the
    reference's trainer / evaluator write .h5 files next to our .pt ones: a promoted best_model.h5 must not be shadowed
    by a stale best_model.pt); on equal timestamps the torch checkpoint."""
    found = [p for p in (os.path.join(_dir(), stem + ext) for ext in (".pt", ".h5")) if os.path.isfile(p)]
    if not found:
        return None
    return max(found, key=lambda p: (os.stat(p).st_mtime_ns, p.endswith(".pt")))


def drop_sibling(path):
    """A model lives in ONE file: after `path` was (re)written, remove the same stem under the other extension."""
    stem, ext = os.path.splitext(path)
    other = stem + (".h5" if ext == ".pt" else ".pt")
    if os.path.isfile(other):
        os.remove(other)


def _latest_stem():
    """model.py:126-135: the file whose stem ends in the largest integer after the last '_'."""
    index, stem = -1, None
    if not os.path.isdir(_dir()):
        return None
    for filename in sorted(os.listdir(_dir())):
        if not filename.endswith((".pt", ".h5")):
            continue
        name = filename.split('.')[0]
        try:
            i = int(name.split('_')[-1])
        except ValueError:
            continue
        if i > index:
            index, stem = i, name
    return stem


def _need_h5py(path):
    from .keras_import import h5_module
    try:
        h5_module()
    except ImportError:
        raise RuntimeError("%s is a Keras HDF5 model file and neither h5py nor the HDF5 C library (libhdf5) is available: "
                           "cannot read it.  Install h5py, or convert the model to a torch checkpoint with "
                           "sejonggo_amd.model.save_model(); refusing to fall back to random weights" % path)


def _read_pt(path):
    return torch.load(path, map_location="cpu", weights_only=True)


def _build(name, size=None, n_blocks=None, channels=None):
    return PolicyValueNet(size or conf['SIZE'], conf['N_RESIDUAL_BLOCKS'] if n_blocks is None else n_blocks,
                          channels or conf.get('NET_CHANNELS', 256), name=name)


def _load_file(path):
    if path.endswith(".pt"):
        ck = _read_pt(path)
        net = _build(ck.get("name", INITIAL_MODEL_NAME), ck.get("size"), ck.get("n_blocks"), ck.get("channels"))
        net.load_state_dict(ck["state_dict"])
        return net.eval()
    _need_h5py(path)
    from .keras_import import keras_model_name, load_keras_h5
    net = _build(keras_model_name(path) or os.path.basename(path).split('.')[0])
    load_keras_h5(path, net)
    return net.eval()


def _name_of_file(path):
    if path.endswith(".pt"):
        return _read_pt(path).get("name", INITIAL_MODEL_NAME)
    _need_h5py(path)
    from .keras_import import keras_model_name
    return keras_model_name(path) or os.path.basename(path).split('.')[0]


def build_model(name):
    """model.py:55-95: a new network of the reference's topology (random init) under `name`."""
    return _build(name).eval()


def save_model(net, fname):
    """fname: file name under MODEL_DIR (any extension is replaced by .pt).  Written atomically."""
    os.makedirs(_dir(), exist_ok=True)
    path = os.path.join(_dir(), os.path.splitext(fname)[0] + ".pt")
    tmp = "%s.tmp%d" % (path, os.getpid())
    torch.save({"state_dict": {k: v.detach().cpu() for k, v in net.state_dict().items()}, "name": net.name,
                "size": net.size, "n_blocks": len(net.blocks), "channels": net.stem.out_channels}, tmp)
    os.replace(tmp, path)
    drop_sibling(path)
    return path


def create_initial_model(name=INITIAL_MODEL_NAME, self_play=False):
    """model.py:97-122: the named model if its file exists, else a new random-init network saved under its own name
    and as the best model.  (self_play=True -- the reference plays N_GAMES with the fresh model first -- is the
    job of selfplay_worker here and is not done inside the loader.)"""
    path = _find(name)
    if path is not None:
        return _load_file(path)
    torch.manual_seed(0)                 # every process that races to create it builds the same weights
    net = _build(name).eval()
    save_model(net, name)
    save_model(net, _best_stem())
    return net


def load_best_model():
    path = _find(_best_stem())
    if path is None:
        print("WARNING: found no best model under %r; initialising a NEW random model %r (model.py:153-155)"
              % (_dir(), INITIAL_MODEL_NAME), file=sys.stderr)
        return create_initial_model(name=INITIAL_MODEL_NAME, self_play=False)
    return _load_file(path)


def load_latest_model():
    stem = _latest_stem()
    if stem is None:
        raise FileNotFoundError("no model_<n> file under %r (model.py:125-144 load_latest_model)" % _dir())
    net = _load_file(_find(stem))
    if stem != net.name:
        print("WARNING: inconsistent model name: file %r holds model %r" % (stem, net.name), file=sys.stderr)
    return net


def load_model_by_name(name):
    path = _find(os.path.splitext(name)[0])
    if path is None:
        raise FileNotFoundError(os.path.join(_dir(), name))
    return _load_file(path)


def model_name(kind):
    """Name of the BEST / LATEST model from file metadata alone (no network is built, no GPU is touched).
    BEST without a file is the initial model's name, exactly what load_best_model would create."""
    if kind == "BEST":
        path = _find(_best_stem())
        return INITIAL_MODEL_NAME if path is None else _name_of_file(path)
    stem = _latest_stem()
    if stem is None:
        # load_best_model() creates model_1 when nothing exists; the latest model then is that same file
        return INITIAL_MODEL_NAME
    return _name_of_file(_find(stem))
