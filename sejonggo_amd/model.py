"""Model loading with the reference's names (model.py:125-163 load_best_model / load_latest_model).

The reference loads Keras .h5 files; `keras_import.load_keras_h5` assigns one to a PolicyValueNet where h5py exists
(the array mapping is tested, the container reader needs h5py).  Here a
model is a PolicyValueNet (net.py) restored from `<MODEL_DIR>/<name>.pt` (a torch state_dict saved by
`save_model`) when present, otherwise random-init with the reference's initial name `model_0`."""
import os

import torch

from .conf import conf
from .net import PolicyValueNet


def _path(fname):
    return os.path.join(conf['MODEL_DIR'], fname)


def _load(fname, default_name):
    size, nb = conf['SIZE'], conf['N_RESIDUAL_BLOCKS']
    path = _path(fname)
    net = PolicyValueNet(size, nb, 256, name=default_name)
    if os.path.isfile(path):
        ck = torch.load(path, map_location="cpu", weights_only=True)
        net.load_state_dict(ck["state_dict"])
        net.name = ck.get("name", default_name)
    else:
        torch.manual_seed(0)
        net = PolicyValueNet(size, nb, 256, name=default_name)
    return net.eval()


def save_model(net, fname):
    os.makedirs(conf['MODEL_DIR'], exist_ok=True)
    torch.save({"state_dict": net.state_dict(), "name": net.name}, _path(fname))


def load_best_model():
    return _load(os.path.splitext(conf['BEST_MODEL'])[0] + ".pt", "model_0")


def load_latest_model():
    return _load("latest_model.pt", "model_0")
