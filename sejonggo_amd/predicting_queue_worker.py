"""Drop-in for the reference's predicting_queue_worker.py API (init/destroy workers, put_predict_request,
put_name_request; predicting_queue_worker.py:12-20,109-124).

The reference forks one inference process per GPU and funnels single boards through a shared Queue +
one Pipe per request, batching at most PREDICTING_BATCH_SIZE of them (predicting_queue_worker.py:40-102).
On MI355X the leaf batching happens on the device (engine.SelfPlayEngine: every step evaluates the
leaves of all resident games in one forward pass), so these functions only keep the *call surface*:
the nets live in this process, one per GPU, and `put_predict_request` is a synchronous single-board
evaluation for callers that still want one (GTP front-ends, debugging).  Indicators are the reference's:
"BEST", "LATEST", "BEST_SYM", "LATEST_SYM" (+ "*_NAME" through put_name_request).  The reference's
LATEST_SYM-uses-best-model slip (predicting_queue_worker.py:92) is NOT reproduced."""
import numpy as np

from . import _lib
from .conf import conf

_models = {}     # gpu_id -> {"BEST": net, "LATEST": net}
_default_gpu = None
_factory = None  # optional callable(kind) -> net, for tests / stub nets


def set_model_factory(fn):
    """fn(kind) -> model object for kind in ("BEST", "LATEST"); replaces model.load_*_model."""
    global _factory
    _factory = fn


def _load(kind, gpu_id):
    import torch
    if _factory is not None:
        return _factory(kind)
    from .model import load_best_model, load_latest_model
    net = load_best_model() if kind == "BEST" else load_latest_model()
    dt = torch.float16 if conf.get('NET_DTYPE', 'fp16') == 'fp16' else torch.float32
    fused = net.fused(dt).to(torch.device("cuda", gpu_id))
    fused.name = net.name
    return fused


def init_predicting_workers(GPUs):
    global _default_gpu
    _lib.require_gpu()
    for gpu_id in GPUs:
        if gpu_id not in _models:
            _models[gpu_id] = {"BEST": _load("BEST", gpu_id), "LATEST": _load("LATEST", gpu_id)}
        if _default_gpu is None:
            _default_gpu = gpu_id


def destroy_predicting_workers(GPUs):
    global _default_gpu
    for gpu_id in GPUs:
        _models.pop(gpu_id, None)
    if _default_gpu not in _models:
        _default_gpu = next(iter(_models), None)


def get_model(model_indicator, gpu_id=None):
    g = _default_gpu if gpu_id is None else gpu_id
    if g is None or g not in _models:
        raise _lib.SgoError("init_predicting_workers(GPUs) has not been called")
    kind = "BEST" if model_indicator.startswith("BEST") else "LATEST"
    return _models[g][kind]


def put_name_request(model_indicator):
    return get_model(model_indicator).name


def put_predict_request(model_indicator, board, response_now=False):
    """-> (policy float32[S*S+1], value float32) for ONE board [1,S,S,17] (predicting_queue_worker.py:120-124)."""
    from .symmetry import random_symmetry_predict
    net = get_model(model_indicator)
    if model_indicator.endswith("_SYM"):
        p, v = random_symmetry_predict(_NumpyNet(net), np.array(board))
    else:
        p, v = _NumpyNet(net).predict_on_batch(np.asarray(board))
    return p[0], v[0][0]


def predict_batch(model_indicator, boards):
    """Batched form of put_predict_request for boards [n,S,S,17] -> (policy [n,A] float32, value [n,1] float32)."""
    from .symmetry import random_symmetry_predict
    net = get_model(model_indicator)
    if model_indicator.endswith("_SYM"):
        return random_symmetry_predict(_NumpyNet(net), np.array(boards))
    return _NumpyNet(net).predict_on_batch(np.asarray(boards))


class _NumpyNet(object):
    def __init__(self, net):
        self.net = net
        self.name = getattr(net, "name", "model")

    def predict_on_batch(self, X):
        import torch
        if getattr(self.net, "numpy_native", False) or not hasattr(self.net, "parameters") and not hasattr(self.net, "device"):
            p, v = self.net.predict_on_batch(np.ascontiguousarray(X))       # stub nets compute on numpy directly
        else:
            p, v = self.net.predict_on_batch(torch.from_numpy(np.ascontiguousarray(X)).cuda() if not torch.is_tensor(X) else X)
        if torch.is_tensor(p):
            p = p.float().cpu().numpy()
            v = v.float().cpu().numpy()
        return p.astype(np.float32), v.astype(np.float32).reshape(-1, 1)
