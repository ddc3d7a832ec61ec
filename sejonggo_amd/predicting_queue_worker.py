"""Drop-in for the reference's predicting_queue_worker.py API (init/destroy workers, put_predict_request,
put_name_request; predicting_queue_worker.py:12-20,109-124).

The reference forks one inference process per GPU and funnels single boards through a shared Queue +
one Pipe per request, batching at most PREDICTING_BATCH_SIZE of them (predicting_queue_worker.py:40-102).
On MI355X the leaf batching happens on the device (engine.SelfPlayEngine: every step evaluates the
leaves of all resident games in one forward pass), so these functions only keep the *call surface*.

Process model.  main_selfplay.main() calls init_predicting_workers(GPUs) and put_name_request("BEST") in the
PARENT and then forks the self-play workers (main_selfplay.py:16-28).  A forked child cannot use a GPU runtime its
parent has initialised, so nothing here touches the GPU until a network is actually needed:

* init_predicting_workers(GPUs) only registers the GPU ids;
* put_name_request(indicator) answers from file metadata (model.model_name) unless this process already holds the net;
* get_model / put_predict_request load the net onto the GPU on first use, in the process that uses it (a self-play
  worker, a GTP front-end, a test), and keep it resident there.  A cache inherited over a fork is discarded.

The resident form of a PolicyValueNet is net.FusedInferenceNet: BatchNorm folded, NHWC fp16, every 3x3 convolution
through the hand-written tower kernel of libsgo_hip.so (conf['NET_DTYPE'] = 'fp32' keeps the plain torch module).

Indicators are the reference's: "BEST", "LATEST", "BEST_SYM", "LATEST_SYM" (+ "*_NAME").  The reference evaluates
LATEST_SYM requests with the BEST model (predicting_queue_worker.py:92); conf['COMPAT_LATEST_SYM'] (default on, like
COMPAT_Z) reproduces that, off routes LATEST_SYM to the latest model."""
import os

import numpy as np

from . import _lib
from .conf import conf

_gpus = []        # ids registered by init_predicting_workers
_models = {}      # (gpu_id, "BEST" | "LATEST") -> resident net of THIS process
_models_pid = None
_factory = None   # optional callable(kind) -> net, for tests / stub nets


def set_model_factory(fn):
    """fn(kind) -> model object for kind in ("BEST", "LATEST"); replaces model.load_*_model.  The object may carry
    `cpu_only_name` semantics simply by having `.name`; it is built in the process that evaluates with it."""
    global _factory
    _factory = fn
    _models.clear()


def _own_cache():
    """Models loaded by another process (our parent before a fork) are not usable here."""
    global _models_pid
    if _models_pid != os.getpid():
        _models.clear()
        _models_pid = os.getpid()


def resident_form(net, gpu_id):
    """The inference form of a loaded PolicyValueNet on GPU `gpu_id` (same weights, same contract, `.name` kept)."""
    import torch
    from .net import FusedInferenceNet, PolicyValueNet
    if not isinstance(net, PolicyValueNet):
        return net                                   # stub nets and caller-supplied objects are used as they are
    dev = torch.device("cuda", gpu_id)
    if conf.get('NET_DTYPE', 'fp16') == 'fp16' and net.stem.out_channels % 8 == 0:
        with torch.cuda.device(dev):
            fused = FusedInferenceNet(net, torch.float16, dev)
        fused.name = net.name
        return fused
    m = net.fused(torch.float32).to(dev)
    m.name = net.name
    return m


def _load(kind, gpu_id):
    if _factory is not None:
        return resident_form(_factory(kind), gpu_id)
    from .model import load_best_model, load_latest_model
    net = load_best_model() if kind == "BEST" else load_latest_model()
    return resident_form(net, gpu_id)


def init_predicting_workers(GPUs):
    """Registers the GPUs this process (or its forked workers) will use.  No GPU call happens here."""
    for gpu_id in GPUs:
        if gpu_id not in _gpus:
            _gpus.append(gpu_id)


def destroy_predicting_workers(GPUs):
    _own_cache()
    for gpu_id in GPUs:
        if gpu_id in _gpus:
            _gpus.remove(gpu_id)
        for kind in ("BEST", "LATEST"):
            _models.pop((gpu_id, kind), None)


def _kind(model_indicator, for_name=False):
    if model_indicator.startswith("BEST"):
        return "BEST"
    if not model_indicator.startswith("LATEST"):
        raise KeyError(model_indicator)              # the reference's worker dies on an unknown indicator, too
    if model_indicator == "LATEST_SYM" and not for_name and conf.get('COMPAT_LATEST_SYM', True):
        return "BEST"                                # predicting_queue_worker.py:92
    return "LATEST"


def get_model(model_indicator, gpu_id=None):
    """The resident net behind an indicator, loaded onto the GPU on first use in this process."""
    _own_cache()
    if gpu_id is None:
        if not _gpus:
            raise _lib.SgoError("init_predicting_workers(GPUs) has not been called")
        gpu_id = _gpus[0]
    kind = _kind(model_indicator)
    key = (gpu_id, kind)
    if key not in _models:
        _lib.require_gpu()
        _models[key] = _load(kind, gpu_id)
    return _models[key]


def put_name_request(model_indicator):
    """Model name behind BEST* / LATEST* (predicting_queue_worker.py:109-117).  Never initialises a GPU."""
    _own_cache()
    kind = _kind(model_indicator, for_name=True)
    for (g, k), net in _models.items():
        if k == kind:
            return net.name
    if _factory is not None:
        return _factory(kind).name
    from .model import model_name
    return model_name(kind)


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


class PredictingQueueWorker(object):
    """predicting_queue_worker.py:23-106 is a Process that owns one GPU and serves the request queue.  Here nothing needs
    serving -- leaf batching is on the device and single requests are answered in the caller's process -- so the class keeps
    the constructor and the start / join / load_model surface for code that instantiates it, and does no work of its own."""

    def __init__(self, gpu_id):
        self.gpu_id = gpu_id
        self.best_model = None
        self.latest_model = None

    def load_model(self):
        init_predicting_workers([self.gpu_id])
        self.best_model = get_model("BEST", self.gpu_id)
        self.latest_model = get_model("LATEST", self.gpu_id)

    def start(self):
        init_predicting_workers([self.gpu_id])

    def join(self, timeout=None):
        return None

    def run(self):
        return None
