"""sejonggo_amd -- MI355X-native self-play hot path of sejonggo (Go rules + virtual-loss PUCT + leaf
batching) behind the reference's own Python surface.  The compute lives in libsgo_hip.so
(hand-written HIP for gfx950, C ABI in include/sgo.h); this package is the host-side mirror of the
reference modules play / symmetry / tree_util / nomodel_self_play / predicting_queue_worker /
selfplay_worker for that path.  There is no CPU fallback: without the HIP library and a GPU the
compute entry points raise."""
__version__ = "0.1.0"

import os as _os

# MIOpen picks much faster solvers for the tower convolution once it has been tuned for the exact shapes
# (3.3 ms vs 5.1 ms per 8192x256x17x17 fp16 conv on MI355X).  The tuned entries live in-tree as a MIOpen *user
# database* (plain text, written by tools/tune_miopen.py); point MIOpen at it unless the user chose another one.
# Must happen before the first convolution initialises MIOpen, hence at package import.
_db = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "miopen_db")
if _os.path.isdir(_db):
    _os.environ.setdefault("MIOPEN_USER_DB_PATH", _db)
# MIOpen's reference solver (naive_conv, ~1.5 s per call at this batch size) is benchmarked by every Find on a box
# whose kernel cache is cold; it can never win, so keep it out of the candidate list.
_os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "0")
