"""sejonggo_amd -- MI355X-native self-play hot path of sejonggo (Go rules + virtual-loss PUCT + leaf
batching) behind the reference's own Python surface.  The compute lives in libsgo_hip.so
(hand-written HIP for gfx950, C ABI in include/sgo.h); this package is the host-side mirror of the
reference modules play / symmetry / tree_util / nomodel_self_play / predicting_queue_worker /
selfplay_worker for that path.  There is no CPU fallback: without the HIP library and a GPU the
compute entry points raise."""
__version__ = "0.1.0"
