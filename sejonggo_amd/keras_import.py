"""Weights of the reference's Keras model <-> net.PolicyValueNet.

The reference builds its network in model.py:55-95 and saves it with Keras (`model.save`, model.py:147-157): an HDF5 file
whose `model_weights` group holds one sub-group per layer with the layer's arrays in Keras layouts.  This module does the
layout and ordering work on plain numpy arrays:

  * Conv2D kernel [kh][kw][in][out] -> torch [out][in][kh][kw]; Dense kernel [in][out] -> torch [out][in];
    BatchNormalization (gamma, beta, moving_mean, moving_variance) -> (weight, bias, running_mean, running_var),
    epsilon 1e-3 as in Keras' default (net.PolicyValueNet is built with it);
  * layers are matched by Keras' auto-generated names: `conv2d_<n>` / `batch_normalization_<n>` sorted by <n> are in
    creation order (stem, [block conv1, conv2] x N, policy 1x1, value 1x1 -- model.py:57-84), the heads' Dense layers are
    `policy_out`, `value_out` (explicit names, model.py:80, 90) and the one remaining `dense_<n>` (value hidden, :89);
  * Keras flattens the heads' [t][t][2] activations channels-last, which is the order net.PolicyValueNet.forward uses, so the
    Dense kernels carry over without a permutation.

The HDF5 container is read through h5py where that is installed, otherwise through h5lite (this package's ctypes binding to
libhdf5, the C library h5py itself wraps; present in this image under /opt/conda/lib).  No Keras file ships with the reference
(`*.h5` is git-ignored there) and Keras is not installable here, so the tests write a file in Keras' documented layout
(`save_keras_h5`) with libhdf5 and read it back: the container format is libhdf5's own, the layer / weight naming is pinned
only by Keras' source (keras/engine/saving.py), not by a Keras-written file -- PARITY of that naming is UNPINNED."""
import re

import numpy as np
import torch


def _num(name):
    m = re.search(r"_(\d+)$", name)
    return int(m.group(1)) if m else 0


def _modules(net):
    """(conv, bn) pairs and dense layers of a PolicyValueNet in the reference's creation order."""
    convs = [(net.stem, net.stem_bn)]
    for b in net.blocks:
        convs += [(b.conv1, b.bn1), (b.conv2, b.bn2)]
    convs += [(net.p_conv, net.p_bn), (net.v_conv, net.v_bn)]
    return convs, {"policy_out": net.p_fc, "value_hidden": net.v_fc1, "value_out": net.v_fc2}


def export_keras_layers(net, first_index=1):
    """[(keras_layer_name, [arrays in Keras layouts])] of a PolicyValueNet, in creation order."""
    convs, dense = _modules(net)
    out = []
    for i, (c, bn) in enumerate(convs):
        out.append(("conv2d_%d" % (first_index + i), [c.weight.detach().cpu().numpy().transpose(2, 3, 1, 0).copy(),
                                                      c.bias.detach().cpu().numpy().copy()]))
        out.append(("batch_normalization_%d" % (first_index + i),
                    [bn.weight.detach().cpu().numpy().copy(), bn.bias.detach().cpu().numpy().copy(),
                     bn.running_mean.detach().cpu().numpy().copy(), bn.running_var.detach().cpu().numpy().copy()]))
    for kname, key in (("policy_out", "policy_out"), ("dense_%d" % first_index, "value_hidden"), ("value_out", "value_out")):
        d = dense[key]
        out.append((kname, [d.weight.detach().cpu().numpy().T.copy(), d.bias.detach().cpu().numpy().copy()]))
    return out


@torch.no_grad()
def assign_keras_layers(net, layers):
    """Copy Keras-layout arrays into `net` (a PolicyValueNet of matching size / depth).  `layers`: iterable of
    (layer_name, [arrays]) in any order; layers without weights (Input, Activation, Add, Reshape) may be present."""
    layers = [(n, [np.asarray(a) for a in w]) for n, w in layers if len(w)]
    convs = sorted([l for l in layers if l[0].startswith("conv2d")], key=lambda l: _num(l[0]))
    bns = sorted([l for l in layers if l[0].startswith("batch_normalization")], key=lambda l: _num(l[0]))
    named = {n: w for n, w in layers}
    hidden = [l for l in layers if l[0].startswith("dense")]
    want, dense = _modules(net)
    if len(convs) != len(want) or len(bns) != len(want):
        raise ValueError("expected %d Conv2D / BatchNormalization layers (1 stem + 2 x %d blocks + 2 heads), got %d / %d"
                         % (len(want), len(net.blocks), len(convs), len(bns)))
    if "policy_out" not in named or "value_out" not in named or len(hidden) != 1:
        raise ValueError("expected Dense layers policy_out, value_out and one dense_<n> (value hidden)")

    def put(param, arr):
        if tuple(param.shape) != tuple(arr.shape):
            raise ValueError("shape mismatch: have %s, file has %s" % (tuple(param.shape), tuple(arr.shape)))
        param.copy_(torch.from_numpy(np.ascontiguousarray(arr)).to(param.dtype))

    for (c, bn), (cn, cw), (bname, bw) in zip(want, convs, bns):
        if len(cw) != 2 or len(bw) != 4:
            raise ValueError("%s / %s: expected [kernel, bias] and [gamma, beta, moving_mean, moving_variance]" % (cn, bname))
        put(c.weight, cw[0].transpose(3, 2, 0, 1))
        put(c.bias, cw[1])
        put(bn.weight, bw[0]); put(bn.bias, bw[1]); put(bn.running_mean, bw[2]); put(bn.running_var, bw[3])
    for key, w in (("policy_out", named["policy_out"]), ("value_hidden", hidden[0][1]), ("value_out", named["value_out"])):
        put(dense[key].weight, w[0].T)
        put(dense[key].bias, w[1])
    return net


def h5_module():
    """h5py when it is installed, otherwise this package's ctypes binding to the same C library (h5lite); ImportError when
    neither can be had."""
    try:
        import h5py
        return h5py
    except Exception:
        from . import h5lite
        if not h5lite.available():
            raise ImportError("reading a Keras .h5 file needs h5py or the HDF5 C library (libhdf5); neither was found")
        return h5lite


def keras_model_name(path):
    """`model.name` of a Keras model file: the `name` of its `model_config` attribute (model.py:92 builds the model with
    an explicit name and train.py renames it before saving).  None for a weights-only file."""
    import json
    h5py = h5_module()
    with h5py.File(path, "r") as f:
        cfg = f.attrs.get("model_config")
    if cfg is None:
        return None
    if isinstance(cfg, bytes):
        cfg = cfg.decode()
    return json.loads(cfg).get("config", {}).get("name")


def load_keras_h5(path, net):
    """Read a Keras model / weights file (keras/engine/saving.py layout: group `model_weights`, attribute `layer_names`,
    one group per layer with attribute `weight_names` and one dataset per weight) and assign it to `net`."""
    h5py = h5_module()
    with h5py.File(path, "r") as f:
        g = f["model_weights"] if "model_weights" in f else f
        layers = []
        for lname in [n.decode() if isinstance(n, bytes) else n for n in g.attrs["layer_names"]]:
            lg = g[lname]
            names = [n.decode() if isinstance(n, bytes) else n for n in lg.attrs["weight_names"]]
            layers.append((lname, [np.asarray(lg[n]) for n in names]))
    return assign_keras_layers(net, layers)


def save_keras_h5(path, net, keras_version=b"2.2.2", backend=b"tensorflow"):
    """Write `net` the way keras.Model.save lays a file out (saving.py: save_model + save_weights_to_hdf5_group): root
    attributes keras_version / backend / model_config (JSON with the model's name), group `model_weights` with attribute
    layer_names, one group per layer with attribute weight_names and datasets `<layer>/<weight>:0`.  (No optimizer state.)"""
    import json
    h5py = h5_module()
    layers = export_keras_layers(net)
    weight_suffix = {2: ["kernel:0", "bias:0"], 4: ["gamma:0", "beta:0", "moving_mean:0", "moving_variance:0"]}
    with h5py.File(path, "w") as f:
        f.attrs["keras_version"] = keras_version
        f.attrs["backend"] = backend
        f.attrs["model_config"] = json.dumps({"class_name": "Model", "config": {"name": net.name}}).encode("utf8")
        g = f.create_group("model_weights")
        g.attrs["layer_names"] = np.array([n.encode() for n, _ in layers], dtype="S")
        g.attrs["backend"] = backend
        g.attrs["keras_version"] = keras_version
        for lname, arrays in layers:
            lg = g.create_group(lname)
            names = ["%s/%s" % (lname, sfx) for sfx in weight_suffix[len(arrays)]]
            lg.attrs["weight_names"] = np.array([n.encode() for n in names], dtype="S")
            for n, a in zip(names, arrays):
                lg.create_dataset(n, data=np.asarray(a, dtype=np.float32))
    return path
