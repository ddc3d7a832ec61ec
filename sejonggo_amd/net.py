"""Resident PyTorch-ROCm policy/value network with the reference's topology and tensor contract.

Topology restated from model.py:55-95 (Keras): stem Conv3x3(17->256) with padding omitted => 'valid'
(the tower runs on (S-2)x(S-2)), BN, ReLU; N_RESIDUAL_BLOCKS x [Conv3x3 same, BN, ReLU, Conv3x3 same, BN,
+skip, ReLU] (model.py:37-46); policy head Conv1x1(->2), BN, ReLU, Flatten (channels-last order),
Dense(S*S+1, softmax) (:73-80); value head Conv1x1(->2), BN, ReLU, Flatten, Dense(256, relu),
Dense(1, tanh) (:83-90).  Contract (model.py:57,80,90,92): `.name`, `.predict_on_batch(X[n,S,S,17])`
-> [policy float32 [n,S*S+1], value float32 [n,1]].

Inference form: `fused()` folds every BatchNorm into the preceding convolution and switches to
channels_last + the requested dtype, so a forward pass is conv/GEMM (MFMA through MIOpen/hipBLASLt) +
ReLU only.  Weights are random-init unless loaded; there is no network access for checkpoints.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ResidualBlock(nn.Module):
    def __init__(self, ch):
        super().__init__()
        self.conv1 = nn.Conv2d(ch, ch, 3, padding=1)
        self.bn1 = nn.BatchNorm2d(ch, eps=1e-3, momentum=0.01)  # Keras BatchNormalization defaults
        self.conv2 = nn.Conv2d(ch, ch, 3, padding=1)
        self.bn2 = nn.BatchNorm2d(ch, eps=1e-3, momentum=0.01)

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + x)


class PolicyValueNet(nn.Module):
    def __init__(self, size=19, n_blocks=20, channels=256, name="model_0", stem_padding=0):
        super().__init__()
        self.size = size
        self.name = name
        self.A = size * size + 1
        t = size - 2 + 2 * stem_padding  # tower side length ('valid' stem in the reference)
        self.tower_side = t
        self.stem = nn.Conv2d(17, channels, 3, padding=stem_padding)
        self.stem_bn = nn.BatchNorm2d(channels, eps=1e-3, momentum=0.01)
        self.blocks = nn.ModuleList([ResidualBlock(channels) for _ in range(n_blocks)])
        self.p_conv = nn.Conv2d(channels, 2, 1)
        self.p_bn = nn.BatchNorm2d(2, eps=1e-3, momentum=0.01)
        self.p_fc = nn.Linear(2 * t * t, self.A)
        self.v_conv = nn.Conv2d(channels, 2, 1)
        self.v_bn = nn.BatchNorm2d(2, eps=1e-3, momentum=0.01)
        self.v_fc1 = nn.Linear(2 * t * t, 256)
        self.v_fc2 = nn.Linear(256, 1)
        self._fused = False

    # ---- training-form forward on NCHW input
    def forward(self, x):
        y = self.stem(x)
        y = F.relu(y if self._fused else self.stem_bn(y))
        for b in self.blocks:
            if self._fused:
                z = F.relu(b.conv1(y))
                y = F.relu(b.conv2(z) + y)
            else:
                y = b(y)
        p = self.p_conv(y)
        p = F.relu(p if self._fused else self.p_bn(p))
        p = p.permute(0, 2, 3, 1).reshape(p.shape[0], -1)   # Keras flattens channels-last
        p = torch.softmax(self.p_fc(p).float(), dim=1)
        v = self.v_conv(y)
        v = F.relu(v if self._fused else self.v_bn(v))
        v = v.permute(0, 2, 3, 1).reshape(v.shape[0], -1)
        v = torch.tanh(self.v_fc2(F.relu(self.v_fc1(v))).float())
        return p, v

    @torch.no_grad()
    def predict_on_batch(self, X):
        """X: [n,S,S,17] (NHWC like the reference), numpy or torch, any real dtype."""
        dev = next(self.parameters()).device
        dt = next(self.parameters()).dtype
        if not torch.is_tensor(X):
            import numpy as np
            X = torch.from_numpy(np.ascontiguousarray(X))
        x = X.to(device=dev, dtype=dt).permute(0, 3, 1, 2)   # NCHW view; stays channels_last in memory
        p, v = self.forward(x)
        return p, v

    @torch.no_grad()
    def fused(self, dtype=torch.float16):
        """Inference copy: BN folded into the convolutions, channels_last, `dtype`."""
        import copy
        m = copy.deepcopy(self).float().eval()

        def fold(conv, bn):
            s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            conv.weight.mul_(s.reshape(-1, 1, 1, 1))
            conv.bias.copy_((conv.bias - bn.running_mean) * s + bn.bias)

        fold(m.stem, m.stem_bn)
        for b in m.blocks:
            fold(b.conv1, b.bn1)
            fold(b.conv2, b.bn2)
        fold(m.p_conv, m.p_bn)
        fold(m.v_conv, m.v_bn)
        m._fused = True
        m = m.to(dtype).to(memory_format=torch.channels_last)
        return m

    def flops_per_eval(self):
        """Forward FLOPs (MAC = 2) of one position, as SURVEY.md §8d counts them."""
        t2 = self.tower_side ** 2
        ch = self.stem.out_channels
        f = 2 * t2 * 9 * 17 * ch
        f += len(self.blocks) * 2 * 2 * t2 * 9 * ch * ch
        f += 2 * 2 * t2 * ch * 2
        f += 2 * (2 * t2) * self.A + 2 * (2 * t2) * 256 + 2 * 256
        return f


def build_net(size, n_blocks, channels=256, name="model_0", seed=0, device="cuda", dtype=torch.float16):
    torch.manual_seed(seed)
    net = PolicyValueNet(size, n_blocks, channels, name=name)
    # non-trivial BN statistics so that folding is exercised
    for mod in net.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.5, 1.5)
    net.eval()
    return net.fused(dtype).to(device)


class FusedInferenceNet(object):
    """Inference-only form of PolicyValueNet for the engine's hot loop (same weights, same contract):

    * input NHWC fp16 with channels zero-padded 17 -> 32 (k_nn_pack layout 2), so the stem runs on an
      MFMA-friendly K instead of MIOpen's slow path for odd channel counts;
    * every 3x3 convolution of the reference's topology runs in a hand-written CDNA4 kernel of libsgo_hip.so with bias
      (+ skip) + ReLU fused: the tower (csrc/sgo_conv4w.hpp, sgo_conv8w.hpp) and the stem (csrc/sgo_stem.hpp); other channel counts
      (small test nets) fall back to the framework's convolution + ONE hand-written epilogue pass (k_bias_act);
    * both 1x1 head convolutions are one [n*t*t, C] x [C, 4] GEMM on the channels-last view.
    """
    in_channels = 32

    def __init__(self, net, dtype=torch.float16, device="cuda"):
        from . import _lib
        self._lib = _lib
        self.lib = _lib.require_gpu()
        assert dtype == torch.float16, "the fused epilogue kernel is fp16"
        # use MIOpen's Find path so the tuned solvers of the in-tree user database are honoured
        torch.backends.cudnn.benchmark = True
        f = net if net._fused else net.fused(torch.float32)
        f = f.float()
        self.name = net.name
        self.size, self.A, self.t = net.size, net.A, net.tower_side
        self.channels = f.stem.out_channels
        dev = torch.device(device)
        self.device = dev

        def cw(w):
            return w.to(dev, dtype).contiguous(memory_format=torch.channels_last)

        w = torch.zeros(self.channels, 32, 3, 3)
        w[:, :17] = f.stem.weight
        self.stem_w, self.stem_b = cw(w), f.stem.bias.to(dev, dtype).contiguous()
        self.stem_pad = f.stem.padding
        # the stem as sgo_stem_packed_dev wants it: [256][10 taps][16 stone planes] fp16 (tap = dy * 3 + dx, tap 9 zero) and the
        # colour plane folded into a per-position bias term: under the 'valid' stem every tap of plane 16 (= +-1 everywhere) is on
        # the board, so it contributes c * sum_taps w[k][16][tap] to every output pixel
        w10 = torch.zeros(self.channels, 10, 16)
        w10[:, :9, :] = f.stem.weight[:, :16].permute(0, 2, 3, 1).reshape(self.channels, 9, 16)
        self.stem_w10 = w10.to(dev, dtype).contiguous()
        self.stem_wcol = f.stem.weight[:, 16].reshape(self.channels, 9).sum(dim=1).to(dev, torch.float32).contiguous()
        self.blocks = [(cw(b.conv1.weight), b.conv1.bias.to(dev, dtype).contiguous(),
                        cw(b.conv2.weight), b.conv2.bias.to(dev, dtype).contiguous()) for b in f.blocks]
        hw = torch.cat([f.p_conv.weight.reshape(2, -1), f.v_conv.weight.reshape(2, -1)], 0)      # [4, C]
        hb = torch.cat([f.p_conv.bias, f.v_conv.bias], 0)
        self.head_w, self.head_b = hw.to(dev, dtype).contiguous(), hb.to(dev, dtype).contiguous()
        self.p_fc_w, self.p_fc_b = f.p_fc.weight.to(dev, dtype).contiguous(), f.p_fc.bias.to(dev, dtype).contiguous()
        self.v_fc1_w, self.v_fc1_b = f.v_fc1.weight.to(dev, dtype).contiguous(), f.v_fc1.bias.to(dev, dtype).contiguous()
        self.v_fc2_w, self.v_fc2_b = f.v_fc2.weight.to(dev, dtype).contiguous(), f.v_fc2.bias.to(dev, dtype).contiguous()
        self._flops = net.flops_per_eval()
        self.split_streams = False
        self.tail_split = False    # see _tail_split(): measured -1.4 % end to end on MI355X (bench.py --tail-split 1), off
        self.n_cu = None
        self.fused_conv = True     # conv + bias + skip + ReLU in one MFMA kernel (sgo_conv.hip); False: MIOpen + k_bias_act
        self.split_min = 1024
        self._side = None
        # tower route: False = OHWI weights through the LDS (k_conv4w, the default), True = fragment-order filter banks straight
        # into registers (k_conv4r, csrc/sgo_conv4r.hpp; same bits, same speed on MI355X: DESIGN.md 4a) -- use_packed_tower()
        self.packed_tower = False
        self._banks = {}
        self.conv_events = None    # list of (start, end, flops) HIP-event brackets around tower convolutions while set (bench.py)
        self.side_flops = 0.0      # tower FLOPs issued on the side stream while conv_events is set
        self.conv_event_stride = 1  # bracket every k-th tower launch only (small batches: the event calls would bound the host)
        self._conv_seq = 0

    def flops_per_eval(self):
        return self._flops

    def use_packed_tower(self, on=True):
        """Route the tower convolutions through sgo_conv3x3_tower_packed_dev (k_conv4r).  The filter banks are written here, once
        per layer (sgo_conv3x3_tower_prepack_dev), not on the evaluation path."""
        if on and self.channels == 256 and self.t <= 19 and not self._banks:
            nbytes = self.lib.sgo_conv3x3_tower_packed_bytes()
            st = torch.cuda.current_stream().cuda_stream
            for (w1, _, w2, _) in self.blocks:
                for w in (w1, w2):
                    bank = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
                    self._lib.check(self.lib.sgo_conv3x3_tower_prepack_dev(w.data_ptr(), bank.data_ptr(), st), "sgo_conv3x3_tower_prepack_dev")
                    self._banks[w.data_ptr()] = bank
        self.packed_tower = bool(on) and bool(self._banks)
        return self.packed_tower

    def _epilogue(self, y, bias, skip=None):
        L = self._lib
        L.check(self.lib.sgo_bias_act_dev(y.numel(), y.shape[1], y.data_ptr(), bias.data_ptr(),
                                          None if skip is None else skip.data_ptr(), y.data_ptr(),
                                          torch.cuda.current_stream().cuda_stream), "sgo_bias_act_dev")
        return y

    @staticmethod
    def has_kernel(c, k, pad, wd, skip):
        """Shapes libsgo_hip.so has a hand-written kernel for: the tower (256 -> 256, 'same', board width <= 19) and the
        stem (32 -> 256, 'valid')."""
        return (c == 256 and k == 256 and pad == 1 and wd <= 19) or (c == 32 and k == 256 and pad == 0 and skip is None)

    def _conv(self, x, w, b, pad, skip=None):
        """relu(conv3x3(x, w) + b (+ skip)) on NCHW views of channels-last tensors.  Other channel counts than the
        reference's (small test nets) go through the framework's convolution + the fused epilogue pass."""
        n, c, h, wd = x.shape
        k = w.shape[0]
        if not self.fused_conv or not self.has_kernel(c, k, pad, wd, skip):
            return self._epilogue(F.conv2d(x, w, None, padding=pad), b, skip=skip)
        y = torch.empty((n, k, h + 2 * pad - 2, wd + 2 * pad - 2), dtype=x.dtype, device=x.device,
                        memory_format=torch.channels_last)
        timed = self.conv_events is not None and c == k == 256 and pad == 1
        if timed and self._side is not None and torch.cuda.current_stream() == self._side:
            # the tail of a split batch runs CONCURRENTLY with the main part: its FLOPs are credited to the main stream's
            # launch time (bench.py), a bracket of its own would count the same wall time twice
            self.side_flops += 2.0 * n * y.shape[2] * y.shape[3] * 9 * c * k
            timed = False
        if timed:
            self._conv_seq += 1
            timed = self._conv_seq % self.conv_event_stride == 0
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        bank = self._banks.get(w.data_ptr()) if self.packed_tower and c == k == 256 and pad == 1 else None
        if bank is not None:
            self._lib.check(self.lib.sgo_conv3x3_tower_packed_dev(n, h, wd, x.data_ptr(), bank.data_ptr(), b.data_ptr(),
                                                              None if skip is None else skip.data_ptr(), y.data_ptr(),
                                                              torch.cuda.current_stream().cuda_stream), "sgo_conv3x3_tower_packed_dev")
        else:
            self._lib.check(self.lib.sgo_conv3x3_bias_act_dev(n, h, wd, c, k, pad, x.data_ptr(), w.data_ptr(), b.data_ptr(),
                                                              None if skip is None else skip.data_ptr(), y.data_ptr(),
                                                              torch.cuda.current_stream().cuda_stream), "sgo_conv3x3_bias_act_dev")
        if timed:
            e1.record()
            self.conv_events.append((e0, e1, 2.0 * n * y.shape[2] * y.shape[3] * 9 * c * k))
        return y

    def _tail_split(self, n):
        """The tower kernel runs one 256-pixel tile per workgroup and one workgroup per CU (its LDS fills the CU), so a launch
        takes ceil(tiles / CUs) rounds: 8 192 positions x 17 x 17 = 9 248 tiles = 36.1 rounds on 256 CUs cost 37, the last one
        with 32 tiles on 224 idle CUs -- 2.4 % of every launch.  Returns how many leading positions fill whole rounds when the
        rest is a small tail (it then runs on a side stream through all layers, its few workgroups slipping into the main
        part's round boundaries), or 0 when splitting does not pay."""
        if self.channels != 256 or self.t > 19:
            return 0
        if self.n_cu is None:
            self.n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count
        t2 = self.t * self.t
        tiles = -(-n * t2 // 256)
        rounds = tiles // self.n_cu
        rem = tiles - rounds * self.n_cu
        if rounds < 8 or rem == 0 or rem > self.n_cu // 4:
            return 0
        h = (rounds * self.n_cu * 256) // t2          # positions whose tiles fit the whole rounds
        return h if 0 < h < n else 0

    @property
    def packed_ok(self):
        """True when the stem can read packed position records (sgo_stem_packed_dev): the reference's shape, 'valid' stem."""
        pad0 = self.stem_pad[0] if isinstance(self.stem_pad, (tuple, list)) else self.stem_pad
        return self.fused_conv and self.channels == 256 and pad0 == 0 and self.t <= 19 and self.size in self._lib.SUPPORTED_SIZES

    @torch.no_grad()
    def predict_packed(self, records_ptr, index_ptr, n, k=0, k_dev_ptr=None):
        """The engine's route: rows = packed position records on the device (`records_ptr`: base of a record array,
        `index_ptr`: int32 device list of n record indices or None for records 0..n-1), evaluated under symmetry k (or the
        device int at `k_dev_ptr`).  No network-input tensor is materialised: the stem kernel expands the bit-planes in LDS.
        Returns (policy [n, A] float32, value [n, 1] float32) like predict_on_batch."""
        y = torch.empty((n, self.channels, self.t, self.t), dtype=torch.float16, device=self.device, memory_format=torch.channels_last)
        self._lib.check(self.lib.sgo_stem_packed_dev(self.size, n, records_ptr, index_ptr, int(k), k_dev_ptr, self.stem_w10.data_ptr(),
                                                     self.stem_b.data_ptr(), self.stem_wcol.data_ptr(), y.data_ptr(),
                                                     torch.cuda.current_stream().cuda_stream), "sgo_stem_packed_dev")
        return self._tower_and_heads(y)

    def _forward(self, X):
        x = X.permute(0, 3, 1, 2)                                 # NCHW view of channels-last memory
        pad0 = self.stem_pad[0] if isinstance(self.stem_pad, (tuple, list)) else self.stem_pad
        return self._tower_and_heads(self._conv(x, self.stem_w, self.stem_b, pad0))

    def _tower_and_heads(self, y):
        n = y.shape[0]
        for (w1, b1, w2, b2) in self.blocks:
            z = self._conv(y, w1, b1, 1)
            y = self._conv(z, w2, b2, 1, skip=y)
        t2 = self.t * self.t
        h = F.relu(F.linear(y.permute(0, 2, 3, 1).reshape(n * t2, self.channels), self.head_w, self.head_b))
        h = h.reshape(n, t2, 4)
        p = h[:, :, 0:2].reshape(n, 2 * t2)                       # Keras Flatten of [t, t, 2]
        v = h[:, :, 2:4].reshape(n, 2 * t2)
        p = torch.softmax(F.linear(p, self.p_fc_w, self.p_fc_b).float(), dim=1)
        v = torch.tanh(F.linear(F.relu(F.linear(v, self.v_fc1_w, self.v_fc1_b)), self.v_fc2_w, self.v_fc2_b).float())
        return p, v

    @torch.no_grad()
    def predict_on_batch(self, X):
        """X: [n,S,S,32] fp16 CUDA (channel-padded NHWC) or [n,S,S,17] (padded here).

        With `tail_split` (off by default: measured slower) a batch whose tower tiles do not fill whole rounds of the chip is split unevenly (see
        _tail_split).  With `split_streams` the batch is evaluated as two halves on two HIP streams, so that one half's HBM-bound
        epilogue passes can overlap the other half's MFMA-bound convolutions."""
        if not torch.is_tensor(X):
            import numpy as np
            X = torch.from_numpy(np.ascontiguousarray(X))
        X = X.to(self.device, torch.float16)
        if X.shape[-1] == 17:
            X = F.pad(X, (0, 15))
        n = X.shape[0]
        h = self._tail_split(n) if self.tail_split and not self.split_streams else 0
        if h == 0:
            if not self.split_streams or n < 2 * self.split_min or n % 2:
                return self._forward(X)
            h = n // 2
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        cur = torch.cuda.current_stream()
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            p1, v1 = self._forward(X[h:])
        p0, v0 = self._forward(X[:h])
        cur.wait_stream(self._side)
        for t in (p1, v1):
            t.record_stream(cur)
        return torch.cat([p0, p1]), torch.cat([v0, v1])


def build_fused_net(size, n_blocks, channels=256, name="model_0", seed=0, device="cuda"):
    torch.manual_seed(seed)
    net = PolicyValueNet(size, n_blocks, channels, name=name)
    for mod in net.modules():
        if isinstance(mod, nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.1)
            mod.running_var.uniform_(0.5, 1.5)
    net.eval()
    return FusedInferenceNet(net, torch.float16, device), net
