"""TreeNode view over a host dict node (sejonggo_amd.play / SelfPlayEngine.tree_dict).  Attribute names follow
mcts1/tree_node.py:6-105 (pos, v, w, pv, pw, av, aw, children, expand, rave_urgency, winrate, best_move).
Parity UNPINNED: the reference class is never executed."""
from math import sqrt


class TreeNode(object):
    def __init__(self, node, pos=None):
        self._node = node
        self.pos = pos                 # caller-supplied position object, if any
        self.pv, self.pw = 0, 0        # no prior pseudo-visits in this build (priors live in node['p'])
        self.av, self.aw = 0, 0        # no AMAF statistics

    @property
    def v(self):
        return self._node['count']

    @property
    def w(self):
        # mcts1 counts wins in [0, v]; this build keeps a value sum in [-v, v]
        return (self._node['value'] + self._node['count']) / 2.0

    @property
    def children(self):
        sub = self._node['subtree']
        return [TreeNode(c) for _, c in sorted(sub.items())] if sub else None

    def expand(self):
        raise NotImplementedError("read-only view: expansion happens in the GPU engine")

    def winrate(self):
        return float(self.w) / self.v if self.v > 0 else float('nan')

    def rave_urgency(self):
        # without AMAF statistics the RAVE blend degenerates to the plain win rate with the prior as tie-break
        return (self.winrate() if self.v > 0 else 0.5) + float(self._node['p']) / (1.0 + sqrt(1 + self.v))

    def best_move(self):
        ch = self.children
        return max(ch, key=lambda n: n.v) if ch else None

    @property
    def move(self):
        return self._node['index']
