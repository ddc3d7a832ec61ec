"""Read-only adapter for the attribute surface of the reference's mcts1 package (mcts1/tree_node.py:6-105).
The reference's mcts1 is dead code (no importers; broken under Python 3), so there is nothing to
be bit-compatible with: semantic parity is UNPINNED.  The adapter only lets code written against TreeNode's
attributes walk a tree of this build."""
from .tree_node import TreeNode  # noqa: F401
