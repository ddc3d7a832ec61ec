"""CPU-only tests of the host dict-tree functions (sejonggo_amd.play selectors, tree_util, back_propagation).
They read like the reference's test/tree_util_tests.py and are pinned by tests/golden/puct.npz (outputs of the
reference's own selectors)."""
import numpy as np

from tests.helpers import load


def _subtree(P, N, Q, V, EX, f64):
    sub = {}
    for a in range(len(P)):
        if not EX[a]:
            continue
        sub[a] = {'index': a, 'count': int(N[a]), 'value': 0, 'mean_value': np.float32(Q[a]) if N[a] else 0,
                  'p': np.float64(P[a]) if f64 else np.float32(P[a]), 'subtree': {}, 'parent': None, 'virtual_loss': int(V[a])}
    return sub


def test_selectors_match_reference_outputs():
    from sejonggo_amd.play import top_one_with_virtual_loss, top_one_action, top_n_actions
    z = load("puct.npz")
    for c in range(len(z["F64"])):
        sub = _subtree(z["P"][c], z["N"][c], z["Q"][c], z["V"][c], z["EX"][c], int(z["F64"][c]))
        r = top_one_with_virtual_loss({'subtree': sub})
        assert (r['action'] if r else -1) == z["out_vl"][c], c
        assert top_one_action(sub)['action'] == z["out_one"][c], c
        assert [d['action'] for d in top_n_actions(sub, 8)] == [a for a in z["out_top"][c] if a >= 0], c


def _leaf(index, p, value=0):
    return {'index': index, 'count': 0, 'p': p, 'value': value, 'mean_value': 0, 'virtual_loss': 0, 'subtree': {}}


def _tree():
    tree = {'index': -1, 'count': 0, 'mean_value': 0, 'virtual_loss': 0, 'value': 0, 'parent': None, 'subtree': {
        0: dict(_leaf(2, 1, 1), subtree={3: _leaf(4, 1, 1), 4: _leaf(5, 0)}),
        1: _leaf(3, 0)}}
    for c in tree['subtree'].values():
        c['parent'] = tree
    for c in tree['subtree'][0]['subtree'].values():
        c['parent'] = tree['subtree'][0]
    return tree


def test_find_best_leaf_order_and_backoff():
    """tree_util_tests.py:69-84: leaves come out as [0,3], [0,4], then the busy inner node is flagged and [1]."""
    from sejonggo_amd.play import tree_depth
    from sejonggo_amd.tree_util import find_best_leaf_virtual_loss
    tree = _tree()
    assert tree_depth(tree) == 3
    node, moves = find_best_leaf_virtual_loss(tree)
    assert node['index'] == 4 and node['virtual_loss'] > 0 and moves == [0, 3]
    node, moves = find_best_leaf_virtual_loss(tree)
    assert node['index'] == 5 and moves == [0, 4]
    node, moves = find_best_leaf_virtual_loss(tree)
    assert node['index'] == 3 and moves == [1] and tree['subtree'][0]['virtual_loss'] > 0
    assert find_best_leaf_virtual_loss(tree) == (None, None)       # tree_util_tests.py:86-122


def test_get_node_by_moves():
    from sejonggo_amd.tree_util import get_node_by_moves
    import pytest
    tree = _tree()
    assert get_node_by_moves(tree, [0])['index'] == 2
    assert get_node_by_moves(tree, [0, 4])['index'] == 5
    assert get_node_by_moves(tree, [1])['index'] == 3
    with pytest.raises(Exception):
        get_node_by_moves(tree, [0, 7])


def test_back_propagation_grafts_and_counts():
    """tree_util_tests.py:139-196: the evaluated leaf replaces its placeholder; every ancestor gets +1 / +value."""
    from sejonggo_amd.nomodel_self_play import back_propagation
    from sejonggo_amd.tree_util import find_best_leaf_virtual_loss
    tree = _tree()
    leaf, moves = find_best_leaf_virtual_loss(tree)
    new_leaf = dict(leaf, parent=None, count=1, value=np.float32(0.5), mean_value=np.float32(0.5),
                    subtree={7: dict(_leaf(7, 1), parent=None)})
    back_propagation((new_leaf, moves), tree)
    assert tree['subtree'][0]['subtree'][3] is new_leaf and new_leaf['parent'] is tree['subtree'][0]
    assert new_leaf['virtual_loss'] == 0 and tree['subtree'][0]['virtual_loss'] == 0
    assert tree['subtree'][0]['count'] == 1 and tree['count'] == 1
    assert tree['subtree'][0]['value'] == 1.5 and tree['value'] == 0.5 and tree['mean_value'] == 0.5
