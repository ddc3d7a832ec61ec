"""Drop-in for the reference's tree_util.py on host dict trees (see the note in play.py): the virtual-loss leaf
finder (tree_util.py:4-24) and path lookup (:27-32).  The GPU engine implements the same walk in k_search."""
from .play import top_one_with_virtual_loss


def find_best_leaf_virtual_loss(node):
    """Descend by PUCT over non-busy children; flag the leaf busy (virtual_loss = 2).  A node whose children are
    all busy is flagged itself and the walk backs off to its parent; at the root that means (None, None)."""
    moves = []
    while node['subtree'] != {}:
        pick = top_one_with_virtual_loss(node)
        if pick == {}:
            if node['parent'] is None:
                return None, None
            node['virtual_loss'] = 2
            node = node['parent']
            del moves[-1:]
            continue
        node = pick['node']
        moves.append(pick['action'])
    node['virtual_loss'] = 2
    return node, moves


def get_node_by_moves(node, moves):
    for m in moves:
        if node['subtree'].get(m) is None:
            raise Exception("ERROR: Unable to get node: Invalid moves array")
        node = node['subtree'][m]
    return node
