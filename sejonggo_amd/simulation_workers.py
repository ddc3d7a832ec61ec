"""Drop-in for the reference's simulation_workers.py call surface (simulation_workers.py:19-54): init_simulation_workers,
init_simulation_workers_by_gpuid, destroy_simulation_workers, basic_tasks2, the per-process result queues and
`process_pool`.

The reference keeps a multiprocessing.Pool(ENERGY) of CPU workers per game process; each runs one leaf task
(`basic_tasks2`: replay the moves from the root board, ask the inference process for (policy, value), build the
leaf's subtree, set its statistics) and posts the result on simulation_result_queue[process_id].  On MI355X the
leaf tasks of ALL resident games are one batched launch inside the device engine (k_board_advance_rows + the net +
expansion in k_search), so there is nothing to pool: the init / destroy functions are pool-free, and `process_pool`
is an in-process object with the Pool methods the reference's callers use (apply_async, map, close, join) that runs
the task at once in the caller -- its rules and network calls go to the GPU through play.* / put_predict_request.
Results therefore arrive in launch order, the interleaving the goldens were recorded with."""
from collections import deque

from .conf import conf


class _ResultQueue(object):
    """SimpleQueue look-alike (put / get / empty) for results produced in this process."""

    def __init__(self):
        self._q = deque()

    def put(self, item):
        self._q.append(item)

    def get(self):
        if not self._q:
            raise RuntimeError("simulation_result_queue.get() on an empty queue: no leaf task is in flight")
        return self._q.popleft()

    def empty(self):
        return not self._q


class _Done(object):
    """AsyncResult look-alike of a task that already ran."""

    def __init__(self, value):
        self._value = value

    def get(self, timeout=None):
        return self._value

    def wait(self, timeout=None):
        return None

    def ready(self):
        return True

    def successful(self):
        return True


class _InlinePool(object):
    def apply_async(self, func, args=(), kwds=None, callback=None, error_callback=None):
        value = func(*args, **(kwds or {}))
        if callback is not None:
            callback(value)
        return _Done(value)

    def apply(self, func, args=(), kwds=None):
        return func(*args, **(kwds or {}))

    def map(self, func, iterable, chunksize=None):
        return [func(item) for item in iterable]

    def close(self):
        pass

    def join(self):
        pass

    def terminate(self):
        pass


class _QueueTable(dict):
    """simulation_result_queue[process_id]; the reference pre-creates range(N_GAME_PROCESS), any id works here."""

    def __missing__(self, key):
        q = self[key] = _ResultQueue()
        return q


MCTS_SIMULATIONS_PROCESSES = conf['ENERGY']
simulation_result_queue = _QueueTable()
process_pool = None
lock = None


def init_simulation_workers():
    """simulation_workers.py:19-23."""
    global process_pool, lock
    import threading
    lock = threading.Lock()
    process_pool = _InlinePool()


def init_simulation_workers_by_gpuid(GPU_ID):
    """simulation_workers.py:26-28."""
    global process_pool
    process_pool = _InlinePool()


def init_pool_param(l):
    """simulation_workers.py:31-33."""
    global lock
    lock = l


def destroy_simulation_workers():
    """simulation_workers.py:36-39."""
    global process_pool
    if process_pool is not None:
        process_pool.close()
        process_pool.join()
        process_pool = None


def _finish_leaf(node, board, policy, value, original_player):
    from .play import new_subtree
    node['subtree'] = new_subtree(policy, board, node)
    v = value if board[0, 0, 0, -1] == original_player else -value
    node['count'] += 1
    node['value'] += v
    node['mean_value'] = node['value'] / float(node['count'])
    return v


def basic_tasks2(node, board, moves, model_indicator, original_player, process_id):
    """simulation_workers.py:42-54: one leaf task.  `board` is the ROOT board (mutated, like the reference)."""
    from .play import index2coord, make_play
    from .predicting_queue_worker import put_predict_request
    size = board.shape[-2]
    for m in moves:
        x, y = index2coord(m, size)
        board, _ = make_play(x, y, board)
    policy, value = put_predict_request(model_indicator, board)
    _finish_leaf(node, board, policy, value, original_player)
    simulation_result_queue[process_id].put((node, moves))


def board_worker(input_tuple):
    """simulation_workers.py:81-104: the leaf board below one top_n pick; dic['node'] is moved to the leaf."""
    from .play import index2coord, make_play, top_one_action
    dic, board = input_tuple
    size = board.shape[-2]
    x, y = index2coord(dic['action'], size)
    make_play(x, y, board)
    node = dic['node']
    while node['subtree'] != {}:
        pick = top_one_action(node['subtree'])
        node = pick['node']
        dic['node'] = node
        x, y = index2coord(pick['action'], size)
        make_play(x, y, board)
    return board[0]


def subtree_worker(input_tuple):
    """simulation_workers.py:107-115: (policy, board[S,S,17]) -> subtree with parent None."""
    from .play import new_subtree
    policy, board = input_tuple
    return new_subtree(policy, board.reshape((1,) + tuple(board.shape)), None)
