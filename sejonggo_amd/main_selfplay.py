"""Self-play entry of the package: `python -m sejonggo_amd.main_selfplay [--gpus 0,1,..] [--games-per-gpu G] [--games N]`.

One generation = one worker PROCESS per MI355X (conf['N_GAME_PROCESS'] of them, worker i on conf['GPUs'][i % len]),
each keeping conf['GAMES_PER_GPU'] games resident on its GPU until conf['N_GAMES'] game numbers are used up.
Generations repeat while MODEL_DIR's best model keeps changing under us (a trainer / evaluator promoting a new one);
the first generation that would replay the model just finished ends the run.

API row: `main()` stands where the reference's main_selfplay.py:9-29 stands and prints the same two progress lines; the
reference's own file also runs unchanged on this package's modules (tests/test_entry_path.py).  The parent never touches
the GPU runtime (the best model's name comes from file metadata), so the workers can be forked.
"""
import argparse
import sys

from . import predicting_queue_worker as pq
from . import utils
from .conf import conf
from .selfplay_worker import NoModelSelfPlayWorker


def run_generation(n_workers):
    """Start `n_workers` self-play worker processes and wait for all of them; returns their exit codes."""
    procs = [NoModelSelfPlayWorker(rank) for rank in range(n_workers)]
    for proc in procs:
        proc.start()
    codes = []
    for proc in procs:
        proc.join()
        codes.append(proc.exitcode)
    return codes


def generations():
    """Yields the best model's name once per generation to play, registering / releasing the GPU ids around each."""
    played = None
    while True:
        pq.init_predicting_workers(conf['GPUs'])
        try:
            best = pq.put_name_request("BEST")
            if best == played:
                return
            yield best
            played = best
        finally:
            pq.destroy_predicting_workers(conf['GPUs'])


def main():
    sys.setrecursionlimit(10000)          # host dict-tree helpers recurse like the reference's (main_selfplay.py:10)
    utils.init_directories()
    utils.clean_up_empty()
    for best in generations():
        print("SELF-PLAYING BEST MODEL ", best)
        run_generation(conf['N_GAME_PROCESS'])
    print("No new best model for self-playing. Stopping..")


def _cli(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--gpus", help="comma-separated GPU ids; one worker process per id")
    ap.add_argument("--games-per-gpu", type=int, help="games resident on each GPU")
    ap.add_argument("--games", type=int, help="game numbers to play per model (conf['N_GAMES'])")
    args = ap.parse_args(argv)
    if args.gpus:
        conf['GPUs'] = [int(g) for g in args.gpus.split(",")]
        conf['N_GAME_PROCESS'] = len(conf['GPUs'])
    if args.games_per_gpu:
        conf['GAMES_PER_GPU'] = args.games_per_gpu
    if args.games:
        conf['N_GAMES'] = args.games
    main()


if __name__ == "__main__":
    _cli()
