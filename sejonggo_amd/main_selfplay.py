"""Same orchestration as the reference's main_selfplay.py:9-29, importing this package's modules."""
import sys

from .conf import conf
from .predicting_queue_worker import init_predicting_workers, destroy_predicting_workers, put_name_request
from .selfplay_worker import NoModelSelfPlayWorker
from .utils import init_directories, clean_up_empty


def main():
    sys.setrecursionlimit(10000)
    init_directories()
    clean_up_empty()
    GPUs = conf['GPUs']
    finished_best_model_name = None
    while True:
        init_predicting_workers(GPUs[:1])
        curr_best_model_name = put_name_request("BEST")
        if curr_best_model_name != finished_best_model_name:
            finished_best_model_name = curr_best_model_name
        else:
            print("No new best model for self-playing. Stopping..")
            destroy_predicting_workers(GPUs[:1])
            break
        print("SELF-PLAYING BEST MODEL ", curr_best_model_name)
        destroy_predicting_workers(GPUs[:1])   # workers load their own copy after the fork
        workers = [NoModelSelfPlayWorker(i) for i in range(conf['N_GAME_PROCESS'])]
        for p in workers:
            p.start()
        for p in workers:
            p.join()


if __name__ == "__main__":
    main()
