"""Same orchestration as the reference's main_selfplay.py:9-29, importing this package's modules.

The parent process stays GPU-free: init_predicting_workers only registers GPU ids and put_name_request("BEST")
reads the model name from file metadata (predicting_queue_worker.py of this package), so the workers can be
forked exactly as the reference does and each initialises the GPU itself.  Run one worker per MI355X:
conf['N_GAME_PROCESS'] = number of GPUs, conf['GAMES_PER_GPU'] games resident in each."""
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
        init_predicting_workers(GPUs)
        #  Check if we did self-play on this best model or not
        curr_best_model_name = put_name_request("BEST")
        if curr_best_model_name != finished_best_model_name:
            finished_best_model_name = curr_best_model_name
        else:
            print("No new best model for self-playing. Stopping..")
            destroy_predicting_workers(GPUs)
            break
        print("SELF-PLAYING BEST MODEL ", curr_best_model_name)
        workers = [NoModelSelfPlayWorker(i) for i in range(conf['N_GAME_PROCESS'])]
        for p in workers:
            p.start()
        for p in workers:
            p.join()
        destroy_predicting_workers(GPUs)


if __name__ == "__main__":
    main()
