"""Hot-path configuration keys, same names and defaults as the reference's conf.py:3-106 (only the
keys the self-play path reads; hosts/credentials of the reference's cluster config are not part of it)."""
conf = {
    'MODEL_DIR': 'sp_models',
    'EVAL_DIR': 'sp_eval_games',
    'SELF_PLAY_DIR': 'sp_self_play_data',
    'LOG_DIR': 'logs',
    'TMP_DIR': 'temp',
    'GAMES_DIR': 'sp_eval_games',
    'SIMULATION_MODE': "ASYNC",
    'THREAD_SIMULATION': True,
    'SLEEP_SECONDS': 120,
    'EVALUATE_N_GAMES': 100,
    'EVALUATE_MARGIN': .55,
    'BEST_MODEL': 'best_model.h5',
    'SHOW_EACH_MOVE': False,
    'SHOW_END_GAME': False,
    'GPUs': [0, 1, 2, 3, 4, 5, 6, 7],
    'PREDICTING_BATCH_SIZE': 32,
    'N_RESIDUAL_BLOCKS': 20,
    'N_GAMES': 5000,
    'GAME_RANGE': [0, 5000],
    'MCTS_SIMULATIONS': 1600,
    'N_GAME_PROCESS': 32,
    'ENERGY': 8,
    'SIZE': 19,
    'KOMI': 5.5,
    'STOP_EXPLORATION': 30,
    'MCTS_BATCH_SIZE': 100,
    'DIRICHLET_ALPHA': .03,
    'DIRICHLET_EPSILON': .25,
    'RESIGNATION_PERCENT': .10,
    'RESIGNATION_ALLOWED_ERROR': .05,
    'TRAINING_SERVER': None,
    'SGF_ENABLED': False,
    # build-side knobs (not in the reference)
    'GAMES_PER_GPU': 1024,       # concurrent game slots resident on one MI355X
    'NET_DTYPE': 'fp16',
    'SYMMETRY_MODE': 'random1',  # 'random1' = reference behaviour (symmetry.py:127-132); 'avg8' = 8-fold averaging
    'ENGINE_HALVES': 0,          # 2: two half-populations alternating on two HIP streams (engine.DualEngine), captured rounds; 1: one population; 0: by round size
    'ENGINE_GRAPH': False,       # every engine round one captured launch chain (hipGraph): for launch-bound configurations
    'BLOCKS_PER_GAME': 0,        # PRIVATE tree blocks per resident game; 0 = the engine's default (8 * sims + 128)
    'SHARED_BLOCKS': 0,          # tree blocks shared by all games of a GPU; 0 = default (2 * sims per game) unless BLOCKS_PER_GAME is set, < 0 = default
    'WRITER_THREADS': 2,         # sample-file writer threads per self-play worker (off the stepping thread)
    'WRITER_PROCESSES': 0,       # > 0: that many torch-free writer PROCESSES instead (own libhdf5 each; for small boards)
    'NET_CHANNELS': 256,         # filters of the tower (model.py:58 hard-codes 256)
    'COMPAT_Z': True,            # reproduce sgfsave.py:56 value_target quirk
    'COMPAT_LATEST_SYM': True,   # reproduce predicting_queue_worker.py:92: LATEST_SYM requests are answered by the BEST model
    'COMPAT_WINNER_MODEL': True, # reproduce nomodel_self_play.py:247: winner_model names the loser when model1 plays white
    'SELFPLAY_WORKER_ENGINE': 'sync',  # SelfPlayWorker: 'sync' = self_play.model_self_play like the reference, 'device' = device engine
    'WRITE_NPZ_TWIN': True,      # without h5py: keep sample.npz beside the spec-written sample.h5
}
