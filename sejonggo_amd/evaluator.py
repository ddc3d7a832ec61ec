"""Drop-in for the evaluation bookkeeping of the reference's evaluator.py (eval_statistic :50-64, promote_best_model
:66-80, clean_up_result :83-85) -- pure host logic around files; the games themselves are
nomodel_self_play.play_game_async("BEST_SYM", "LATEST_SYM", ..., stop_exploration=0) (evaluate_worker.py:137).

Win bookkeeping of the reference: evaluate_worker.py:104-106 touches EVAL_DIR/<latest>/game_%03d/<winner_model>; a game
counts as a win of the tested model when a file named after the model directory exists in the game directory.
Promotion copies MODEL_DIR/<model>.h5 over conf['BEST_MODEL']; here the model file may be the torch checkpoint
<model>.pt (sejonggo_amd.model), copied to the best-model name with the same extension."""
import os
import shutil

from .conf import conf


def save_eval_game(model_name, game_no, winner_model):
    """evaluate_worker.py:104-106."""
    d = os.path.join(conf['EVAL_DIR'], model_name, "game_%03d" % game_no)
    os.makedirs(d, exist_ok=True)
    open(os.path.join(d, str(winner_model)), "a").close()


def eval_statistic():
    """evaluator.py:50-64: {model directory: wins / games} over conf['EVAL_DIR']."""
    result = {}
    root = conf['EVAL_DIR']
    for model_name in os.listdir(root):
        model_dir = os.path.join(root, model_name)
        if not os.path.isdir(model_dir):
            continue
        wins = total = 0
        for game_dir in os.listdir(model_dir):
            if game_dir.startswith('game'):
                total += 1
                if os.path.isfile(os.path.join(model_dir, game_dir, model_name)):
                    wins += 1
        result[model_name] = wins / total if total != 0 else 0
    return result


def clean_up_result(result):
    for model_name in result.keys():
        shutil.rmtree(os.path.join(conf['EVAL_DIR'], model_name))


def promote_best_model(cleanup=True):
    """evaluator.py:66-80: the first model whose win rate exceeds EVALUATE_MARGIN becomes the best model."""
    result = eval_statistic()
    best_stem = os.path.splitext(conf['BEST_MODEL'])[0]
    for model_name in result.keys():
        if result[model_name] > conf['EVALUATE_MARGIN']:
            from .model import _find, drop_sibling
            src = _find(model_name)
            if src is None:
                raise FileNotFoundError("promote_best_model: no file for %r under %r" % (model_name, conf['MODEL_DIR']))
            dst = os.path.join(conf['MODEL_DIR'], best_stem + os.path.splitext(src)[1])
            tmp = "%s.tmp%d" % (dst, os.getpid())
            shutil.copyfile(src, tmp)
            os.replace(tmp, dst)
            drop_sibling(dst)          # a best_model.pt left beside a promoted best_model.h5 would keep being loaded
            if cleanup:
                clean_up_result(result)
            return True
    return False


def elect_model_as_best_model(model):
    """evaluator.py:17-20: N_GAMES of self-play with the new best model, then it replaces conf['BEST_MODEL']."""
    from .model import save_model
    from .self_play import self_play
    self_play(model, n_games=conf['N_GAMES'], mcts_simulations=conf['MCTS_SIMULATIONS'])
    save_model(getattr(model, "net", model), conf['BEST_MODEL'])


def evaluate(best_model, tested_model):
    """evaluator.py:22-47 (the sync path: self_play.play_game on host dict trees, rules and nets on the GPU): EVALUATE_N_GAMES
    games best vs tested at temperature 0; above EVALUATE_MARGIN the tested model is elected.  Models are objects with the
    numpy face of the model contract (predicting_queue_worker._NumpyNet wraps a resident net)."""
    from .self_play import play_game
    from .sgfsave import save_game_data
    total = wins = 0
    for game in range(conf['EVALUATE_N_GAMES']):
        game_data = play_game(best_model, tested_model, conf['MCTS_SIMULATIONS'], stop_exploration=0)
        if game_data['winner_model'] == tested_model.name:
            wins += 1
        total += 1
        save_game_data(best_model.name, game, game_data)
    if wins / total > conf['EVALUATE_MARGIN']:
        print("We found a new best model : %s!" % tested_model.name)
        elect_model_as_best_model(tested_model)
        return True
    return False
