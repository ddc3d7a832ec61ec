import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sejonggo_amd.engine import DualEngine, SelfPlayEngine
from sejonggo_amd.net import build_fused_net
from sejonggo_amd.conf import conf
conf['ROUND_TIMEOUT_S'] = float(os.environ.get('SGO_ROUND_TIMEOUT_S', '20'))
mode = sys.argv[1]
S, G = int(sys.argv[2]), int(sys.argv[3]); blocks = int(sys.argv[4]); sims = int(sys.argv[5])
net, _ = build_fused_net(S, blocks, 256, name="dbg", seed=0)
kw = dict(n_games=G, size=S, sims=sims, energy=8, stop_exploration=30, symmetry=sys.argv[7] if len(sys.argv) > 7 else "random1", seed=1, raise_on_error=False)
eng = DualEngine(net, allow_large=True, **kw) if mode == "dual" else SelfPlayEngine(net, graph=(mode == "graph"), **kw)
eng.start_games(np.arange(G))
t0 = time.time()
for i in range(int(sys.argv[6])):
    st = eng.step()
    if st.n_records >= G:
        eng.drain()
        for k in range(G):
            eng.records[k] = []
    if st.n_active < G:
        print("n_active", st.n_active, "at step", i, [(e.status.n_active, e.status.n_done, e.status.n_eval) for e in getattr(eng, "halves", [eng])], flush=True)
    if st.error:
        print("ERROR at step", i, st.error, st.error_game, "moves", st.total_moves, flush=True)
        break
    if i % 26 == 0:
        print(i, "act", st.n_active, "eval", st.n_eval, "moves", st.total_moves, "evals", st.total_evals, "%.2fs" % (time.time() - t0), flush=True)
if mode == "dual":
    eng.sync()
res = eng.results()
print("high water", sorted(int(r["blocks_high_water"]) for r in res)[-5:], "moves", sorted(int(r["n_moves"]) for r in res)[:3], "time %.2f" % (time.time() - t0))
if mode != "dual":
    p, v = eng._policy, eng._value
    if p is not None:
        print("policy finite", bool(torch.isfinite(p).all()), float(p.sum(1).min()), float(p.sum(1).max()), "value", float(v.min()), float(v.max()))
    else:
        print("static policy finite", bool(torch.isfinite(eng._pol_static).all()), float(eng._pol_static[:64].sum(1).min()), float(eng._pol_static[:64].sum(1).max()))
eng.close()
