"""Two eager SelfPlayEngines on two HIP streams driven by two host threads (no graphs): do our kernels run side by side for long?"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from sejonggo_amd.engine import SelfPlayEngine
from sejonggo_amd.net import build_fused_net
S, G, blocks, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
net, _ = build_fused_net(S, blocks, 256, name="dbg", seed=0)
engs = [SelfPlayEngine(net, n_games=G, size=S, sims=400, energy=8, stop_exploration=30, symmetry="random1", seed=i,
                       stream=torch.cuda.Stream(), raise_on_error=False) for i in range(2)]
prog = [0, 0]
def run(i):
    e = engs[i]
    e.start_games(np.arange(G))
    for k in range(steps):
        e.step()
        prog[i] = k + 1
ts = [threading.Thread(target=run, args=(i,), daemon=True) for i in range(2)]
t0 = time.time()
for t in ts: t.start()
while any(t.is_alive() for t in ts):
    time.sleep(2.0)
    print("%.0f s: steps %s" % (time.time() - t0, prog), flush=True)
    if time.time() - t0 > float(sys.argv[5]):
        print("STALLED", flush=True); os._exit(3)
print("done in %.1f s" % (time.time() - t0))
