"""Which part of a per-move tuple exchange slows the following rounds?  (config 2 engine loop, one rank, RCCL)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
from sejonggo_amd.engine import SelfPlayEngine
from sejonggo_amd.net import build_fused_net
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
from sejonggo_amd.distributed import init_from_env, TupleGather, tuple_dtype
init_from_env("nccl")
S, G = 9, 256
net, _ = build_fused_net(S, 4, 256, name="dbg", seed=0)
eng = SelfPlayEngine(net, n_games=G, size=S, sims=200, energy=8, stop_exploration=30, symmetry="random1", seed=1, raise_on_error=False)
eng.start_games(np.arange(G))
dev = torch.device("cuda", 0)
host = torch.empty(150000, dtype=torch.uint8, pin_memory=True)
dbuf = torch.zeros(150000, dtype=torch.uint8, device=dev)
cnt = torch.ones(1, dtype=torch.int64, device=dev)
outs = [torch.zeros(1, dtype=torch.int64, device=dev)]
blk_out = [torch.empty_like(dbuf)]
tg = TupleGather(tuple_dtype(S))
tg2 = TupleGather(tuple_dtype(S), collective="all_gather")
recs = np.zeros(G, dtype=tuple_dtype(S))

def move():
    t = eng.status.total_moves + G
    while eng.status.total_moves < t:
        st = eng.step()
        if st.n_active < G:
            res = eng.results(); again = [s for s in range(G) if res[s]["done"] == 1]
            eng.drain(); eng.start_games(again)
    eng.drain()
    for s in range(G):
        eng.records[s] = []

sl = tg._slot(0, recs.nbytes)
raw = np.ascontiguousarray(recs).view(np.uint8).reshape(-1)
def a_cpu_copy():
    sl["host_in"][:raw.size].copy_(torch.from_numpy(raw))
def b_h2d_block():
    sl["block"][:raw.size].copy_(sl["host_in"][:raw.size], non_blocking=True)
def c_cnt():
    sl["cnt_host"][0] = 256
    sl["cnt"].copy_(sl["cnt_host"], non_blocking=True)
def d_allgather():
    w = dist.all_gather(sl["counts"], sl["cnt"], async_op=True); w.wait(); return w
def e_all():
    a_cpu_copy(); b_h2d_block(); c_cnt(); return d_allgather()
def f_all_keep():
    keep.append(e_all())
    del keep[:-2]
keep = []
modes = {
    "nothing": lambda: None,
    "a_cpu_copy": a_cpu_copy,
    "b_h2d_block": b_h2d_block,
    "c_cnt": c_cnt,
    "d_allgather": d_allgather,
    "e_all": e_all,
    "f_all_keep_work": f_all_keep,
    "nothing_end": lambda: None,
}
def s1():
    tg.n_submitted += 1
    return tg._stage1(recs)
modes.update({"tg_stage1": s1, "tg_submit": lambda: tg.submit(recs), "tg_submit_flush": lambda: (tg.submit(recs), tg.flush()),
              "nothing_last": lambda: None})
for name, fn in modes.items():
    eng.start_games(np.arange(G))            # every phase plays plies 0..9 of fresh games: same work per phase
    eng.records.clear()
    for _ in range(2):
        move(); fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(8):
        move(); fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
    print("%-24s %.2f ms per move" % (name, dt * 1e3), flush=True)
tg.flush()
eng.close()
dist.destroy_process_group()
