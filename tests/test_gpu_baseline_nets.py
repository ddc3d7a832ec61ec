"""GPU parity at the sizes BASELINE.json names (VERDICT r2 "Next round" item 1):

* pi / value tolerance of the resident fp16 net against its own fp32 torch module for the reference's 20-block / 256-filter
  topology at 19x19 (model.py:55-95, conf.py N_RESIDUAL_BLOCKS) and the 4-block / 256 net of config 2 at 9x9, on packed
  records board_advance itself wrote, read by the stem kernel under all 8 symmetries, from the opening to a nearly full board;
* the tower convolution at the headline LAUNCH (8 192 positions x 17 x 17, with skip) against fp32 on sampled tiles;
* config 2's batch shape (256 games x 9x9 x 200 sims) against the oracle on sampled slots;
* go_game.GoGame.do_move replaying a scripted golden game on the device.

Tolerance (stated by north_star as "pi / value within stated fp tolerance"): max |dp| <= 2e-3, |dv| <= 5e-3.
"""
import numpy as np
import pytest

from tests.helpers import load, sha8, unpack_mask

pytestmark = pytest.mark.gpu

P_TOL, V_TOL = 2e-3, 5e-3


@pytest.fixture(scope="module")
def L():
    from sejonggo_amd import _lib
    _lib.require_gpu()
    return _lib


def _playout_records(L, S, n, ply, seed):
    """n packed position records after `ply` seeded random legal moves each (passes rare), written by board_advance itself
    (sgo_advance_legal_dev), on the device."""
    import torch
    lib = L.load()
    A, NW, RW = S * S + 1, lib.sgo_plane_words(S), lib.sgo_packed_words(S)
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    cur = torch.zeros((n, RW), dtype=torch.int32, device="cuda")
    nxt = torch.zeros_like(cur)
    legal = torch.full((n, NW), -1, dtype=torch.int32, device="cuda")
    legal[:, NW - 1] = (1 << ((A - 1) % 32 + 1)) - 1
    shifts = torch.arange(32, device="cuda", dtype=torch.int32)
    st = L.stream_ptr()
    for _ in range(ply):
        bits = ((legal.unsqueeze(-1) >> shifts) & 1).reshape(n, NW * 32)[:, :A].float()
        bits[:, A - 1] = 0.01
        mv = torch.multinomial(bits, 1, generator=g).reshape(-1).to(torch.int32)
        L.check(lib.sgo_advance_legal_dev(S, n, L.ptr(cur), None, L.ptr(mv), None, L.ptr(nxt), None, L.ptr(legal), None, st))
        cur, nxt = nxt, cur
    return cur


@pytest.mark.parametrize("S,blocks,plies", [(19, 20, (0, 30, 120, 250)), (9, 4, (0, 10, 30, 60))],
                         ids=["19x19_20block_256", "9x9_4block_256"])
def test_baseline_nets_match_fp32_within_the_stated_tolerance(L, S, blocks, plies):
    """The resident fp16 net on the engine's own route (packed records -> sgo_stem_packed_dev -> tower kernel -> heads) against
    the fp32 torch module of the same weights on the 17-plane tensor of the same records: the reference's 20-block / 256-filter
    topology at 19x19 and config 2's 4-block net at 9x9; 64 positions per ply from the empty board to a nearly full one, every
    one of the 8 symmetries on every batch."""
    import torch
    from sejonggo_amd.net import build_fused_net
    lib = L.load()
    fnet, ref = build_fused_net(S, blocks, 256, name="parity", seed=11)
    assert fnet.packed_ok
    ref = ref.cuda().float()
    n = 64
    worst_p = worst_v = 0.0
    report = []
    for ply in plies:
        recs = _playout_records(L, S, n, ply, seed=1000 + ply)
        for k in range(8):
            x = torch.zeros((n, S, S, 17), dtype=torch.float32, device="cuda")
            L.check(lib.sgo_nn_pack_dev(S, n, L.ptr(recs), None, k, 0, 1, L.ptr(x), L.stream_ptr()))
            p0, v0 = ref.predict_on_batch(x)
            p1, v1 = fnet.predict_packed(recs.data_ptr(), None, n, k)
            assert p1.shape == (n, S * S + 1) and v1.shape == (n, 1)
            dp, dv = float((p1 - p0).abs().max()), float((v1 - v0).abs().max())
            if k == 0:
                report.append((ply, round(float(x[..., :2].sum()) / n, 1), dp, dv))
            worst_p, worst_v = max(worst_p, dp), max(worst_v, dv)
    print("\nNET_TOLERANCE S=%d blocks=%d positions=%d x 8 symmetries: max|dp|=%.3e max|dv|=%.3e; per ply (ply, stones, dp, dv at k=0): %s"
          % (S, blocks, n * len(plies), worst_p, worst_v, report))
    assert worst_p <= P_TOL and worst_v <= V_TOL, report


def test_packed_route_equals_the_tensor_route_of_the_same_net(L):
    """FusedInferenceNet.predict_packed (stem from records, colour plane folded into the bias) against predict_on_batch of the
    SAME fp16 net on the channel-padded tensor (k_stem, the route of put_predict_request callers): the two stems differ only in
    summation order, so policies and values agree far inside the fp32 tolerance."""
    import torch
    from sejonggo_amd.net import build_fused_net
    lib = L.load()
    S = 19
    fnet, _ = build_fused_net(S, 2, 256, name="parity", seed=3)
    recs = _playout_records(L, S, 96, 80, seed=77)
    for k in (0, 3, 6):
        x = torch.zeros((96, S, S, 32), dtype=torch.float16, device="cuda")
        L.check(lib.sgo_nn_pack_dev(S, 96, L.ptr(recs), None, k, 2, 0, L.ptr(x), L.stream_ptr()))
        p0, v0 = fnet.predict_on_batch(x)
        p1, v1 = fnet.predict_packed(recs.data_ptr(), None, 96, k)
        assert float((p1 - p0).abs().max()) <= 5e-4 and float((v1 - v0).abs().max()) <= 2e-3


def test_tower_conv_at_the_headline_launch(L):
    """sgo_conv3x3_tower_dev at the bench's own launch -- 8 192 positions x 17 x 17 x 256, bias + skip + ReLU -- against
    torch fp32 on sampled positions: the first and the last tile, positions spread so that every XCD's tile range is hit
    (tiles are dealt to the 8 XCDs in contiguous ranges), and the slice seam of the entry point."""
    import torch
    import torch.nn.functional as F
    lib = L.load()
    torch.manual_seed(8)
    n, h, wd = 8192, 17, 17
    st = torch.cuda.current_stream().cuda_stream
    x = torch.relu(torch.randn(n, h, wd, 256, device="cuda") * 0.5).half()          # post-ReLU activations like the net's
    w = (torch.randn(256, 3, 3, 256, device="cuda") * 0.03).half()
    b = torch.randn(256, device="cuda").half()
    skip = torch.relu(torch.randn(n, h, wd, 256, device="cuda")).half()
    y = torch.full((n, h, wd, 256), 7.0, device="cuda", dtype=torch.float16)
    L.check(lib.sgo_conv3x3_tower_dev(n, h, wd, x.data_ptr(), w.data_ptr(), b.data_ptr(), skip.data_ptr(), y.data_ptr(), st))
    sample = sorted(set([0, 1, n - 2, n - 1] + [int((j + f) * n / 8) for j in range(8) for f in (0.0, 0.37, 0.999)]
                        + [int(v) for v in np.random.RandomState(3).randint(0, n, size=16)]))
    sample = [min(n - 1, max(0, i)) for i in sample]
    idx = torch.tensor(sample, device="cuda")
    ref = F.conv2d(x[idx].float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), b.float(), padding=1).permute(0, 2, 3, 1)
    ref = torch.relu(ref + skip[idx].float())
    err = (y[idx].float() - ref).abs()
    assert bool((err <= 2e-3 * ref.abs() + 2e-3).all()), float(err.max())
    # nothing outside the tensor was written and every position was (7.0 would survive a skipped tile)
    assert not bool((y == 7.0).all(dim=-1).all(dim=-1).all(dim=-1).any())
    y2 = torch.empty_like(y)
    L.check(lib.sgo_conv3x3_tower_dev(n, h, wd, x.data_ptr(), w.data_ptr(), b.data_ptr(), skip.data_ptr(), y2.data_ptr(), st))
    assert torch.equal(y, y2)
    # the register-fed kernel (k_conv4r, filter bank in fragment order) at the same launch: identical bits
    bank = torch.empty(lib.sgo_conv3x3_tower_packed_bytes(), device="cuda", dtype=torch.uint8)
    L.check(lib.sgo_conv3x3_tower_prepack_dev(w.data_ptr(), bank.data_ptr(), st))
    y2.fill_(7.0)
    L.check(lib.sgo_conv3x3_tower_packed_dev(n, h, wd, x.data_ptr(), bank.data_ptr(), b.data_ptr(), skip.data_ptr(), y2.data_ptr(), st))
    assert torch.equal(y, y2)


def test_config2_batch_shape_equals_the_oracle_on_sampled_slots(L):
    """BASELINE config 2's shape: 256 concurrent 9x9 games, 200 sims per move in rounds of 8 leaves = 2 048-leaf launches.
    Two plies; sampled slots move for move, tree for tree and root table for root table against the oracle."""
    from sejonggo_amd.engine import SelfPlayEngine
    from sejonggo_amd.stub_nets import make_stub
    from tests.test_gpu_engine import _compare_with_oracle
    S, sims, E, G, nm = 9, 200, 8, 256, 2
    net = make_stub("hash", S)
    rng = np.random.RandomState(78)
    noises = rng.dirichlet([0.03] * (S * S + 1), size=G)
    uni = rng.random_sample((G, nm))
    eng = SelfPlayEngine(net, size=S, n_games=G, sims=sims, energy=E, stop_exploration=30, num_moves=nm, komi=5.5,
                         symmetry="identity")
    eng.start_games(np.arange(G), noises=noises, uniforms=uni)
    games = eng.run()
    assert len(games) == G and eng.status.total_moves == G * nm
    assert eng.status.total_evals == G * nm * (1 + sims) and eng.status.none_events == 0
    ms, launches, leaves = eng.advance_timing()
    assert leaves == G * nm * sims and launches == nm * (sims // E)          # every launch carried all 2 048 leaves
    _compare_with_oracle(eng, games, [0, 1, 31, 32, 127, 128, 200, 255], S, sims, E, 30, nm, uni, noises, net)
    eng.close()


def test_go_game_do_move_replays_a_golden_game(L):
    """go_game.GoGame.do_move (go_game.py:25-30) drives make_play on the device: every scripted 9x9 game of rules_S9.npz
    (recorded from the reference) replayed through it, board hash and legal mask after every ply."""
    from sejonggo_amd import go_game, play
    from sejonggo_amd.conf import conf
    z = load("rules_S9.npz")
    S, A = 9, 82
    keep = conf['SIZE']
    conf['SIZE'] = S
    try:
        for gi in range(int(z["n_games"])):
            p = "g%02d_" % gi
            g = go_game.GoGame(size=S, komi=float(z["komi"]))
            assert g.board.shape == (1, S, S, 17) and g.current_player == 1
            assert g.do_move(go_game.RESIGN, None) is None and g.do_move("pass", 1) is None   # string no-ops
            assert np.array_equal(play.legal_moves(g.board), unpack_mask(z[p + "masks"][0], A))
            for ply, (x, y, color) in enumerate(z[p + "moves"]):
                to_play = int(g.board[0, 0, 0, 16])
                # color None would always mean black (current_player never advances, go_game.py:12,27): name the side
                mover = g.do_move(int(x) + S * int(y), to_play if color == 0 else int(color))
                assert mover == (to_play if color == 0 else int(color))
                assert np.array_equal(sha8(g.board), z[p + "hashes"][ply + 1]), (gi, ply)
                assert np.array_equal(play.legal_moves(g.board), unpack_mask(z[p + "masks"][ply + 1], A)), (gi, ply)
        g = go_game.GoGame()
        assert g.do_move(0, None) == 1 and g.do_move(1, None) == 1       # None: black both times, like the reference
        assert g.board[0, 0, 0, 1] == 1 and g.board[0, 0, 1, 1] == 1     # two black stones, white to play
    finally:
        conf['SIZE'] = keep
