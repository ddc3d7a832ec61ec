"""GPU parity tests for the Go-rules kernels, all through the C ABI (libsgo_hip.so):
 * the golden games recorded from the Python reference (bit-exact boards, masks, scores);
 * the CPU oracle on seeded random positions (batched device API);
 * size-independent properties at full batch sizes (fused legal == stand-alone legal, pack/unpack
   round trip, history shift, symmetry group laws)."""
import ctypes as C
import hashlib

import numpy as np
import pytest

from tests.helpers import load, sha8, unpack_mask, name_of

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from sejonggo_amd import _lib
    _lib.require_gpu()
    return _lib


def _replay_gpu(play, z, gi, S):
    p = "g%02d_" % gi
    A = S * S + 1
    board, _ = play.game_init(S)
    moves, masks, hashes = z[p + "moves"], z[p + "masks"], z[p + "hashes"]
    full_at = list(z[p + "full_at"])
    assert np.array_equal(play.legal_moves(board), unpack_mask(masks[0], A))
    for ply, (x, y, color) in enumerate(moves):
        _, mover = play.make_play(int(x), int(y), board, None if color == 0 else int(color))
        if p + "players" in z:
            assert mover == z[p + "players"][ply]
        assert np.array_equal(sha8(board), hashes[ply + 1]), (name_of(z, p + "name"), ply)
        assert np.array_equal(play.legal_moves(board), unpack_mask(masks[ply + 1], A)), (name_of(z, p + "name"), ply)
        if (ply + 1) in full_at:
            k = full_at.index(ply + 1)
            assert np.array_equal(board[0], z[p + "fulls"][k].astype(np.int32))
            assert play.get_winner(board, float(z["komi"])) == tuple(z[p + "winners"][k])
    return board


@pytest.mark.parametrize("S", [5, 7, 9, 13, 19])
def test_golden_rules(L, S):
    from sejonggo_amd import play
    z = load("rules_S%d.npz" % S)
    for gi in range(int(z["n_games"])):
        _replay_gpu(play, z, gi, S)


def test_golden_sgf(L):
    from sejonggo_amd import play
    z = load("sgf_S19.npz")
    want = ["db2fbf7f0bad", "79090a83e78b", "0fe2086efa5d", "44d6ba4da179", "6f45efb35c1e"]
    for gi in range(int(z["n_games"])):
        board = _replay_gpu(play, z, gi, 19)
        assert hashlib.sha1(board.tobytes()).hexdigest()[:12] == want[gi]


def test_reference_style_unit_cases(L):
    """Reads like the reference's TestBoardMethods (test/tests.py:250-330)."""
    from sejonggo_amd.play import game_init, make_play, legal_moves
    board, player = game_init(9)
    make_play(0, 0, board)  # black
    make_play(1, 0, board)  # white
    make_play(8, 9, board)  # black passes
    make_play(2, 1, board)  # white
    make_play(8, 8, board)  # black
    make_play(3, 0, board)  # white
    make_play(2, 0, board)  # black suicides
    assert board[0][0][1][0] == 1 and board[0][0][1][1] == 0
    assert board[0][0][2][0] == 0 and board[0][0][2][1] == 0
    board, player = game_init(9)
    for (x, y) in [(0, 0), (1, 0), (1, 1), (2, 1), (8, 8), (3, 0)]:
        make_play(x, y, board)
    assert legal_moves(board)[2] == 0  # not a suicide when it captures
    board, player = game_init(9)
    for (x, y) in [(0, 1), (1, 0), (1, 1), (2, 1), (8, 8), (3, 0)]:
        make_play(x, y, board)
    assert legal_moves(board)[2] == 1  # suicide is illegal
    with pytest.raises(AssertionError):
        make_play(1, 0, board)          # occupied (play.py:233-234)


@pytest.mark.parametrize("S", [5, 9, 19])
def test_symmetry_golden(L, S):
    from sejonggo_amd import symmetry as sy
    z = load("sym_S%d.npz" % S)
    boards = z["boards"].astype(np.int32)
    fwd = [sy._id, sy.left_diagonal, sy.vertical_axis, sy.horizontal_axis, sy.rotation_90, sy.rotation_180,
           sy.rotation_270, sy.right_diagonal]
    rev = [sy._id, sy.reverse_left_diagonal, sy.reverse_vertical_axis, sy.reverse_horizontal_axis,
           sy.reverse_rotation_90, sy.reverse_rotation_180, sy.reverse_rotation_270, sy.reverse_right_diagonal]
    for k in range(8):
        assert np.array_equal(fwd[k](np.copy(boards)), z["fwd%d" % k].astype(np.int32)), k
        assert np.array_equal(rev[k](np.copy(z["policy"])), z["rev%d" % k]), k


def _random_positions(S, n, seed, max_ply):
    """Seeded positions from oracle playouts (the oracle is the checker here)."""
    from oracle import oracle as ora
    rng = np.random.RandomState(seed)
    out = np.zeros((n, S, S, 17), dtype=np.int32)
    b, _ = ora.game_init(S)
    ply = 0
    target = rng.randint(0, max_ply)
    i = 0
    while i < n:
        if ply >= target:
            out[i] = b[0]
            i += 1
            target = ply + rng.randint(1, 4)
        mask = ora.legal_moves(b)
        legal = np.flatnonzero(mask[:-1] == 0)
        if len(legal) == 0 or ply > max_ply:
            b, _ = ora.game_init(S)
            ply = 0
            target = rng.randint(0, max_ply)
            continue
        emp = np.flatnonzero((b[0, :, :, 0].reshape(-1) == 0) & (b[0, :, :, 1].reshape(-1) == 0))
        a = int(emp[rng.randint(len(emp))]) if rng.rand() < 0.1 else int(legal[rng.randint(len(legal))])
        if rng.rand() < 0.03:
            a = S * S
        ora.make_play(a % S if a < S * S else 0, a // S, b)
        ply += 1
    return out


@pytest.mark.parametrize("S", [9, 19])
def test_batched_device_api_vs_oracle(L, S):
    import torch
    from oracle import oracle as ora
    lib = L.load()
    n = 192 if S == 19 else 256
    A, NW, RW = S * S + 1, lib.sgo_plane_words(S), lib.sgo_packed_words(S)
    boards = _random_positions(S, n, 99 + S, 2 * S * S - 20)
    rng = np.random.RandomState(5)
    moves = np.zeros(n, dtype=np.int32)
    colors = np.zeros(n, dtype=np.int32)
    for i in range(n):
        emp = np.flatnonzero((boards[i, :, :, 0].reshape(-1) == 0) & (boards[i, :, :, 1].reshape(-1) == 0))
        moves[i] = S * S if (len(emp) == 0 or rng.rand() < 0.05) else emp[rng.randint(len(emp))]
        colors[i] = rng.choice([0, 0, 1, -1])
    d_boards = torch.from_numpy(boards).cuda()
    d_packed = torch.zeros((n, RW), dtype=torch.int32, device="cuda")
    d_out = torch.zeros((n, RW), dtype=torch.int32, device="cuda")
    d_legal = torch.zeros((n, NW), dtype=torch.int32, device="cuda")
    d_status = torch.zeros(n, dtype=torch.int32, device="cuda")
    d_moves = torch.from_numpy(moves).cuda()
    d_colors = torch.from_numpy(colors).cuda()
    st = L.stream_ptr()
    L.check(lib.sgo_pack_dev(S, n, L.ptr(d_boards), L.ptr(d_packed), st))
    # round trip
    d_rt = torch.zeros_like(d_boards)
    L.check(lib.sgo_unpack_dev(S, n, L.ptr(d_packed), L.ptr(d_rt), st))
    assert torch.equal(d_rt, d_boards)
    L.check(lib.sgo_advance_legal_dev(S, n, L.ptr(d_packed), None, L.ptr(d_moves), L.ptr(d_colors), L.ptr(d_out), None,
                                      L.ptr(d_legal), L.ptr(d_status), st))
    d_new = torch.zeros_like(d_boards)
    L.check(lib.sgo_unpack_dev(S, n, L.ptr(d_out), L.ptr(d_new), st))
    d_res = torch.zeros((n, 3), dtype=torch.int32, device="cuda")
    L.check(lib.sgo_score_dev(S, n, L.ptr(d_out), None, C.c_double(5.5), L.ptr(d_res), st))
    torch.cuda.synchronize()
    new = d_new.cpu().numpy()
    legal_bits = d_legal.cpu().numpy().view(np.uint32)
    status = d_status.cpu().numpy()
    res = d_res.cpu().numpy()
    for i in range(n):
        b = boards[i:i + 1].copy()
        a = int(moves[i])
        _, mover = ora.make_play(a % S if a < S * S else 0, a // S, b, None if colors[i] == 0 else int(colors[i]))
        assert status[i] == mover, i
        assert np.array_equal(new[i], b[0]), i
        want = ora.legal_moves(b)
        got = np.array([1 - ((legal_bits[i, k >> 5] >> (k & 31)) & 1) for k in range(A)], dtype=np.uint8)
        assert np.array_equal(got, want), i
        w, bl, wh = ora.get_winner(b, 5.5)
        assert (res[i, 0], res[i, 1], res[i, 2] + 5.5) == (w, bl, wh), i


@pytest.mark.parametrize("S", [9, 19])
def test_nn_pack_all_symmetries(L, S):
    import torch
    from oracle import oracle as ora
    lib = L.load()
    n = 24
    RW = lib.sgo_packed_words(S)
    boards = _random_positions(S, n, 7 + S, S * S)
    d_boards = torch.from_numpy(boards).cuda()
    d_packed = torch.zeros((n, RW), dtype=torch.int32, device="cuda")
    st = L.stream_ptr()
    L.check(lib.sgo_pack_dev(S, n, L.ptr(d_boards), L.ptr(d_packed), st))
    idx = torch.arange(n - 1, -1, -1, dtype=torch.int32, device="cuda")  # exercise the index list
    for k in range(8):
        want = ora.sym_board(k, boards)[::-1]
        for layout in (0, 1):
            for dtype, tdt in ((0, torch.float16), (1, torch.float32)):
                shape = (n, S, S, 17) if layout == 0 else (n, 17, S, S)
                out = torch.zeros(shape, dtype=tdt, device="cuda")
                L.check(lib.sgo_nn_pack_dev(S, n, L.ptr(d_packed), L.ptr(idx), k, layout, dtype, L.ptr(out), st))
                got = out.float().cpu().numpy()
                if layout == 1:
                    got = got.transpose(0, 2, 3, 1)
                assert np.array_equal(got, want.astype(np.float32)), (k, layout, dtype)


def test_full_size_properties_19(L):
    """BASELINE-scale batch (1024 games x 8 leaves = 8192 positions, and 131072): properties that need
    no oracle: fused legal set == stand-alone legal kernel on the produced record; history planes shift
    by one ply with colours swapped; in-place == out-of-place; playing only legal non-suicide moves never
    leaves a stone without liberties (checked through the score kernel's stone counts)."""
    import torch
    lib = L.load()
    S = 19
    A, NW, RW = S * S + 1, lib.sgo_plane_words(S), lib.sgo_packed_words(S)
    for n in (8192, 131072):
        g = torch.Generator(device="cuda")
        g.manual_seed(n)
        cur = torch.zeros((n, RW), dtype=torch.int32, device="cuda")
        nxt = torch.zeros_like(cur)
        legal = torch.zeros((n, NW), dtype=torch.int32, device="cuda")
        legal[:, :] = -1
        legal[:, NW - 1] = (1 << ((A - 1) % 32 + 1)) - 1  # all points + pass legal on the empty board
        st = L.stream_ptr()
        shifts = torch.arange(32, device="cuda", dtype=torch.int32)
        for ply in range(40):
            bits = ((legal.unsqueeze(-1) >> shifts) & 1).reshape(n, NW * 32)[:, :A].float()
            bits[:, A - 1] = 0.02  # pass now and then
            moves = torch.multinomial(bits, 1, generator=g).reshape(n).to(torch.int32)
            status = torch.zeros(n, dtype=torch.int32, device="cuda")
            L.check(lib.sgo_advance_legal_dev(S, n, L.ptr(cur), None, L.ptr(moves), None, L.ptr(nxt), None, L.ptr(legal),
                                              L.ptr(status), st))
            assert int((status.abs() != 1).sum()) == 0
            # stand-alone legal kernel agrees with the fused one
            legal2 = torch.zeros_like(legal)
            L.check(lib.sgo_legal_dev(S, n, L.ptr(nxt), None, L.ptr(legal2), st))
            assert torch.equal(legal, legal2), ply
            # history: pairs 0..6 of the old record become pairs 1..7 of the new one (absolute colours: a plain shift)
            assert torch.equal(nxt[:, 2 * NW:16 * NW], cur[:, 0:14 * NW])
            meta = lambda r: (r[:, NW - 1] >> 31) & 1
            assert torch.equal(meta(nxt), meta(cur) ^ 1)
            # in place gives the same record
            inpl = cur.clone()
            L.check(lib.sgo_advance_legal_dev(S, n, L.ptr(inpl), None, L.ptr(moves), None, L.ptr(inpl), None, None, None, st))
            assert torch.equal(inpl, nxt)
            cur, nxt = nxt, cur
        torch.cuda.synchronize()


@pytest.mark.parametrize("S", [5, 7, 9, 13, 19])
def test_advance_kernel_forms_agree(L, S):
    """The three kernel forms behind board_advance (history stream + lane per position, the same in one launch, and one
    half-wavefront per position = csrc/sgo_rows.hpp) give identical records, legal sets and statuses: long seeded playouts
    (captures, suicides, ko shapes, passes, dense end-game boards), colour overrides, occupied and out-of-range moves."""
    import torch
    lib = L.load()
    A, NW, RW = S * S + 1, lib.sgo_plane_words(S), lib.sgo_packed_words(S)
    n = 2048 + 1                          # odd: the last wavefront of the row-per-lane form runs one half only
    g = torch.Generator(device="cuda")
    g.manual_seed(100 + S)
    cur = torch.zeros((n, RW), dtype=torch.int32, device="cuda")
    legal = torch.full((n, NW), -1, dtype=torch.int32, device="cuda")
    legal[:, NW - 1] = (1 << ((A - 1) % 32 + 1)) - 1
    shifts = torch.arange(32, device="cuda", dtype=torch.int32)
    st = L.stream_ptr()
    old = lib.sgo_advance_mode(-1)
    try:
        for ply in range(min(2 * S * S + 8, 330)):
            bits = ((legal.unsqueeze(-1) >> shifts) & 1).reshape(n, NW * 32)[:, :A].float()
            bits[:, A - 1] = 0.02
            moves = torch.multinomial(bits, 1, generator=g).reshape(n).to(torch.int32)
            # a few illegal-by-mask points (suicides / own eyes are executed by make_play), occupied points and bad indices
            r = torch.rand(n, generator=g, device="cuda")
            rnd = torch.randint(0, A - 1, (n,), generator=g, device="cuda", dtype=torch.int32)
            moves = torch.where(r < 0.05, rnd, moves)
            moves = torch.where(r > 0.995, torch.full_like(moves, A + 3), moves)
            colors = torch.randint(-1, 2, (n,), generator=g, device="cuda", dtype=torch.int32)
            colors = torch.where(torch.rand(n, generator=g, device="cuda") < 0.8, torch.zeros_like(colors), colors)
            outs = []
            for mode in (0, 1, 2):
                lib.sgo_advance_mode(mode)
                o = torch.zeros_like(cur)
                lg = torch.zeros_like(legal)
                stt = torch.zeros(n, dtype=torch.int32, device="cuda")
                L.check(lib.sgo_advance_legal_dev(S, n, L.ptr(cur), None, L.ptr(moves), L.ptr(colors), L.ptr(o), None, L.ptr(lg), L.ptr(stt), st))
                ok = (stt.abs() == 1)
                outs.append((o * ok.unsqueeze(1), lg * ok.unsqueeze(1), stt))   # failed plies leave their outputs undefined
            for k in (1, 2):
                assert torch.equal(outs[0][2], outs[k][2]), (S, ply, k, "status")
                assert torch.equal(outs[0][0], outs[k][0]), (S, ply, k, "record")
                assert torch.equal(outs[0][1], outs[k][1]), (S, ply, k, "legal")
            ok = (outs[2][2].abs() == 1).unsqueeze(1)
            cur = torch.where(ok, outs[2][0], cur)
            legal = torch.where(ok, outs[2][1], legal)
        torch.cuda.synchronize()
    finally:
        lib.sgo_advance_mode(old)
