"""CPU model of the hand-counted `s_waitcnt vmcnt(N)` waits of the tower convolution kernel (csrc/sgo_conv8w.hpp).

The kernel never waits for vmcnt(0) inside its K loop: each wait names how many of the wave's YOUNGEST vector-memory
operations may still be in flight, and everything older -- in particular the LDS-DMA whose data the next phase reads --
has then landed (vector-memory operations of a wave retire in issue order).  A wrong count is a data race that only shows
as run-to-run differences on the GPU; round 1 shipped such a bug for a few hours (the window piece of a K-tile is issued
by SOME waves only -- those whose rows of the piece exist -- and the count must follow the wave).  This model replays,
per wave, the order in which the kernel issues its DMAs / loads / stores and checks at every wait that what is read next is
older than the N youngest.  It is tied to the source: the wait expressions and the piece condition are read from the
header and the test fails when they change, so the model cannot silently drift from the kernel."""
import os
import re

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sejonggo_amd", "csrc", "sgo_conv8w.hpp")


def _src():
    return open(HDR).read()


def test_model_is_in_step_with_the_kernel_source():
    s = _src()
    # the piece condition: taps 0..4 of a chunk stage the next chunk's window; piece 4 only by waves whose rows exist
    assert "#define SGW_WP_COND(CP, TP) ((TP) >= 0 && (TP) < 5 && ((CP) == 0 || kk == 0) && ((TP) < 4 || (4 * 8 + swid) * 8 < NROWS))" in s
    # phase B: 4 weight DMAs, then the piece, then the counted wait
    assert "SGW_WAIT_B(4 + (wp_ ? 1 : 0) + (wpprev_ ? 1 : 0));" in s
    assert re.search(r"SGW_STAGE_BK\(BUF_, 0, koff_\);\s*\\\s*SGW_STAGE_BK\(BUF_, 1, koff_\);\s*\\\s*const bool wp_", s)
    assert "_Pragma(\"unroll\") for (int i_ = 0; i_ < 2; i_++)" in s          # 2 DMAs per (buffer, granule) and wave
    assert "if (id_ * 8 < NROWS)" in s and "const int id_ = (pc) * 8 + swid;" in s
    # prologue: 5 pieces, weights of K-tiles 0 and 1, wait for all but K-tile 1's
    assert re.search(r"for \(int pc = 0; pc < 5; pc\+\+\) SGW_STAGE_W\(0, pc\);\s*SGW_STAGE_B\(0, 0, 0\);\s*SGW_STAGE_B\(0, 1, 0\);\s*"
                     r"SGW_STAGE_B\(1, 0, 1\);\s*SGW_STAGE_B\(1, 1, 1\);", s)
    assert "SGW_VMWAIT(4);\n    asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");   // the zero row" in s
    # tail of the loop and the epilogue
    assert "const bool last2_ = (CP) == 1 && (T) >= 7 && kk == 1;" in s
    assert "SGW_VMWAIT(0);   /* K-tile 34: K-tile 35's weights */" in s
    assert "if (HAS_SKIP && (CP) == 1 && (T) == 8 && kk == 1) SGW_SKIP_LO(0);" in s and "if (HAS_SKIP) SGW_SKIP_LO(1);" in s
    assert s.count("asm volatile(\"s_waitcnt vmcnt(8)\" ::: \"memory\");") == 1
    assert "for (int j = 0; j < 8; j++) SGW_STAGE_SKIP(1, j);" in s


class Wave(object):
    """Issue log of one wave: every vector-memory operation gets a tag; wait(n, need) checks `need` is older than the n youngest."""

    def __init__(self):
        self.ops = []

    def issue(self, tag):
        self.ops.append(tag)

    def wait(self, n, need, where):
        young = self.ops[len(self.ops) - n:] if n else []
        for tag in need:
            assert tag in self.ops, "%s: %r was never issued" % (where, tag)
            assert tag not in young, "%s: vmcnt(%d) leaves %r in flight, but it is read next" % (where, n, tag)


def piece_cond(CP, TP, kk, wid, nrows):
    return TP >= 0 and TP < 5 and (CP == 0 or kk == 0) and (TP < 4 or (4 * 8 + wid) * 8 < nrows)


def run_wave(wid, W, has_skip, count_piece_per_wave=True):
    """Replays one wave.  count_piece_per_wave=False is the bug of commit 3e0d1de's parent: the wait counted the piece as
    if every wave issued it."""
    nrows = 256 + 2 * (W + 1)
    w = Wave()
    for pc in range(5):
        if (pc * 8 + wid) * 8 < nrows:
            w.issue(("win", 0, pc))
    for t in (0, 1):
        for g in (0, 1):
            for i in (0, 1):
                w.issue(("wgt", t, g, i))
    need = [("wgt", 0, g, i) for g in (0, 1) for i in (0, 1)] + [("win", 0, pc) for pc in range(5) if (pc * 8 + wid) * 8 < nrows]
    w.wait(4, need, "prologue")
    for kk in (0, 1):
        for CP in (0, 1):
            for T in range(9):
                t = 18 * kk + 9 * CP + T
                cc = 2 * kk + CP
                where = "K-tile %d (wave %d, w %d)" % (t, wid, W)
                # phase A reads weights[t] and (G = 0) the window of chunk cc: both must have landed at an EARLIER wait
                if has_skip and CP == 1 and T == 8 and kk == 1:
                    for j in range(4):
                        w.issue(("skip", 0, j))
                last2 = CP == 1 and T >= 7 and kk == 1
                if not last2:
                    for g in (0, 1):
                        for i in (0, 1):
                            w.issue(("wgt", t + 2, g, i))
                    wp, wpprev = piece_cond(CP, T, kk, wid, nrows), piece_cond(CP, T - 1, kk, wid, nrows)
                    if wp:
                        w.issue(("win", cc + 1, T))
                    if count_piece_per_wave:
                        n = 4 + (1 if wp else 0) + (1 if wpprev else 0)
                    else:
                        gen = lambda TP: TP >= 0 and TP < 5 and (CP == 0 or kk == 0)
                        n = 4 + (1 if gen(T) else 0) + (1 if gen(T - 1) else 0)
                    # the next K-tile's phase A reads weights[t+1]
                    need = [("wgt", t + 1, g, i) for g in (0, 1) for i in (0, 1)]
                    # the first K-tile of the next chunk reads that chunk's window: all its pieces must be retired by then
                    if T == 8 and cc < 3:
                        need += [("win", cc + 1, pc) for pc in range(5) if (pc * 8 + wid) * 8 < nrows]
                    w.wait(n, need, where)
                elif T == 7:
                    w.wait(0, [("wgt", 35, g, i) for g in (0, 1) for i in (0, 1)], where)
                else:
                    if has_skip:
                        for j in range(4, 8):
                            w.issue(("skip", 0, j))
    # epilogue: bias (4 loads), the hi half's skip rows, then per half: wait, process, 8 row stores
    for q in range(4):
        w.issue(("bias", q))
    if has_skip:
        for j in range(8):
            w.issue(("skip", 1, j))
    for hf in (0, 1):
        if has_skip:
            w.wait(8, [("skip", hf, j) for j in range(8)] + [("bias", q) for q in range(4)], "epilogue half %d" % hf)
        elif hf == 0:
            w.wait(0, [("bias", q) for q in range(4)], "epilogue (no skip)")
        for j in range(8):
            w.issue(("store", hf, j))
    return w


def test_every_counted_wait_retires_what_is_read_next():
    for W in (5, 7, 9, 13, 17, 19):
        for has_skip in (False, True):
            for wid in range(8):
                run_wave(wid, W, has_skip)


def test_window_pieces_are_retired_two_k_tiles_before_their_chunk_starts():
    """Stronger than needed for correctness, and what the kernel's comment promises: a chunk's window pieces are issued in
    taps 0..4 and all retired by the wait of tap 6 (the barrier of tap 6 publishes them to the other waves)."""
    for W in (7, 17, 19):
        for wid in range(8):
            nrows = 256 + 2 * (W + 1)
            w = Wave()
            ops_at_wait = {}
            # replay again, recording the youngest set after each wait of chunk 0
            for pc in range(5):
                if (pc * 8 + wid) * 8 < nrows:
                    w.issue(("win", 0, pc))
            for t in (0, 1):
                for g in (0, 1):
                    for i in (0, 1):
                        w.issue(("wgt", t, g, i))
            for T in range(9):
                for g in (0, 1):
                    for i in (0, 1):
                        w.issue(("wgt", T + 2, g, i))
                wp, wpprev = piece_cond(0, T, 0, wid, nrows), piece_cond(0, T - 1, 0, wid, nrows)
                if wp:
                    w.issue(("win", 1, T))
                n = 4 + (1 if wp else 0) + (1 if wpprev else 0)
                ops_at_wait[T] = set(w.ops[len(w.ops) - n:])
            assert not any(tag[0] == "win" for tag in ops_at_wait[6]), (W, wid)


def test_the_model_catches_the_round_1_race():
    """Counting the window piece as if every wave issued it (the bug fixed in 3e0d1de) must be flagged for the waves that
    do not issue piece 4 -- which waves those are depends on the board width."""
    import pytest
    flagged = []
    for W in (7, 17, 19):
        for wid in range(8):
            try:
                run_wave(wid, W, True, count_piece_per_wave=False)
            except AssertionError as e:
                assert "in flight" in str(e)
                flagged.append((W, wid))
    nrows17 = 256 + 2 * 18
    assert (17, 7) in flagged and all(((4 * 8 + wid) * 8 >= 256 + 2 * (W + 1)) for W, wid in flagged)
    assert [(W, wid) for W, wid in flagged if W == 17] == [(17, wid) for wid in range(8) if (32 + wid) * 8 >= nrows17]


# ------------------------------------------------------------------------------------------------ k_conv4w
HDR4 = os.path.join(os.path.dirname(HDR), "sgo_conv4w.hpp")


def test_conv4w_model_is_in_step_with_the_kernel_source():
    s = open(HDR4).read()
    assert "const bool boundary_ = (T) == 8 && cc < 3;" in s
    assert "if ((VAR & 4) && boundary_) S4_STAGE_WP((cc + 1) * 128, 0, 4);" in s          # early pieces: rows [0, 128), 4 per wave
    assert "if ((VAR & 4) && boundary_) S4_VMWAIT(8);" in s and "else S4_VMWAIT(4);" in s
    assert "if (boundary_) S4_STAGE_WP((cc + 1) * 128, (VAR & 4) ? 4 : 0, 10);" in s
    assert "if ((VAR & 6) != 6) S4_VMWAIT(0);" in s
    assert "else if (nlate == 6) S4_VMWAIT(10);" in s and "else if (nlate == 5) S4_VMWAIT(9);" in s and "else if (nlate == 4) S4_VMWAIT(8);" in s
    assert re.search(r"if \(\(VAR & 6\) == 6 && \(T\) == 0 && cc > 0\) \{[^}]*S4_VMWAIT\(0\);[^}]*S4_KBARRIER\(\);", s)
    assert "for (int pc = 4; pc < 10; pc++) nlate += ((pc * 4 + wid) * 8 < NROWS) ? 1 : 0;" in s
    assert "const int id_ = pc_ * 4 + swid;" in s and "if (id_ * 8 < NROWS)" in s
    assert "S4_VMWAIT(0);                        /* K-tile 34: K-tile 35's weights */" in s
    assert re.search(r"S4_STAGE_W\(0\);\s*S4_STAGE_BK\(0, 0, 0\);\s*S4_STAGE_BK\(0, 1, 0\);\s*S4_STAGE_BK\(1, 0, CIN \* 2\);\s*S4_STAGE_BK\(1, 1, CIN \* 2\);", s)
    assert "S4_VMWAIT(4);\n    asm volatile(\"s_waitcnt lgkmcnt(0)\" ::: \"memory\");   // the zero area" in s
    assert "#define S4_SHIFT(T) (((T) / 3 == 0 ? -W : (T) / 3 == 2 ? W : 0) + (T) % 3 - 1)" in s
    assert "default: return launch_var<7>(n, h, w, x, wgt, bias, skip, y, st);" in s          # the modelled variant is the shipped one


def run_wave4(wid, W, has_skip, early=4):
    """k_conv4w: 4 waves, single-buffered window.  Besides the vmcnt accounting this checks the LDS LIFETIME claim behind the
    early restage: the window rows overwritten before the last tap's phase B are rows that phase B no longer reads."""
    nrows = 256 + 2 * (W + 1)
    w = Wave()

    def pieces(chunk, pc0, pc1):
        out = []
        for pc in range(pc0, pc1):
            idn = pc * 4 + wid
            if idn * 8 < nrows:
                w.issue(("win", chunk, idn))
                out.append(idn)
        return out

    pieces(0, 0, 10)
    for t in (0, 1):
        for g in (0, 1):
            for i in (0, 1):
                w.issue(("wgt", t, g, i))
    w.wait(4, [("wgt", 0, g, i) for g in (0, 1) for i in (0, 1)] + [("win", 0, pc * 4 + wid) for pc in range(10) if (pc * 4 + wid) * 8 < nrows],
           "prologue")
    for cc in range(4):
        for T in range(9):
            t = 9 * cc + T
            where = "K-tile %d (wave %d, w %d)" % (t, wid, W)
            boundary = T == 8 and cc < 3
            if T == 0 and cc > 0:
                # phase A read rows [0, 128) (checked below); before phase B everything of the new window must be there
                lo_rows_max = (W + 1) + (-W - 1) + 63 + 48 + 15
                assert lo_rows_max < 128
                w.wait(0, [("win", cc, pc * 4 + wid) for pc in range(10) if (pc * 4 + wid) * 8 < nrows], where + " late pieces")
            if boundary:
                got = pieces(cc + 1, 0, early)
                # rows restaged now must lie below everything phase B of this tap still reads: min row = 128 + HALO + shift(8)
                lowest_read = 128 + (W + 1) + (W + 1)
                assert all((idn + 1) * 8 <= lowest_read for idn in got), where
                assert len(got) == early                     # every wave issues the same number: the counted wait relies on it
            last2 = cc == 3 and T >= 7
            if not last2:
                for g in (0, 1):
                    for i in (0, 1):
                        w.issue(("wgt", t + 2, g, i))
                w.wait(4 + (early if boundary else 0), [("wgt", t + 1, g, i) for g in (0, 1) for i in (0, 1)], where)
            elif T == 7:
                w.wait(0, [("wgt", 35, g, i) for g in (0, 1) for i in (0, 1)], where)
            if boundary:
                late = pieces(cc + 1, early, 10)
                n = {6: 10, 5: 9, 4: 8}.get(len(late), 0) if early == 4 else 0
                # the next tap's phase A reads rows [0, 128): pieces 0..15, i.e. this wave's early ones
                w.wait(n, [("win", cc + 1, pc * 4 + wid) for pc in range(4)], where + " boundary (early pieces)")
    for q in range(4):
        w.issue(("bias", q))
    if has_skip:
        for hf in (0, 1):
            for j in range(8):
                w.issue(("skip", hf, j))
    for hf in (0, 1):
        if has_skip:
            w.wait(8, [("skip", hf, j) for j in range(8)] + [("bias", q) for q in range(4)], "epilogue half %d" % hf)
        elif hf == 0:
            w.wait(0, [("bias", q) for q in range(4)], "epilogue (no skip)")
        for j in range(8):
            w.issue(("store", hf, j))
    return w


def test_conv4w_counted_waits_and_window_lifetime():
    for W in (1, 3, 5, 7, 9, 13, 17, 19):
        for has_skip in (False, True):
            for wid in range(4):
                run_wave4(wid, W, has_skip)


def test_conv4w_model_rejects_a_too_early_restage():
    """Five early pieces per wave (rows [0, 160)) would be fine at w = 17 but overwrite rows that phase B still reads on
    narrower boards: the lifetime assertion must catch it (this was the first draft of the optimisation)."""
    import pytest
    for wid in range(4):
        run_wave4(wid, 17, True, early=5)
    with pytest.raises(AssertionError):
        run_wave4(3, 7, True, early=5)           # wave 3's fifth piece covers rows 152..159; phase B reads from row 144 on


# ---- k_conv4r (csrc/sgo_conv4r.hpp): the weights go L2 -> registers; three rotating register sets; both tile forms -------------
HDR4R = os.path.join(os.path.dirname(HDR), "sgo_conv4r.hpp")


def test_conv4r_model_is_in_step_with_the_kernel_source():
    s = open(HDR4R).read()
    assert "constexpr int SLO_ = (2 * (T)) % 3, SHI_ = (2 * (T) + 1) % 3, SNX_ = (2 * (T) + 2) % 3;" in s
    assert s.count("(T) == 8 && cc < 3;") == 2 and s.count("(T) == 0 && cc > 0;") == 2
    # grouped form (R4_TILE): L(2t+2) at the start of phase A, L(2t+3) between the two MFMA groups of phase B
    assert re.search(r"R4_LOADW\(SNX_\);[^\n]*\n\s*R4_READ_A\(0, T\);", s)
    assert "if (restaged_) R4_VMWAIT_LATE(8);" in s and "else R4_VMWAITW(8);" in s
    assert "if (boundary_) R4_VMWAIT(8);" in s and "else R4_VMWAITW(4);" in s
    assert "if (!(VAR & 2)) R4_LOADW(SLO_);" in s and "R4_VMWAIT_LATE(4);" in s
    # spread form (R4_TILE_S): two pieces of L(2t+2) in each half of phase A's burst, L(2t+3) in the second half of phase B's
    assert "if (restaged_) R4_VMWAIT_LATE(4);" in s and "if (boundary_) R4_VMWAIT(6);" in s and "else R4_VMWAITW(2);" in s
    assert "R4_G2(0, 0, SLO_, SNX_, 0);" in s and "R4_G2(0, 1, SHI_, SNX_, 2);" in s and "R4_G4(1, 1, SHI_, SLO_);" in s
    assert "R4_VMWAIT_LATE(8);" in s
    assert "if (nlate == 6) { R4_VMWAIT_SUM(base, 6); }" in s and "else { R4_VMWAIT_SUM(base, 0); }" in s
    assert "for (int pc = 4; pc < 10; pc++) nlate += ((pc * 4 + wid) * 8 < NROWS) ? 1 : 0;" in s
    assert re.search(r"if \(!\(VAR & 32\)\) R4_STAGE_WP\(0, 0, 10\);[^\n]*\n\s*R4_LOADW\(0\);\s*R4_LOADW\(1\);", s)
    assert "R4_VMWAIT(8);                                        // the window has landed; in flight: L(0), L(1)" in s
    assert "R4_STAGE_WP((cc + 1) * 128, 0, 4);" in s and "R4_STAGE_WP((cc + 1) * 128, 4, 10);" in s
    assert "default: return launch_var<1>(n, h, w, x, wpk, bias, skip, y, st);" in s       # grouped form + priorities is what ships


def run_wave4r(wid, W, has_skip, spread, drop_drain=False):
    """Issue order and counted waits of one wave of k_conv4r.  Besides 'what is read next has landed' this checks the REGISTER
    claim of the three-set rotation: a load group is issued into a set only after the last MFMA group that reads the set's
    previous contents."""
    nrows = 256 + 2 * (W + 1)
    w = Wave()
    nlate = sum(1 for pc in range(4, 10) if (pc * 4 + wid) * 8 < nrows)
    late_wait = lambda base: base + (nlate if nlate in (4, 5, 6) else 0)
    holds = {}                                   # set -> load group it holds (or is about to receive)
    busy = {}                                    # set -> True while MFMA groups of the current K-tile still read it

    def pieces(chunk, pc0, pc1):
        out = []
        for pc in range(pc0, pc1):
            idn = pc * 4 + wid
            if idn * 8 < nrows:
                w.issue(("win", chunk, idn))
                out.append(("win", chunk, idn))
        return out

    def load(j, s, ps=range(4)):
        assert not busy.get(s), "load group %d into set %d while MFMAs still read it" % (j, s)
        holds[s] = j
        for p in ps:
            w.issue(("L", j, p))

    def need(j, s, n, where):
        assert holds.get(s) == j, "%s: set %d holds group %r, not %d" % (where, s, holds.get(s), j)
        w.wait(n, [("L", j, p) for p in range(4)], where)

    win0 = pieces(0, 0, 10)
    load(0, 0)
    load(1, 1)
    w.wait(8, win0, "prologue")
    for cc in range(4):
        for T in range(9):
            t = 9 * cc + T
            slo, shi, snx = (2 * T) % 3, (2 * T + 1) % 3, (2 * T + 2) % 3
            where = "K-tile %d (wave %d, w %d, %s)" % (t, wid, W, "spread" if spread else "grouped")
            boundary, restaged = T == 8 and cc < 3, T == 0 and cc > 0
            busy[slo] = busy[shi] = True
            # ---- phase A
            if not spread:
                load(2 * t + 2, snx)
            need(2 * t, slo, (late_wait(8) if restaged else 8) if not spread else (late_wait(4) if restaged else 4), where + " lo")
            if boundary:
                early = pieces(cc + 1, 0, 4)
                assert len(early) == 4
                assert all((tag[2] + 1) * 8 <= 128 + 2 * (W + 1) for tag in early), where      # rows phase B no longer reads
            if spread:
                load(2 * t + 2, snx, (0, 1))
                need(2 * t + 1, shi, 6 if boundary else 2, where + " hi")
                for p in (2, 3):
                    w.issue(("L", 2 * t + 2, p))
            else:
                need(2 * t + 1, shi, 8 if boundary else 4, where + " hi")
            if restaged:                         # barrier: every wave's late pieces must have been retired by the wait above
                late = [("win", cc, pc * 4 + wid) for pc in range(4, 10) if (pc * 4 + wid) * 8 < nrows]
                w.wait(2 if spread else 4, late, where + " late pieces")
            # ---- phase B
            if boundary:
                pieces(cc + 1, 4, 10)
            busy[slo] = False                    # MFMA(1, lo) is the set's last reader
            load(2 * t + 3, slo)
            busy[shi] = False
            if boundary:
                w.wait(late_wait(8 if spread else 4), early, where + " boundary (early pieces)")
    if not drop_drain:
        w.wait(0, [("L", 72, p) for p in range(4)] + [("L", 73, p) for p in range(4)], "drain of the look-ahead loads")
    for q in range(4):
        w.issue(("bias", q))
    if has_skip:
        for hf in (0, 1):
            for j in range(8):
                w.issue(("skip", hf, j))
    for hf in (0, 1):
        if has_skip:
            w.wait(8, [("skip", hf, j) for j in range(8)] + [("bias", q) for q in range(4)], "epilogue half %d" % hf)
        elif hf == 0:
            w.wait(0, [("bias", q) for q in range(4)], "epilogue (no skip)")
        for j in range(8):
            w.issue(("store", hf, j))
    return w


def test_conv4r_counted_waits_and_register_sets():
    for spread in (False, True):
        for W in (1, 3, 5, 7, 9, 13, 17, 19):
            for has_skip in (False, True):
                for wid in range(4):
                    run_wave4r(wid, W, has_skip, spread)


def test_conv4r_model_rejects_a_weaker_wait():
    """The model must notice a wait that leaves a needed group in flight: replay with every 'hi' wait one group too weak."""
    import pytest
    real = Wave.wait

    def weak(self, n, need, where):
        return real(self, n + 4 if where.endswith(" hi") else n, need, where)

    Wave.wait = weak
    try:
        with pytest.raises(AssertionError):
            run_wave4r(0, 17, True, False)
        with pytest.raises(AssertionError):
            run_wave4r(0, 17, True, True)
    finally:
        Wave.wait = real


def test_conv4r_isa_leaves_load_destinations_alone_until_their_wait():
    """The compiled k_conv4r (both instantiations that ship): between a register load's issue and the counted wait that retires it,
    no instruction reads or writes its destination (tools/vmcnt_isa_check.py; the hazard behind it is in that file's header).
    Control: with the waits deleted from the same text the checker must report violations."""
    import importlib.util
    import pytest
    spec = importlib.util.spec_from_file_location("vmcnt_isa_check", os.path.join(os.path.dirname(os.path.dirname(HDR)), "..", "tools", "vmcnt_isa_check.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    asm = chk.device_asm()
    if asm is None:
        pytest.skip("no hipcc")
    for skip in (1, 0):
        k = chk.kernel_text(asm, "_ZN10sgo_conv4r8k_conv4rILb%dELi1E" % skip)
        assert k is not None
        first, last = chk.k_loop(k)
        assert sum("v_mfma" in l for l in k[first:last + 1]) == 9 * 64          # the whole K loop body: 9 taps x 64 MFMAs
        bad, loads = chk.check(k, first, last)
        assert loads == 9 * 8 and not bad, bad[:5]
        no_waits = [l for l in k if "vmcnt" not in l]
        f2, l2 = chk.k_loop(no_waits)
        bad2, _ = chk.check(no_waits, f2, l2)
        assert bad2, "the checker did not notice loads that are never waited for"
