"""CPU-only: the C-ABI library builds for gfx950, loads, and exports every symbol include/sgo.h declares.
No compute calls (there is no GPU in the CPU test tier)."""
import ctypes
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from sejonggo_amd.build import build_lib
    from sejonggo_amd import _lib as L
    build_lib()
    return L


def test_library_exports_every_declared_symbol():
    L = _lib()
    lib = L.load()
    hdr = open(os.path.join(ROOT, "include", "sgo.h")).read()
    declared = set(re.findall(r"\b(sgo_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"sgo_config", "sgo_status", "sgo_move_record", "sgo_game_result", "sgo_ctx"}
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    for s in sorted(declared):
        assert hasattr(lib, s), s


def test_geometry_and_errors():
    L = _lib()
    lib = L.load()
    assert lib.sgo_version() >= 1
    assert [lib.sgo_plane_words(s) for s in L.SUPPORTED_SIZES] == [1, 2, 3, 6, 12]
    assert [lib.sgo_packed_words(s) for s in L.SUPPORTED_SIZES] == [16, 32, 48, 96, 192]
    assert lib.sgo_apad(19) == 384 and lib.sgo_apad(9) == 96
    assert lib.sgo_plane_words(8) < 0  # unsupported size is an error, not a fallback


def test_struct_layouts_match_header(tmp_path):
    """ctypes structs / numpy dtypes of _lib.py against the C compiler's view of include/sgo.h: sizes and every field offset."""
    import os
    import subprocess
    L = _lib()
    assert ctypes.sizeof(L.Config) == 64
    assert ctypes.sizeof(L.Status) == 48
    assert ctypes.sizeof(L.MoveRecord) == 24 == L.MOVE_RECORD_DTYPE.itemsize
    assert ctypes.sizeof(L.GameResult) == 40 == L.GAME_RESULT_DTYPE.itemsize
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    structs = {"sgo_config": L.Config, "sgo_status": L.Status, "sgo_move_record": L.MoveRecord, "sgo_game_result": L.GameResult}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "sgo.h"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for fname, _ in cls._fields_:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, fname, cname, fname))
    lines += ['return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = str(tmp_path / "layout")
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", exe])
    got = dict(l.split() for l in subprocess.check_output([exe], text=True).splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got["%s.%s" % (cname, fname)]) == getattr(cls, fname).offset, (cname, fname)
    for name in L.GAME_RESULT_DTYPE.names:
        assert L.GAME_RESULT_DTYPE.fields[name][1] == getattr(L.GameResult, name).offset
    for name in L.MOVE_RECORD_DTYPE.names:
        assert L.MOVE_RECORD_DTYPE.fields[name][1] == getattr(L.MoveRecord, name).offset


def test_sym_lut_is_host_side_and_matches_golden():
    """sgo_sym_lut is pure host arithmetic (symmetry.py:12-42), so it can be pinned without a GPU."""
    from sejonggo_amd import symmetry
    from tests.helpers import load
    for S in (5, 9, 19):
        z = load("sym_S%d.npz" % S)
        for k in range(8):
            assert np.array_equal(symmetry.sym_lut(S, k), z["luts"][k])


def test_no_cpu_fallback_without_gpu():
    """On a box without a HIP device the product entry points must raise, not compute on the CPU."""
    L = _lib()
    lib = L.load()
    if lib.sgo_device_count() > 0:
        return
    from sejonggo_amd import play
    import pytest
    with pytest.raises(L.SgoError):
        play.game_init(9)
    with pytest.raises(L.SgoError):
        play.legal_moves(np.zeros((1, 9, 9, 17), dtype=np.int32))


def test_header_is_plain_c(tmp_path):
    """include/sgo.h is the drop-in boundary: it must compile as C99 (and as C++) on its own, no HIP or torch types."""
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "sgo.h"\nint main(void) { return 0; }\n')
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", inc, str(src)])
    subprocess.check_call(["g++", "-std=c++11", "-fsyntax-only", "-I", inc, "-x", "c++", str(src)])
    hdr = open(os.path.join(inc, "sgo.h")).read()
    assert "#include <hip" not in hdr and "#include <torch" not in hdr and "#include <ATen" not in hdr
