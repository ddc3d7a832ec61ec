"""Live differential test: when the Python reference is mounted (/root/reference, i.e. in the build container, never
on the GPU box) fresh random playouts with NEW seeds are generated from it and replayed through the oracle.  This goes
beyond the committed goldens; it is skipped wherever the reference is absent."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.test_oracle_golden import _replay

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference not mounted")


@pytest.mark.parametrize("S,seed", [(5, 101), (9, 202), (13, 303)])
def test_fresh_reference_playouts(tmp_path, S, seed):
    out = str(tmp_path / ("live_S%d.npz" % S))
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", SGO_GOLDEN_SEED=str(seed))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "golden", "gen_golden.py"), "--child", "rules", str(S), out],
                          cwd=str(tmp_path), env=env)
    z = np.load(out, allow_pickle=False)
    assert int(z["n_games"]) >= 2
    for gi in range(int(z["n_games"])):
        _replay(z, gi, S)
