"""__graft_entry__.smoke(): one small invocation of the hot path on cuda:0, checked against the oracle."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import oracle_lib  # noqa: E402
import synth  # noqa: E402


def run():
    import __graft_entry__ as ge
    pkg = ge.load_pkg()
    orc = oracle_lib.load()
    ctx = pkg.Context()
    n, k = 113, 98
    P = synth.spd_cov(n)
    cols = synth.col_map(n, k)
    rows, Hf, Hx, res = synth.msckf_batch(F=20, M=15, k=k, seed=1)
    rc0, P0, dx0, acc0, _ = orc.msckf_update(P, rows, Hf, Hx, res, cols, 2.25, synth.q95_table())
    rc1, P1, dx1, acc1, _ = ctx.msckf_update(P, rows, Hf, Hx, res, cols, 2.25)
    assert rc0 == 0 and rc1 == 0, (rc0, rc1)
    assert np.array_equal(acc0, acc1)
    err = np.max(np.abs(P1 - P0)) / np.max(np.abs(P0))
    assert err < 1e-8, err
    assert np.max(np.abs(dx1 - dx0)) / np.max(np.abs(dx0)) < 1e-8
    print(f"smoke: msckf update parity ok (rel err P {err:.2e}, accepted {int(acc1.sum())}/{len(rows)})")
    if hasattr(sys.modules.get("smoke_frontend"), "run"):
        pass
    try:
        import smoke_frontend
    except ImportError:
        smoke_frontend = None
    if smoke_frontend is not None:
        smoke_frontend.run(pkg, ctx)
    ctx.close()
