"""End-to-end: a simulated visual-inertial sequence through the whole filter on the GPU and on the CPU oracle, scored with
the ATE evaluator (BASELINE.json's accuracy criterion: ATE of the accelerated path vs the CPU path within 1 cm)."""
import numpy as np
import pytest

import vio_sequence as vs

pytestmark = pytest.mark.gpu


def test_short_sequence_hip_equals_oracle(pkg, tmp_path):
    world = vs.make_world(4.0, seed=3)
    runs, stats = {}, {}
    for B in (vs.HipBackend, vs.OracleBackend):
        b = B(pkg)
        t, p, s = vs.run_filter(pkg, b, world)
        runs[b.name], stats[b.name] = (t, p), s
    assert stats["hip"] == stats["oracle"] and stats["hip"]["updates"] > 25 and stats["hip"]["features_accepted"] > 800
    ctx = pkg.Context(pkg.default_config(752, 480))
    res = vs.evaluate(pkg, ctx, str(tmp_path), runs)
    # the filter works: centimetres over ~5 m, a fraction of a degree
    for name in ("hip_vs_truth", "oracle_vs_truth"):
        assert res[name]["n"] == 41 and res[name]["pos"]["rmse"] < 0.10 and res[name]["ori"]["rmse"] < 1.0
        assert 4.0 < res[name]["length_m"] < 6.0
    # the two backends agree far inside the 1 cm budget (raw states), and to the log resolution through the files
    assert res["hip_vs_oracle_raw"]["max_pos_diff_m"] < 1e-6 and res["hip_vs_oracle_raw"]["max_quat_diff"] < 1e-7
    assert res["hip_vs_oracle_logged"]["pos"]["max"] < 2e-6
    assert abs(res["hip_vs_truth"]["pos"]["rmse"] - res["oracle_vs_truth"]["pos"]["rmse"]) < 1e-6
