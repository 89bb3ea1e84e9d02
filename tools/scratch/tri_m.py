import sys, os
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import __graft_entry__ as ge, oracle_lib, synth
pkg = ge.load_pkg()
ctx = pkg.Context(pkg.default_config(752, 480))
jo, fo = oracle_lib.load_jac(pkg), oracle_lib.load_front()
for M, nc, off in ((21, 21, 0.0), (21, 21, 0.004), (21, 21, -0.004), (20, 21, -0.004), (21, 22, -0.004), (11, 11, -0.004), (11, 11, 0.004)):
    sc = synth.vio_scene(n_clones=nc, F=60, M=M, noise_px=0.3, obs_offset=off)
    uvn = fo.undistort(sc["K8"], sc["obs_uv"])
    st = pkg.StateView(sc["t"], sc["R"], sc["p"], sc["ids"], sc["R_ItoC"], sc["p_IinC"], sc["K8"], intrinsic_state_id=15)
    tr = pkg.Tracks(sc["obs_ptr"], sc["obs_time"], sc["obs_uv"], np.zeros((60, 3)), obs_uvn=uvn)
    opt = dict(max_cond=1e7, max_dist=150.0, max_baseline=2000.0)
    p0, ok0, e0 = jo.triangulate_batch(st, tr, **opt)
    p1, ok1, e1 = ctx.triangulate(st, tr, **opt)
    good = ok0.astype(bool) & ok1.astype(bool)
    print("M", M, "clones", nc, "offset", off, "ok equal", np.array_equal(ok0, ok1), "n good", good.sum(), "max |dp| %.3g" % (np.abs(p1[good] - p0[good]).max() if good.any() else -1),
          "obs/feature", np.diff(sc["obs_ptr"])[:3], "t0 obs - t0 clone %.4g" % (sc["obs_time"][0] - sc["t"][0]))
