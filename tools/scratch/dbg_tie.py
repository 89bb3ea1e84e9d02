import sys, os, importlib, tempfile
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as ge, oracle_context as oc, synth_dataset as sd, decision_trace as dt
sd.set_camera(1280, 720)
pkg = ge.load_pkg()
options, rp = importlib.import_module("plviwo_amd.options"), importlib.import_module("plviwo_amd.replay")
d = tempfile.mkdtemp(prefix="plv_synth_")
sd.make_dataset(d, 4.0, cam_hz=20, style="avenue", workers=16)
runs = {}
for name, kw in (("hip", {}), ("cpu", dict(context_factory=oc.OracleContext, iw_initializer_factory=oc.OracleIwInitializer)),
                 ("pym", dict(context_factory=oc.PyMirrorContext, iw_initializer_factory=oc.OracleIwInitializer))):
    if name == "pym" and not os.environ.get("PYM"):
        continue
    if name != "pym" and os.environ.get("PYM") == "only":
        continue
    if name == "pym":
        import oracle_lib
        hctx = pkg.Context(pkg.default_config(1280, 720))
        orig = oracle_lib.JacOracle.triangulate_batch
        cnt = [0]
        def both(self, st, tr, **kw):
            p0, ok0, e0 = orig(self, st, tr, **kw)
            p1, ok1, e1 = hctx.triangulate(st, tr, **kw)
            g = ok0.astype(bool) & ok1.astype(bool)
            d = np.abs(p1[g] - p0[g]).max() if g.any() else 0.0
            cnt[0] += 1
            nobs = np.diff(np.ctypeslib.as_array(tr.c.obs_ptr, shape=(tr.c.n_feat + 1,)))
            if d > 1e-7 or not np.array_equal(ok0, ok1):
                bad = np.nonzero(np.abs(p1 - p0).max(axis=1) > 1e-7)[0]
                if not os.path.exists("gpurun_out/tri_case.npz"):
                    np.savez("gpurun_out/tri_case.npz", t=st.t, R=st.R, p=st.p, Rf=st.Rf, pf=st.pf, ids=st.ids, R_ItoC=np.array(list(st.c.R_ItoC)), p_IinC=np.array(list(st.c.p_IinC)),
                             K8=np.array(list(st.c.intrinsics)), cam_dt=st.c.cam_dt, dt_exp=st.c.dt_exp, ptr=tr.ptr, ot=tr.t, uv=tr.uv, uvn=tr.uvn, kw=np.array([kw.get(k, np.nan) for k in ("min_dist", "max_dist", "max_cond", "max_baseline")]),
                             p0=p0, p1=p1, ok0=ok0, ok1=ok1)
                print("triangulation call", cnt[0], "F", tr.c.n_feat, "ok equal", np.array_equal(ok0, ok1), "max|dp| %.3g" % d, "obs of the differing:", nobs[bad][:10], "obs max", nobs.max(),
                      "n_clones", st.c.n_clones, "cam_dt", st.c.cam_dt)
            return p0, ok0, e0
        oracle_lib.JacOracle.triangulate_batch = both
    op = options.load_options(sd.write_config(os.path.join(d, "config"), d, os.path.join(d, f"t_{name}.txt"), clone_freq=20, n_pts=780, max_msckf=int(os.environ.get("MAXM", "70")), calib_int=True, sigma_px=1.5))
    op.est.cam.use_lines = True
    class L(list):
        pass
    dec = L(); dec.probe_state = True; dec.states = []; dec.states_pre = []; dec.states_prop = []
    if name == "hip" and os.environ.get("MODE"):
        system = importlib.import_module("plviwo_amd.system")
        init = system.SystemManager.__init__
        def init2(self, *a, _init=init, **k):
            _init(self, *a, **k)
            if hasattr(self.ctx, "update_compression_mode"):
                self.ctx.update_compression_mode(int(os.environ["MODE"]))
        system.SystemManager.__init__ = init2
    rp.replay(op, decisions=dec, **kw)
    runs[name] = dec
thr = dt.thresholds(op)
dr = dt.value_drift(runs["hip"], runs["cpu"], thr)
np.set_printoptions(linewidth=250, precision=12)
for k, rel, nm, fid in dr[:60]:
    if rel > 1e-2 and 0:
        ra, rb = runs["hip"][k], runs["cpu"][k]
        ia, ib = list(ra[7][0]).index(fid), list(rb[7][0]).index(fid)
        print(k, rel, nm, fid); print(" hip", ra[7][1][ia]); print(" cpu", rb[7][1][ib])
print([d for d in dr if 50 <= d[0] <= 80])

n = min(len(runs["hip"]), len(runs["cpu"]))
for k in range(n):
    a, b = runs["hip"][k], runs["cpu"][k]
    da, db = a[8], b[8]
    m = max(np.abs(da).max(), np.abs(db).max(), 1e-300)
    rel = np.abs(da - db).max() / m
    if a[1] in (29, 30, 31):
        print("   ", k, a[0], "frame", a[1], "n_acc", int(a[5].sum()), int(b[5].sum()), "status", a[6], b[6], "max|dx| %.3g  diff %.3g rel %.3g" % (m, np.abs(da - db).max(), rel))
    if rel > 1e-6 or a[0] != b[0] or int(a[5].sum()) != int(b[5].sum()):
        print(k, a[0], b[0], "frame", a[1], "n_acc", int(a[5].sum()), int(b[5].sum()), "max|dx| %.3g  max|dx_h - dx_c| %.3g rel %.3g" % (m, np.abs(da - db).max(), rel))
        if k > 0:
            a, b = runs["hip"][k - 1], runs["cpu"][k - 1]
            print("   before:", a[0], "frame", a[1], "n_acc", int(a[5].sum()), int(b[5].sum()), "dx diff %.3g" % np.abs(a[8] - b[8]).max())
        break

for (fa, a, Pa), (fb, b, Pb) in zip(runs["hip"].states_pre, runs["cpu"].states_pre):
    d = np.abs(a - b)
    print("pre", fa, "%.2g@%d cov %.2g;" % (d.max(), int(np.argmax(d)), np.abs(Pa - Pb).max() / np.abs(Pb).max()), end=" ")

pa, pb = runs["hip"].states_prop, runs["cpu"].states_prop
print("propagations", len(pa), len(pb))
prev = 0
for i, (a, b) in enumerate(zip(pa, pb)):
    d = np.abs(a[2] - b[2]).max()
    if d > 1e-8:
        np.set_printoptions(precision=15, linewidth=200)
        print("propagation", i, "to t", a[0], b[0], "samples", a[1], b[1], "diff %.3g" % d, "(the one before: %.3g)" % prev)
        print(" t hip", a[3], " cpu", b[3]); print(" am hip", a[4].ravel()[:9], " cpu", b[4].ravel()[:9])
        print(" hip", a[2]); print(" cpu", b[2])
        a0, b0 = pa[i - 1], pb[i - 1]
        print(" prev t", a0[0], "hip", a0[2]); print("            cpu", b0[2])
        break
    prev = d

if "pym" in runs:
    for k in range(n):
        a, b, c = runs["hip"][k], runs["cpu"][k], runs["pym"][k]
        if a[0] != "points": continue
        m = max(np.abs(b[8]).max(), 1e-300)
        d = (np.abs(a[8] - b[8]).max() / m, np.abs(a[8] - c[8]).max() / m, np.abs(b[8] - c[8]).max() / m)
        if max(d) > 1e-6:
            print("update", k, "frame", a[1], "n_acc", int(a[5].sum()), int(b[5].sum()), int(c[5].sum()), "dx rel diff hip-cpu %.3g  hip-pym %.3g  cpu-pym %.3g" % d)
            break
print()
np.set_printoptions(precision=14, linewidth=250)
a, b = runs["hip"][60], runs["cpu"][60]
(ia, va), (ib, vb) = a[7], b[7]
i, j = list(ia).index(59), list(ib).index(59)
print("59 hip", va[i]); print("59 cpu", vb[j])
