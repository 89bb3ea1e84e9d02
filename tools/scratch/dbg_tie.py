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
for name, kw in (("hip", {}), ("cpu", dict(context_factory=oc.OracleContext, iw_initializer_factory=oc.OracleIwInitializer))):
    op = options.load_options(sd.write_config(os.path.join(d, "config"), d, os.path.join(d, f"t_{name}.txt"), clone_freq=20, n_pts=780, max_msckf=70, calib_int=True, sigma_px=1.5))
    op.est.cam.use_lines = True
    dec = []
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
print(dr[:12])

for k in range(12):
    a, b = runs["hip"][k], runs["cpu"][k]
    da, db = a[8], b[8]
    m = max(np.abs(da).max(), np.abs(db).max(), 1e-300)
    i = np.argmax(np.abs(da - db))
    print(k, a[0], "n_acc", int(a[5].sum()), int(b[5].sum()), "max|dx| %.3g  max|dx_h - dx_c| %.3g at %d (|dx| there %.3g)  rel %.3g" % (m, np.abs(da - db).max(), i, abs(db[i]), np.abs(da - db).max() / m))
