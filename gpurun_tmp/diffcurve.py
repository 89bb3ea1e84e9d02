import sys, os, importlib, tempfile
import numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import __graft_entry__ as ge, oracle_context as oc, synth_dataset as sd
pkg=ge.load_pkg()
options, rp = importlib.import_module("plviwo_amd.options"), importlib.import_module("plviwo_amd.replay")
d=tempfile.mkdtemp(prefix='dc_')
sd.make_dataset(d, 8.0, workers=16)
runs={}
for wheel in (True,):
  for name, kw in (("hip", {}), ("cpu", dict(context_factory=oc.OracleContext, iw_initializer_factory=oc.OracleIwInitializer))):
    op=options.load_options(sd.write_config(os.path.join(d,'config'), d, os.path.join(d,'o',f't{name}.txt'), use_wheel=wheel))
    op.est.cam.use_lines=False; op.sys.bag_durr=5.0
    if not wheel:
        op.est.init.imu_only_init=True
    stats,times,poses=rp.replay(op, **kw)
    runs[name]=(stats,times,poses)
  a,b=runs['hip'][2],runs['cpu'][2]
  n=min(len(a),len(b))
  print('wheel',wheel,'n',n, {k:runs['hip'][0][k] for k in ('cam_features','cam_accepted','wheel_accepted','startup_time')})
  for i in range(0,n,max(1,n//12)):
    print(i, runs['hip'][1][i], np.abs(a[i,:3]-b[i,:3]).max(), np.abs(a[i,3:]-b[i,3:]).max())
