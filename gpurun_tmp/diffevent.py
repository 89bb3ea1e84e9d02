import sys, os, importlib, tempfile
import numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import __graft_entry__ as ge, oracle_context as oc, synth_dataset as sd
pkg=ge.load_pkg()
options, rp = importlib.import_module("plviwo_amd.options"), importlib.import_module("plviwo_amd.replay")
d=tempfile.mkdtemp(prefix='dc_')
sd.make_dataset(d, 8.0, workers=16)
logs={}
def wrap(cls, name):
    class W(cls):
        def camera_update_points(self, st, n, *a, **kw):
            out=super().camera_update_points(st, n, *a, **kw)
            logs[name].append(('cam', kw.get('state_time'), list(map(int,out['ids'])), list(map(int,out['accepted'])), np.array(out['dx']).copy(), out['n_pool']))
            return out
        def wheel_update(self, *a):
            r=super().wheel_update(*a)
            logs[name].append(('whl', None, r[0], r[1], np.array(r[2]).copy(), 0))
            return r
    return W
for name, base, kw in (("hip", pkg.Context, {}), ("cpu", oc.OracleContext, dict(iw_initializer_factory=oc.OracleIwInitializer))):
    logs[name]=[]
    op=options.load_options(sd.write_config(os.path.join(d,'config'), d, os.path.join(d,'o',f't{name}.txt')))
    op.est.cam.use_lines=False; op.sys.bag_durr=5.0
    stats,times,poses=rp.replay(op, context_factory=wrap(base,name), **kw)
a,b=logs['hip'],logs['cpu']
print(len(a),len(b))
for i,(x,y) in enumerate(zip(a,b)):
    same = x[0]==y[0] and x[2]==y[2] and x[3]==y[3]
    dd=np.abs(x[4]-y[4]).max() if len(x[4])==len(y[4]) else -1
    if not same or dd>1e-7:
        print(i,x[0],x[1],'same',same,'dxdiff',dd, 'pool',x[5],y[5])
        if x[0]=='cam':
            print('  ids',x[2][:40]); print('  ids',y[2][:40]); print('  acc',x[3]); print('  acc',y[3])
        else:
            print('  ',x[2],x[3],y[2],y[3])
        break
