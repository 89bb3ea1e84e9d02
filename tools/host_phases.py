"""Wall time per host phase of one camera frame (workload C): the Python-level calls of the driver and, with PLV_DEBUG_KNOBS=16384, the
library's own phase timers (printed when it unloads).   usage (GPU box): PLV_DEBUG_KNOBS=16384 python tools/host_phases.py"""
import sys, os, time, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
wl=bench.WORKLOADS['C']
stream=bench.build_stream(wl, bench.PROLOGUE+225, 32)
import __graft_entry__ as ge
pkg=ge.load_pkg()
system=importlib.import_module("plviwo_amd.system")
sm=system.SystemManager(bench.load_options(wl))
pl=bench.Player(stream, sm, staged=True)
acc={}
def wrap(obj, name, label):
    f=getattr(obj,name)
    def g(*a,**k):
        t0=time.perf_counter(); r=f(*a,**k); acc[label]=acc.get(label,0)+time.perf_counter()-t0; return r
    setattr(obj,name,g)
wrap(sm.state,'view','view'); wrap(sm.state,'apply','apply')
wrap(sm.ctx,'camera_update_points','py:update_points'); wrap(sm.ctx,'camera_update_lines','py:update_lines')
wrap(sm.ctx,'tracker_feed_staged','py:tracker_feed'); wrap(sm.ctx,'line_tracker_feed_async','py:line_async'); wrap(sm.ctx,'line_tracker_feed_wait','py:line_wait')
wrap(sm.ctx,'vanishing_points','py:vps'); wrap(sm.ctx,'line_db_size','py:line_db_size')
wrap(sm,'_camera_try_update','try_update'); wrap(sm.ctx,'camera_frame','py:camera_frame'); wrap(sm,'_try_update_args','py:args'); 
for f in range(bench.PROLOGUE+20): pl.camera(*pl.next_frame())
acc.clear(); tot=0; N=200
for f in range(N):
    nf=pl.next_frame(); t0=time.perf_counter(); pl.camera(*nf); tot+=time.perf_counter()-t0
print('frame %.1f us'%(tot/N*1e6))
for k,v in sorted(acc.items(), key=lambda kv:-kv[1]): print('%-22s %.1f us/frame'%(k, v/N*1e6))
