#!/usr/bin/env python3
"""What the Python replay driver adds to a step of bench.py: wall time per call of the driver's functions around and inside
SystemManager.feed_measurement_camera, next to the library's own clock (plv_counters frame_ns + sync_ns).  Round 4, configs[2]:
~35 us around the library call (State.view 17: the IMU pose's row of the window and its two rotation matrices; the update's
argument dictionaries), ~10 inside Context.camera_frame, ~11 per Context.synchronize call — the 44 us between `ms_per_step` and
`config.ms_per_step_inside_the_library`.

    python tools/debug/python_overhead.py          (needs a GPU; runs bench.py --steps 200 without its CPU / stress / variant legs)
"""
import sys, os, time
sys.path.insert(0, os.getcwd())
sys.argv = ["bench.py", "--steps", "200", "--warmup", "20", "--no-variants", "--no-stress", "--no-cpu", "--no-pcie"]
import importlib, types
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_pkg()
system = importlib.import_module("plviwo_amd.system")
acc = {}
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            d = acc.setdefault(label, [0.0, 0]); d[0] += time.perf_counter() - t0; d[1] += 1
    setattr(obj, name, g)
wrap(system.SystemManager, "feed_measurement_camera", "feed_measurement_camera (all)")
wrap(system.SystemManager, "_try_update_args", "  _try_update_args")
wrap(system.State, "view", "  State.view")
wrap(system.State, "_window_arrays", "  State._window_arrays")
wrap(pkg.Context, "camera_frame", "  Context.camera_frame (python + C)")
wrap(pkg.Context, "_try_update_io", "    _try_update_io")
wrap(system.SystemManager, "_count_points", "  _count_points")
wrap(system.SystemManager, "_count_lines", "  _count_lines")
wrap(pkg.Context, "synchronize", "Context.synchronize")
c0 = pkg.counters()
import runpy
try:
    runpy.run_path("bench.py", run_name="__main__")
except SystemExit:
    pass
c1 = pkg.counters()
n = acc["  Context.camera_frame (python + C)"][1]
print("library inside per frame (frame_ns + sync_ns): %.1f us over %d frames" % ((c1["frame_ns"] - c0["frame_ns"] + c1["sync_ns"] - c0["sync_ns"]) / 1e3 / n, n), file=sys.stderr)
for k, (s, m) in acc.items():
    print("%-44s %8.1f us/call  %d calls" % (k, s / m * 1e6, m), file=sys.stderr)
