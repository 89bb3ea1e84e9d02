#!/usr/bin/env python3
"""Where the host time of a camera step goes: cProfile of SystemManager.feed_measurement_camera over the bench stream (needs a GPU).
    python tools/frame_profile.py [--frames 100] [--workload C]"""
import argparse
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--workload", default="C")
    a = ap.parse_args()
    wl = bench.WORKLOADS[a.workload]
    stream = bench.build_stream(wl, bench.PROLOGUE + a.frames + 5, min(32, os.cpu_count() or 1))
    import importlib
    import __graft_entry__ as ge
    ge.load_pkg()
    system = importlib.import_module("plviwo_amd.system")
    sm = system.SystemManager(bench.load_options(wl))
    pl = bench.Player(stream, sm, staged=True)
    for f in range(bench.PROLOGUE):
        pl.camera(*pl.next_frame())
    pr = cProfile.Profile()
    for f in range(a.frames):
        nf = pl.next_frame()
        pr.enable()
        pl.camera(*nf)
        pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
