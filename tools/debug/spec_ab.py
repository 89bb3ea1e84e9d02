"""Speculative point update on / off on the KAIST-layout street drive (lines + wheel): stats, route counts, trajectory distance."""
import importlib, os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kaist_synth, synth_dataset as sd
import __graft_entry__ as ge
pkg = ge.load_pkg()
options, rp = (importlib.import_module("plviwo_amd." + m) for m in ("options", "replay"))
lines = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
tmp = tempfile.mkdtemp(prefix="spec_ab_")
src = os.path.join(tmp, "src"); os.makedirs(src)
sd.make_dataset(src, seconds=5.0, cam_hz=10.0, style="street", workers=min(16, os.cpu_count() or 1))
dst = kaist_synth.convert(src, os.path.join(tmp, "kaist"), sd.RL, sd.RR, sd.BASE, t0_ns=1000 * 10**9)
out = {}
for name, knobs in (("spec", 0), ("classic", 1 << 24), ("spec_nochain", 2048), ("classic_nochain", (1 << 24) | 2048)):
    pkg.debug_knobs(knobs)
    r0, c0 = pkg.route_counts(), pkg.chain_count()
    op = options.load_options(sd.write_config(os.path.join(tmp, "config"), dst, os.path.join(tmp, f"traj_{name}.txt"), use_wheel=True))
    op.est.cam.use_lines = lines
    stats, times, poses = rp.replay(op)
    out[name] = (stats, poses)
    print(name, "routes", [a - b for a, b in zip(pkg.route_counts(), r0)], "chained", pkg.chain_count() - c0,
          {k: stats[k] for k in ("cam_features", "cam_accepted", "cam_updates", "line_pool", "lines_triangulated", "lines_accepted", "line_updates", "not_psd")})
pkg.debug_knobs(0)
import oracle_context as oc
op = options.load_options(sd.write_config(os.path.join(tmp, "config"), dst, os.path.join(tmp, "traj_cpu.txt"), use_wheel=True))
op.est.cam.use_lines = lines
stats, times, poses = rp.replay(op, context_factory=oc.OracleContext, iw_initializer_factory=oc.OracleIwInitializer)
out["cpu"] = (stats, poses)
print("cpu", {k: stats[k] for k in ("cam_features", "cam_accepted", "cam_updates", "line_pool", "lines_triangulated", "lines_accepted", "line_updates", "not_psd")})
for a, b in (("spec", "cpu"), ("classic", "cpu"), ("spec", "classic"), ("spec_nochain", "classic_nochain"), ("classic", "classic_nochain")):
    d = np.abs(out[a][1][:, :3] - out[b][1][:, :3]).max(axis=1)
    first = int(np.argmax(d > 1e-6)) if (d > 1e-6).any() else -1
    print(a, "vs", b, ": max distance %.3g m, first pose above 1 um: %d of %d" % (d.max(), first, len(d)))
