#!/bin/bash
# A/B of library variants on ONE box (box-to-box variation is larger than most single changes): runs bench.py (GPU leg only) once per
# environment setting given as arguments ("-" = default), alternating three times, and prints mean / p50 / p99 of each run.
# usage: bash tools/ab.sh "-" "PLV_DEBUG_KNOBS=4" ...
python3 bench.py --steps 20 --warmup 5 --no-cpu --no-stress --no-pcie --no-variants --stream-cache /tmp/plv_stream_c.npz > /dev/null 2>&1   # renders once
for rep in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    r=$(env $e python3 bench.py --steps 200 --warmup 20 --no-cpu --no-stress --no-pcie --no-variants --stream-cache /tmp/plv_stream_c.npz 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['config']['latency_ms']; print('mean %.1f us  p50 %.1f  p99 %.1f  max %.1f  kernels %.1f us in %.1f launches, %.1f syncs' % (l['mean']*1e3, l['p50']*1e3, l['p99']*1e3, l['max']*1e3, d['roofline']['kernel_us_per_frame_total'], d['config']['submissions_per_frame']['kernel_launches'], d['config']['submissions_per_frame']['host_synchronisations']))")
    echo "[$v] $r"
  done
done
