#!/bin/bash
# usage: ab_env.sh "ENV1=.. ENV2=.." ...   ("-" = none); 3 alternating repetitions
for rep in 1 2 3; do
  for v in "$@"; do
    if [ "$v" = "-" ]; then e=""; else e="$v"; fi
    r=$(env $e python3 bench.py --steps 300 --warmup 20 --no-cpu --no-stress --no-pcie --no-variants --stream-cache /tmp/plv_stream_c.npz 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); l=d['config']['latency_ms']; print('mean %.1f us  p50 %.1f  p99 %.1f' % (l['mean']*1e3, l['p50']*1e3, l['p99']*1e3))")
    echo "[$v] $r"
  done
done
