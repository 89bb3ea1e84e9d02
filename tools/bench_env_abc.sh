# several values of one environment setting over alternating bench runs (measurement aid): bash tools/bench_env_abc.sh VAR rounds v1 v2 ...
VAR=$1; R=$2; shift 2
for r in $(seq $R); do for v in "$@"; do env $VAR=$v python3 bench.py --steps 150 --warmup 10 --no-cpu --no-stress --no-pcie --no-variants 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$VAR=$v', round(d['ms_per_step'],4), 'p50', round(d['latency_p50_ms'],4), 'chained', c['line_launches_chained_per_frame'])"; done; done
