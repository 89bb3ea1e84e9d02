for r in 1 2; do
for cfg in "ccx 0" "none 0" "ccx 2097152"; do set -- $cfg
PLV_DEBUG_KNOBS=$2 python3 bench.py --workload B --pin $1 --steps 150 --warmup 10 --no-cpu --no-stress --no-pcie --no-variants 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('B pin=$1 knobs=$2', round(d['ms_per_step'],4), 'p50', round(d['latency_p50_ms'],4), c['kernels_us_per_frame'].get('lk_kernel') if 'kernels_us_per_frame' in c else '')"
done; done
