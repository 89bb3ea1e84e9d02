# four modes of the bench's CPU affinity, alternating, three rounds (measurement aid)
for r in 1 2 3; do for pin in none node ccx; do python3 bench.py --pin $pin --steps 100 --warmup 10 --no-cpu --no-stress --no-pcie --no-variants 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$pin', round(d['ms_per_step'],4), round(d['latency_p50_ms'],4), c['us_per_step_in_plv_ctx_synchronize'], c['line_launches_chained_per_frame'], c['cpu_affinity'])"; done; done
