set -u
bash tools/profile_round.sh gpurun_out/prof_c C > gpurun_out/prof_c.log 2>&1
for i in 1 2 3; do timeout 600 python3 bench.py --steps 20 --warmup 5 > gpurun_out/prof_c/driver_form_run$i.json 2> gpurun_out/prof_c/driver_form_run$i.err; done
timeout 900 python3 bench.py --workload D > gpurun_out/prof_c/bench_d_line.json 2> gpurun_out/prof_c/bench_d.err
timeout 900 python3 bench.py --workload B > gpurun_out/prof_c/bench_b_line.json 2> gpurun_out/prof_c/bench_b.err
timeout 900 python3 bench.py --workload C_avenue --no-stress > gpurun_out/prof_c/bench_c_avenue_line.json 2> gpurun_out/prof_c/bench_c_avenue.err
ls -la gpurun_out/prof_c
timeout 900 python3 tests/replay_vs_cpu.py --seconds 20 --style boulevard --mount 16,90 --cam-hz 15 --points 440 --out gpurun_out/prof_c/replay_vs_cpu_c_boulevard.json > gpurun_out/prof_c/replay_vs_cpu.log 2>&1
