# round 6, experiment 5: lk_ahead_kernel against lk_kernel<1> (bits + launch time), the LK parity tests, a bench run
set -u
REPO=$(pwd); OUT=gpurun_out/r6_e5; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_frontend.py tests/test_gpu_tracker.py -m gpu -x -q 2>&1 | tail -5 > $OUT/pytest_lk.txt
PLV_STREAM_CACHE=/tmp/lk_stream.npz timeout 900 python3 tools/lk_exp.py C 4,0 > $OUT/lk_exp.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $OUT/pytest.txt
CACHE=/tmp/plv_stream_C.npz
A="--no-cpu --no-stress --no-pcie --no-variants --stream-cache $CACHE"
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 300 --warmup 10 $A --alternate-knobs 0,8388608 > $OUT/alt.txt 2> $OUT/alt.err
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --workload B --steps 300 --warmup 10 --no-cpu --no-stress --no-pcie --no-variants --alternate-knobs 0,8388608 > $OUT/alt_b.txt 2> $OUT/alt_b.err
