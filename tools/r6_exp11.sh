set -u
REPO=$(pwd); OUT=gpurun_out/r6_e11; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_linefront.py tests/test_gpu_lines.py tests/test_gpu_replay.py -m gpu -x -q 2>&1 | tail -4 > $OUT/pytest_lines.txt
A="--no-cpu --no-stress --no-pcie --no-variants"
PLV_DEBUG_KNOBS=16384 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 $A --stream-cache /tmp/plv_stream_C.npz > $OUT/ht_c.txt 2> $OUT/ht_c.err
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 200 --warmup 10 $A --stream-cache /tmp/plv_stream_C.npz > $OUT/c.txt 2> $OUT/c.err
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --workload D --steps 200 --warmup 10 $A --stream-cache /tmp/plv_stream_D.npz > $OUT/d.txt 2> $OUT/d.err
