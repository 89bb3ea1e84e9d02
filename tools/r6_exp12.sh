set -u
REPO=$(pwd); OUT=gpurun_out/r6_e12; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $OUT/pytest.txt
A="--no-cpu --no-stress --no-pcie --no-variants"
PLV_DEBUG_KNOBS=16384 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 $A --stream-cache /tmp/plv_stream_C.npz > $OUT/ht_c.txt 2> $OUT/ht_c.err
for i in 1 2 3; do PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 200 --warmup 10 $A --stream-cache /tmp/plv_stream_C.npz > $OUT/c$i.txt 2> $OUT/c$i.err; done
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 900 python3 bench.py --steps 300 --warmup 10 $A --stream-cache /tmp/plv_stream_C.npz --alternate-knobs 0,16777216 > $OUT/alt_C.txt 2> $OUT/alt_C.err
