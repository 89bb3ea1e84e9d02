# round 6, experiment 7: the point update enqueued behind the flow (default) against knob 1 << 24 (after the flow's result), workloads C, B, D
set -u
REPO=$(pwd); OUT=gpurun_out/r6_e7; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $OUT/pytest.txt
A="--no-cpu --no-stress --no-pcie --no-variants"
for wl in C B D; do
  PLV_BENCH_STOP_AFTER_MAIN=1 timeout 900 python3 bench.py --workload $wl --steps 300 --warmup 10 $A --stream-cache /tmp/plv_stream_$wl.npz --alternate-knobs 0,16777216 > $OUT/alt_$wl.txt 2> $OUT/alt_$wl.err
done
PLV_DEBUG_KNOBS=16384 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 $A --stream-cache /tmp/plv_stream_C.npz > $OUT/ht_c.txt 2> $OUT/ht_c.err
PLV_DEBUG_KNOBS=16384 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --workload B --steps 100 --warmup 10 $A --stream-cache /tmp/plv_stream_B.npz > $OUT/ht_b.txt 2> $OUT/ht_b.err
