set -u
REPO=$(pwd); OUT=gpurun_out/r6_e14; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
PLV_DEBUG_KNOBS=$((16384+32768)) PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 $A --stream-cache /tmp/plv_stream_C.npz > $OUT/ht_c.txt 2> $OUT/ht_c.err
PLV_DEBUG_KNOBS=16384 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 $A --stream-cache /tmp/plv_stream_C.npz > $OUT/ht_c2.txt 2> $OUT/ht_c2.err
