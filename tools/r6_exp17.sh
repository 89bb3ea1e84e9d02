set -u
OUT=gpurun_out/r6_e17; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 400 --warmup 10 $A --alternate-knobs 0,33554432 > $OUT/alt.txt 2> $OUT/alt.err
PLV_DEBUG_KNOBS=$((16384+32768)) PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 $A > $OUT/ht.txt 2> $OUT/ht.err
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 400 --warmup 10 $A --alternate-knobs 0,33554432 > $OUT/alt2.txt 2> $OUT/alt2.err
