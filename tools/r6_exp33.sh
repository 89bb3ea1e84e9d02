set -u
OUT=gpurun_out/r6_e33; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
PLV_DEBUG_KNOBS=32768 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 $A > /dev/null 2> $OUT/new.err
PLV_DEBUG_KNOBS=$((32768+67108864)) PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 $A > /dev/null 2> $OUT/old.err
PLV_DEBUG_KNOBS=32768 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 $A > /dev/null 2> $OUT/new2.err
PLV_DEBUG_KNOBS=$((32768+67108864)) PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 $A > /dev/null 2> $OUT/old2.err
