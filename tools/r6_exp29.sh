set -u
OUT=gpurun_out/r6_e29; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
PLV_DEBUG_KNOBS=32768 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 60 --warmup 10 $A > $OUT/lt.txt 2> $OUT/lt.err
