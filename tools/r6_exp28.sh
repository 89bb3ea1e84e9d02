set -u
OUT=gpurun_out/r6_e28; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
PLV_DEBUG_KNOBS=131072 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 $A > $OUT/stamps.txt 2> $OUT/stamps.err
for i in 1 2; do PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 300 --warmup 10 $A > $OUT/c$i.txt 2> $OUT/c$i.err; done
