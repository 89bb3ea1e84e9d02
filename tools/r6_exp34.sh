set -u
OUT=gpurun_out/r6_e34; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
for i in 1 2; do
PLV_BENCH_FRAMES=1 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 600 --warmup 10 --alternate-knobs ${KNOBS:-0,67108864} $A > /dev/null 2> $OUT/alt$i.err
done
