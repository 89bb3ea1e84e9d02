set -u
OUT=gpurun_out/r6_e30; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
timeout 900 python -m pytest tests -m gpu -x -q -k "line or component or detect" 2>&1 | tail -4 > $OUT/pytest.txt
for i in 1 2 3; do
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 300 --warmup 10 $A > /dev/null 2> $OUT/new$i.err
PLV_DEBUG_KNOBS=67108864 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 300 --warmup 10 $A > /dev/null 2> $OUT/old$i.err
done
PLV_DEBUG_KNOBS=32768 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 40 --warmup 10 $A > /dev/null 2> $OUT/lt.err
