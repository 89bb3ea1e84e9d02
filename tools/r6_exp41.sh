set -u
OUT=gpurun_out/r6_e41; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
timeout 900 python -m pytest tests -m gpu -x -q -k "line or component or detect" 2>&1 | tail -3 > $OUT/pytest.txt
PLV_DEBUG_KNOBS=16384 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 300 --warmup 10 $A > /dev/null 2> $OUT/ht.err
for i in 1 2; do PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 300 --warmup 10 $A > /dev/null 2> $OUT/c$i.err; done
