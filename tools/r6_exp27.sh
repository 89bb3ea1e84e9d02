set -u
OUT=gpurun_out/r6_e27; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $OUT/pytest.txt
PLV_DEBUG_KNOBS=16384 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 200 --warmup 10 $A > $OUT/ht.txt 2> $OUT/ht.err
for i in 1 2 3; do PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 300 --warmup 10 $A > $OUT/c$i.txt 2> $OUT/c$i.err; done
