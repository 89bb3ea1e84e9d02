set -u
OUT=gpurun_out/r6_e20; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest.txt
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 400 --warmup 10 $A --alternate-knobs 0,67108864 > $OUT/alt.txt 2> $OUT/alt.err
PLV_BENCH_FRAMES=1 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 400 --warmup 10 $A > $OUT/c.txt 2> $OUT/c.err
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 400 --warmup 10 $A --alternate-knobs 0,67108864 > $OUT/alt2.txt 2> $OUT/alt2.err
