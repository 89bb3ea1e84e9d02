set -u
OUT=gpurun_out/r6_e18; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
PLV_DEBUG_KNOBS=$((16384+32768)) PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 $A > $OUT/ht.txt 2> $OUT/ht.err
PLV_DEBUG_KNOBS=16384 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 $A > $OUT/ht2.txt 2> $OUT/ht2.err
for i in 1 2 3; do PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 200 --warmup 10 $A > $OUT/c$i.txt 2> $OUT/c$i.err; done
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 400 --warmup 10 $A --alternate-knobs 0,33554432 > $OUT/alt.txt 2> $OUT/alt.err
timeout 900 python -m pytest tests/test_gpu_lines.py tests/test_gpu_line_tracker.py -m gpu -x -q 2>&1 | tail -3 > $OUT/pytest.txt
