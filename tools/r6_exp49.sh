set -u
OUT=gpurun_out/r6_e49; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
timeout 600 python3 tools/tsqr_ab.py > $OUT/tsqr_ab.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $OUT/pytest.txt
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 200 --warmup 10 --alternate-modes 0,1 $A > /dev/null 2> $OUT/modes.err
PLV_DEBUG_KNOBS=$((1<<29)) PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 200 --warmup 10 --alternate-modes 0,1 $A > /dev/null 2> $OUT/modes_tree.err
