set -u
OUT=gpurun_out/r6_e26; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > $OUT/pytest.txt
for i in 1 2 3; do PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 600 --warmup 10 $A --alternate-knobs 0,67108864 > $OUT/alt$i.txt 2> $OUT/alt$i.err; done
