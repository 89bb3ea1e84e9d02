set -u
OUT=gpurun_out/r6_e36; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
timeout 900 python -m pytest tests -m gpu -x -q -k "line or component or detect" 2>&1 | tail -4 > $OUT/pytest.txt
for i in 1 2 3; do
PLV_BENCH_FRAMES=1 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 600 --warmup 10 --alternate-knobs 0,134217728 $A > /dev/null 2> $OUT/alt$i.err
done
