# round 6, experiment 3: helper threads of the line detector's host stage over one and two L3 complexes
set -u
REPO=$(pwd); OUT=gpurun_out/r6_e3; mkdir -p $OUT; export TMPDIR=/tmp
CACHE=/tmp/plv_stream_C.npz
A="--no-cpu --no-stress --no-pcie --no-variants --stream-cache $CACHE"
timeout 600 python3 bench.py --steps 10 --warmup 2 $A > /dev/null 2>&1
for pin in ccx ccx2; do
  PLV_LINE_FIT_THREADS=15 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 320 --warmup 10 $A --pin $pin --alternate-fit 7,11,13,15 > $OUT/fit_$pin.txt 2> $OUT/fit_$pin.err
  PLV_DEBUG_KNOBS=16384 PLV_LINE_FIT_THREADS=13 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 $A --pin $pin > $OUT/ht13_$pin.txt 2> $OUT/ht13_$pin.err
done
PLV_DEBUG_KNOBS=16384 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 $A --pin ccx > $OUT/ht7_ccx.txt 2> $OUT/ht7_ccx.err
