set -u
OUT=gpurun_out/r6_soak; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
PLV_DEBUG_KNOBS=$((1<<28)) PLV_BENCH_STOP_AFTER_MAIN=1 timeout 900 python3 bench.py --steps 4000 --warmup 10 $A > /dev/null 2> $OUT/naps.err; echo "naps rc=$?" > $OUT/rc.txt
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 900 python3 bench.py --steps 4000 --warmup 10 $A > /dev/null 2> $OUT/plain.err; echo "plain rc=$?" >> $OUT/rc.txt
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 900 python3 bench.py --workload D --steps 1500 --warmup 10 $A > /dev/null 2> $OUT/d.err; echo "D rc=$?" >> $OUT/rc.txt
