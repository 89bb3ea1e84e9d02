set -u
OUT=gpurun_out/r6_e47; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
PLV_HELPER_NAP_US=3000 PLV_DEBUG_KNOBS=$((1<<28)) PLV_TEST_KNOBS_OR=$((1<<28)) timeout 2400 python -X faulthandler -m pytest tests -m gpu -x -q > $OUT/naps3000.txt 2>&1; echo "rc=$?" >> $OUT/naps3000.txt
PLV_HELPER_NAP_US=1500 PLV_DEBUG_KNOBS=$((1<<28)) PLV_BENCH_STOP_AFTER_MAIN=1 timeout 900 python3 bench.py --workload D --steps 1500 --warmup 10 $A > /dev/null 2> $OUT/d_naps.err; echo "D naps rc=$?" >> $OUT/naps3000.txt
PLV_HELPER_NAP_US=1500 PLV_DEBUG_KNOBS=$((1<<28)) PLV_BENCH_STOP_AFTER_MAIN=1 timeout 900 python3 bench.py --steps 3000 --warmup 10 $A > /dev/null 2> $OUT/c_naps.err; echo "C naps rc=$?" >> $OUT/naps3000.txt
