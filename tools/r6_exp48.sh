set -u
OUT=gpurun_out/r6_e48; mkdir -p $OUT; export TMPDIR=/tmp
run() { name=$1; shift; env "$@" timeout 2400 python -X faulthandler -m pytest tests -m gpu -x -q > $OUT/$name.txt 2>&1; echo "rc=$?" >> $OUT/$name.txt; }
run spin0_naps PLV_LINE_SPIN_US=0 PLV_DEBUG_KNOBS=$((1<<28)) PLV_TEST_KNOBS_OR=$((1<<28))
run spin0 PLV_LINE_SPIN_US=0
run fit2_naps PLV_LINE_FIT_THREADS=2 PLV_DEBUG_KNOBS=$((1<<28)) PLV_TEST_KNOBS_OR=$((1<<28))
run fit0 PLV_LINE_FIT_THREADS=0
