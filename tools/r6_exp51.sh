set -u
OUT=gpurun_out/r6_e51; mkdir -p $OUT; export TMPDIR=/tmp
PLV_DEBUG_KNOBS=$((1<<29)) PLV_TEST_KNOBS_OR=$((1<<29)) timeout 2400 python -X faulthandler -m pytest tests -m gpu -x -q > $OUT/tree.txt 2>&1; echo "rc=$?" >> $OUT/tree.txt
PLV_DEBUG_KNOBS=$((1<<28)) PLV_TEST_KNOBS_OR=$((1<<28)) timeout 2400 python -X faulthandler -m pytest tests -m gpu -x -q > $OUT/naps.txt 2>&1; echo "rc=$?" >> $OUT/naps.txt
