set -u
OUT=gpurun_out/r6_e45; mkdir -p $OUT; export TMPDIR=/tmp
for i in 1 2 3; do timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $OUT/pytest$i.txt; done
PLV_DEBUG_KNOBS=$((1<<28)) timeout 1500 python -m pytest tests -m gpu -x -q -k "replay or kaist or dropin or line or camera" 2>&1 | tail -3 > $OUT/pytest_naps.txt
