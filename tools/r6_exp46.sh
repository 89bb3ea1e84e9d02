set -u
OUT=gpurun_out/r6_e46; mkdir -p $OUT; export TMPDIR=/tmp
PLV_DEBUG_KNOBS=$((1<<28)) timeout 1500 python -X faulthandler -m pytest tests -m gpu -x -q -k "replay or kaist or dropin or line or camera" > $OUT/naps_full.txt 2>&1
echo "rc=$?" >> $OUT/naps_full.txt
