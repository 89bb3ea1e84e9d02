#!/bin/bash
# The whole GPU suite with the line detector's helper threads napping at random (knob 1 << 28 from the start and kept set under every
# test's own knob mask), then with the rounds-5 protocol (1 << 27: every helper's report is waited for).  On the GPU box, repo root.
set -u
OUT=${1:-gpurun_out/suite_naps}; mkdir -p $OUT; export TMPDIR=/tmp
PLV_DEBUG_KNOBS=$((1<<28)) PLV_TEST_KNOBS_OR=$((1<<28)) timeout 2400 python -X faulthandler -m pytest tests -m gpu -x -q > $OUT/naps.txt 2>&1; echo "rc=$?" >> $OUT/naps.txt
PLV_DEBUG_KNOBS=$((1<<27)) PLV_TEST_KNOBS_OR=$((1<<27)) timeout 2400 python -X faulthandler -m pytest tests -m gpu -x -q > $OUT/wait_all.txt 2>&1; echo "rc=$?" >> $OUT/wait_all.txt
