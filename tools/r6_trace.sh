set -u
OUT=gpurun_out/r6_trace; mkdir -p $OUT; export TMPDIR=/tmp
REPO=$GRAFT_REPO_ROOT
A="--steps 200 --warmup 10 --no-cpu --no-stress --no-pcie --no-variants"
cd /tmp && PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/kt -o c -- python3 $REPO/bench.py $A > /dev/null 2> $REPO/$OUT/kt.err
cd $REPO && python3 tools/frame_timeline_median.py $(find $OUT/kt -name "*kernel_trace.csv" | head -1) 80 > $OUT/frame_timeline.txt 2>&1
rm -rf $OUT/kt
