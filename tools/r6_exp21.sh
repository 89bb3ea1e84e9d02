set -u
REPO=$(pwd); OUT=gpurun_out/r6_e24; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > $OUT/pytest.txt
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 400 --warmup 10 $A --alternate-knobs 0,67108864 > $OUT/alt.txt 2> $OUT/alt.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/kst -o run -- python3 $REPO/bench.py --steps 100 --warmup 10 $A > $REPO/$OUT/kst.log 2>&1
cd $REPO
f=$(find $OUT/kst -name "*kernel_stats.csv" | head -1); cp $f $OUT/kernel_stats.csv; rm -rf $OUT/kst
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 400 --warmup 10 $A --alternate-knobs 0,67108864 > $OUT/alt2.txt 2> $OUT/alt2.err
