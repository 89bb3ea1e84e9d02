set -u
REPO=$(pwd); OUT=gpurun_out/kt6; mkdir -p $OUT; export TMPDIR=/tmp
CACHE=/tmp/plv_stream_C.npz
timeout 600 python3 bench.py --steps 30 --warmup 5 --no-cpu --no-stress --no-pcie --no-variants --stream-cache $CACHE > $OUT/b0.json 2> $OUT/b0.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/tr -o run -- python3 $REPO/bench.py --steps 100 --warmup 10 --no-cpu --no-stress --no-pcie --no-variants --stream-cache $CACHE > $REPO/$OUT/tr.log 2>&1
cd $REPO
F=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python3 tools/frame_timeline.py $F -20 -5 0 7 20 > $OUT/timeline.txt
python3 tools/gap_from_trace.py $F > $OUT/gaps.txt
PLV_DEBUG_KNOBS=$((16384+32768)) PLV_BENCH_FRAMES=1 timeout 600 python3 bench.py --steps 100 --warmup 10 --no-cpu --no-stress --no-pcie --no-variants --stream-cache $CACHE > $OUT/bt.json 2> $OUT/bt.err
rm -rf $OUT/tr
tail -5 $OUT/gaps.txt
