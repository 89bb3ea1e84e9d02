set -u
REPO=$(pwd); OUT=gpurun_out/r6_e10; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
PLV_DEBUG_KNOBS=16384 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 $A --stream-cache /tmp/plv_stream_C.npz > $OUT/ht_c.txt 2> $OUT/ht_c.err
timeout 600 python3 bench.py --steps 10 --warmup 2 $A --stream-cache /tmp/plv_stream_C.npz > /dev/null 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/tr -o run -- python3 $REPO/bench.py --steps 60 --warmup 10 $A --stream-cache /tmp/plv_stream_C.npz > $REPO/$OUT/tr.log 2>&1
cd $REPO
F=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python3 tools/frame_timeline.py $F -10 0 5 > $OUT/timeline_c.txt
rm -rf $OUT/tr
timeout 600 python3 bench.py --steps 20 --warmup 5 --stream-cache /tmp/plv_stream_C.npz > $OUT/driver1.json 2> $OUT/driver1.err
