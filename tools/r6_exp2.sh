# round 6, experiment 2: image hand-overs alternating frame by frame; frame timeline + device gaps; phase stamps of the fused launches
set -u
REPO=$(pwd); OUT=gpurun_out/r6_e2; mkdir -p $OUT; export TMPDIR=/tmp
CACHE=/tmp/plv_stream_C.npz
A="--no-cpu --no-stress --no-pcie --no-variants --stream-cache $CACHE"
timeout 600 python3 bench.py --steps 10 --warmup 2 $A > /dev/null 2>&1
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 300 --warmup 10 $A --alternate-images resident,pinned,host > $OUT/alt_img.txt 2> $OUT/alt_img.err
PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 300 --warmup 10 $A --alternate-images host,resident,pinned > $OUT/alt_img2.txt 2> $OUT/alt_img2.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/$OUT/tr -o run -- python3 $REPO/bench.py --steps 100 --warmup 10 $A > $REPO/$OUT/tr.log 2>&1
cd $REPO
F=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python3 tools/frame_timeline.py $F -20 -5 0 7 > $OUT/timeline.txt
python3 tools/gap_from_trace.py $F > $OUT/gaps.txt
rm -rf $OUT/tr
PLV_DEBUG_KNOBS=$((16384+32768)) PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 $A > $OUT/ht.txt 2> $OUT/ht.err
PLV_DEBUG_KNOBS=131072 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 60 --warmup 10 $A > $OUT/stamps.txt 2> $OUT/stamps.err
timeout 900 python -m pytest tests -m gpu -x -q -k "wheel or frontend or tracker" 2>&1 | tail -3 > $OUT/pytest.txt
