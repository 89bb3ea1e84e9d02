set -u
REPO=$(pwd); OUT=gpurun_out/r6_e9; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
timeout 600 python3 bench.py --workload B --steps 10 --warmup 2 $A --stream-cache /tmp/plv_stream_B.npz > /dev/null 2>&1
cd /tmp
for kn in 0 16777216; do
PLV_DEBUG_KNOBS=$kn timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/st$kn -o run -- python3 $REPO/bench.py --workload B --steps 100 --warmup 10 $A --stream-cache /tmp/plv_stream_B.npz > $REPO/$OUT/st$kn.log 2>&1
find $REPO/$OUT/st$kn -name "*kernel_stats.csv" -exec cp {} $REPO/$OUT/kernel_stats_$kn.csv \;
rm -rf $REPO/$OUT/st$kn
done
cd $REPO
PLV_DEBUG_KNOBS=131072 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --workload B --steps 60 --warmup 10 $A --stream-cache /tmp/plv_stream_B.npz > $OUT/stamps_spec.txt 2> $OUT/stamps_spec.err
PLV_DEBUG_KNOBS=$((131072+16777216)) PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --workload B --steps 60 --warmup 10 $A --stream-cache /tmp/plv_stream_B.npz > $OUT/stamps_classic.txt 2> $OUT/stamps_classic.err
