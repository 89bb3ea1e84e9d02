# round 6, experiment 1: parity after the copy-in histogram + scratch fix, the three image hand-overs, WRITE_SIZE of the fused launches
set -u
REPO=$(pwd); OUT=gpurun_out/r6_e1; mkdir -p $OUT; export TMPDIR=/tmp
CACHE=/tmp/plv_stream_C.npz
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $OUT/pytest.txt
timeout 600 python3 bench.py --steps 20 --warmup 5 --stream-cache $CACHE > $OUT/driver1.json 2> $OUT/driver1.err
timeout 600 python3 bench.py --steps 20 --warmup 5 --stream-cache $CACHE > $OUT/driver2.json 2> $OUT/driver2.err
A="--steps 100 --warmup 10 --no-cpu --no-stress --no-pcie --no-variants --stream-cache $CACHE"
PLV_DEBUG_KNOBS=16384 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py $A --images host > $OUT/ht_host.txt 2> $OUT/ht_host.err
PLV_DEBUG_KNOBS=16384 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py $A --images pinned > $OUT/ht_pinned.txt 2> $OUT/ht_pinned.err
PLV_DEBUG_KNOBS=16384 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py $A --images resident > $OUT/ht_res.txt 2> $OUT/ht_res.err
cd /tmp
A="--steps 60 --warmup 10 --no-cpu --no-stress --no-pcie --no-variants --stream-cache $CACHE"
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$REPO/$OUT/pmc_write" -o run -- python3 "$REPO/bench.py" $A > "$REPO/$OUT/pmc_write.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$REPO/$OUT/pmc_fetch" -o run -- python3 "$REPO/bench.py" $A > "$REPO/$OUT/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/stats" -o run -- python3 "$REPO/bench.py" $A --images pinned > "$REPO/$OUT/stats.log" 2>&1
cd $REPO
python3 tools/pmc_summary.py "$OUT/pmc_hbm.csv" FETCH_SIZE="$OUT/pmc_fetch" WRITE_SIZE="$OUT/pmc_write" > /dev/null
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats_pinned.csv" \;
rm -rf $OUT/pmc_write $OUT/pmc_fetch $OUT/stats
