#!/bin/bash
# Reproduces the files under profiles/rNN from a GPU box: bench line, rocprofv3 kernel stats, marker (roctx) summary, PMC passes.
# The first (unprofiled) bench run renders the stream and saves it; every profiled run loads it (--stream-cache): nothing forks in a
# process the profiler's library has already initialised the GPU in.  Every pass runs under its own timeout.
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh gpurun_out/prof [C|B|D]
set -u
OUT=${1:-gpurun_out/prof}
WL=${2:-C}
REPO=$(pwd)
CACHE=/tmp/plv_stream_$WL.npz
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python3 bench.py --workload $WL --stream-cache $CACHE > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.log"
A="--workload $WL --steps 100 --warmup 10 --no-cpu --no-stress --no-pcie --no-variants --stream-cache $CACHE"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/stats" -o run -- python3 "$REPO/bench.py" $A > "$REPO/$OUT/stats.log" 2>&1
export PLV_ROCTX=1
timeout 600 rocprofv3 --marker-trace --stats --output-format csv -d "$REPO/$OUT/marker" -o run -- python3 "$REPO/bench.py" $A > "$REPO/$OUT/marker.log" 2>&1
unset PLV_ROCTX
A="--workload $WL --steps 60 --warmup 10 --no-cpu --no-stress --no-pcie --no-variants --stream-cache $CACHE"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$REPO/$OUT/pmc_fetch" -o run -- python3 "$REPO/bench.py" $A > "$REPO/$OUT/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$REPO/$OUT/pmc_write" -o run -- python3 "$REPO/bench.py" $A > "$REPO/$OUT/pmc_write.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$REPO/$OUT/pmc_sq" -o run -- python3 "$REPO/bench.py" $A > "$REPO/$OUT/pmc_sq.log" 2>&1
cd "$REPO"
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
find "$OUT/marker" -name "*marker_api_stats.csv" -exec cp {} "$OUT/marker_stats.csv" \; 2>/dev/null
find "$OUT/marker" -name "*marker*stats*.csv" -exec cp {} "$OUT/marker_stats.csv" \; 2>/dev/null
python3 tools/pmc_summary.py "$OUT/pmc_hbm.csv" FETCH_SIZE="$OUT/pmc_fetch" WRITE_SIZE="$OUT/pmc_write" > /dev/null
python3 tools/pmc_summary.py "$OUT/pmc_sq.csv" SQ_INSTS_VALU_MFMA_MOPS_F64+SQ_VALU_MFMA_BUSY_CYCLES+SQ_LDS_BANK_CONFLICT+SQ_BUSY_CYCLES="$OUT/pmc_sq" > /dev/null
ls "$OUT/marker" > "$OUT/marker_files.txt" 2>/dev/null; find "$OUT/marker" -type f | head -20 >> "$OUT/marker_files.txt"
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq" "$OUT/marker"
tail -3 "$OUT"/*.log > "$OUT/logs_tail.txt" 2>/dev/null
rm -f "$OUT"/*.log
ls -la "$OUT"
