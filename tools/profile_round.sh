#!/bin/bash
# Reproduces the files under profiles/rNN from a GPU box: bench line, rocprofv3 kernel stats, two PMC passes.
# Every pass runs under its own timeout (a counter pass that stalls must not eat the box's time limit).
# usage (on the GPU box, from the repo root):  bash tools/profile_round.sh gpurun_out/prof [C|B|D]
set -u
OUT=${1:-gpurun_out/prof}
WL=${2:-C}
REPO=$(pwd)
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python3 bench.py --workload $WL > "$OUT/bench_line.json" 2> "$OUT/bench_stderr.log"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$REPO/$OUT/stats" -o run -- python3 "$REPO/bench.py" --workload $WL --steps 200 --warmup 20 --no-cpu > "$REPO/$OUT/stats.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$REPO/$OUT/pmc_fetch" -o run -- python3 "$REPO/bench.py" --workload $WL --steps 60 --warmup 10 --no-cpu > "$REPO/$OUT/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$REPO/$OUT/pmc_write" -o run -- python3 "$REPO/bench.py" --workload $WL --steps 60 --warmup 10 --no-cpu > "$REPO/$OUT/pmc_write.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$REPO/$OUT/pmc_sq" -o run -- python3 "$REPO/bench.py" --workload $WL --steps 60 --warmup 10 --no-cpu > "$REPO/$OUT/pmc_sq.log" 2>&1
cd "$REPO"
find "$OUT/stats" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
python3 tools/pmc_summary.py "$OUT/pmc_hbm.csv" FETCH_SIZE="$OUT/pmc_fetch" WRITE_SIZE="$OUT/pmc_write" > /dev/null
python3 tools/pmc_summary.py "$OUT/pmc_sq.csv" SQ_INSTS_VALU_MFMA_MOPS_F64+SQ_VALU_MFMA_BUSY_CYCLES+SQ_LDS_BANK_CONFLICT+SQ_BUSY_CYCLES="$OUT/pmc_sq" > /dev/null
rm -rf "$OUT/stats" "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_sq"
rm -f "$OUT"/*.log
ls -la "$OUT"
