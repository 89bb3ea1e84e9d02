set -u
REPO=$(pwd); export TMPDIR=/tmp
python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.txt 2>&1; echo tests rc=$?; tail -2 gpurun_out/gpu_tests.txt
timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cpu --no-stress --no-pcie --no-variants --stream-cache /tmp/plv_stream_C.npz > /dev/null 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/kst -o run -- python3 $REPO/bench.py --steps 100 --warmup 10 --no-cpu --no-stress --no-pcie --no-variants --stream-cache /tmp/plv_stream_C.npz > $REPO/gpurun_out/kst.log 2>&1
cd $REPO
f=$(find gpurun_out/kst -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/kernel_stats_new.csv; rm -rf gpurun_out/kst
grep -E "bchol|Name" gpurun_out/kernel_stats_new.csv | cut -c1-200
