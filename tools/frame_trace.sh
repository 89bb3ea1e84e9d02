# One frame of the kernel timeline (tools/timeline.py) and the host phase table of workload C, on a GPU box: bash tools/frame_trace.sh
set -u
REPO=$(pwd); export TMPDIR=/tmp
timeout 600 python3 bench.py --steps 5 --warmup 2 --no-cpu --no-stress --no-pcie --no-variants --stream-cache /tmp/plv_stream_C.npz > /dev/null 2>&1
cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/tl -o run -- python3 $REPO/bench.py --steps 40 --warmup 10 --no-cpu --no-stress --no-pcie --no-variants --stream-cache /tmp/plv_stream_C.npz > $REPO/gpurun_out/tl.log 2>&1
cd $REPO
f=$(find gpurun_out/tl -name "*kernel_trace.csv" | head -1)
for b in 3 4 5 6; do python3 tools/timeline.py $f hist $b > gpurun_out/timeline_$b.txt; done
PLV_DEBUG_KNOBS=16384 timeout 600 python3 bench.py --steps 100 --warmup 10 --no-cpu --no-stress --no-pcie --no-variants --stream-cache /tmp/plv_stream_C.npz > gpurun_out/ht.txt 2> gpurun_out/ht.err
rm -rf gpurun_out/tl
