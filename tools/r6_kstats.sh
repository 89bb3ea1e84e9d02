set -u
OUT=gpurun_out/${1:-r6_ks}; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
REPO=$GRAFT_REPO_ROOT
timeout 600 python -m pytest $REPO/tests -m gpu -x -q -k "${2:-ransac or tracker or frontend}" 2>&1 | tail -3 > $OUT/pytest.txt
cd /tmp && PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof -o c -- python3 $REPO/bench.py --steps 100 --warmup 5 $A > /dev/null 2> $REPO/$OUT/prof.err
