set -u
OUT=gpurun_out/r6_e50; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
REPO=$GRAFT_REPO_ROOT
cd /tmp && PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/$OUT/prof_m1 -o m1 -- python3 $REPO/bench.py --steps 60 --warmup 5 --alternate-modes 1 $A > /dev/null 2> $REPO/$OUT/prof.err
cd $REPO; PLV_DEBUG_KNOBS=16384 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 100 --warmup 10 --alternate-modes 1 $A > /dev/null 2> $OUT/ht.err
