set -u
OUT=gpurun_out/r6_e15; mkdir -p $OUT; export TMPDIR=/tmp
A="--no-cpu --no-stress --no-pcie --no-variants"
PLV_DEBUG_KNOBS=16384 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --steps 200 --warmup 10 $A > $OUT/c.txt 2> $OUT/c.err
PLV_DEBUG_KNOBS=16384 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --workload B --steps 200 --warmup 10 $A > $OUT/b.txt 2> $OUT/b.err
PLV_DEBUG_KNOBS=16384 PLV_BENCH_STOP_AFTER_MAIN=1 timeout 600 python3 bench.py --workload C_avenue --steps 200 --warmup 10 $A > $OUT/ca.txt 2> $OUT/ca.err
grep -h "plv over" $OUT/c.err | sort | uniq -c | sort -k3n > $OUT/over_c.txt
grep -h "plv over" $OUT/b.err | sort | uniq -c | sort -k3n > $OUT/over_b.txt
grep -h "plv over" $OUT/ca.err | sort | uniq -c | sort -k3n > $OUT/over_ca.txt
