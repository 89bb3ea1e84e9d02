set -u
OUT=gpurun_out/r6_e43; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $OUT/pytest.txt
