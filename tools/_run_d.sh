export LIBC_FATAL_STDERR_=1
mkdir -p gpurun_out/r4
python -X faulthandler -m pytest tests/test_gpu_replay.py -x -q -k "configs3" -s > gpurun_out/r4/t8.log 2>&1
grep -v "^  File\|^Extension" gpurun_out/r4/t8.log | tail -15
