#!/bin/bash
# Device-side span of the update chain and of the front-end, from a rocprofv3 kernel trace of the sequential schedule.
# usage (GPU box, repo root): bash tools/chain_span.sh
REPO=$(pwd)
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/plv_trace
rocprofv3 --kernel-trace --output-format csv -d /tmp/plv_trace -o run -- python3 "$REPO/bench.py" --steps 120 --warmup 10 --no-cpu --sequential > /dev/null 2>&1
cd "$REPO"
python3 tools/frame_timeline.py /tmp/plv_trace avg
