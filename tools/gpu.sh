#!/bin/bash
# Local wrapper around gpurun: stamps the snapshot with the git head it was taken from (tools/_head.txt, read by the PMC /
# collection scripts on the GPU box, where there is no .git) and forwards everything to gpurun.
#   bash tools/gpu.sh --timeout 900 -- 'python -m pytest tests -m gpu -x -q'
cd "$(dirname "$0")/.." || exit 1
h=$(git rev-parse --short=12 HEAD 2>/dev/null || echo unknown)
git diff --quiet HEAD 2>/dev/null || h="$h+uncommitted"
echo "$h" > tools/_head.txt
exec /usr/local/graft/bin/gpurun "$@"
