#!/bin/bash
# Build everything in-tree (the .so files travel with the snapshot), then run a command on the GPU box:
#   tools/gpu.sh [--timeout S] '<command>'
cd "$(dirname "$0")/.." || exit 1
T=900
if [ "$1" = "--timeout" ]; then T=$2; shift 2; fi
python -c "import __graft_entry__ as g; g.build()" 2>&1 | grep -v "^+" | grep -v "hip-link"
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
