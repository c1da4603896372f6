#!/bin/bash
# Build libsomhip.so for the current sources, then hand the command to gpurun (the built .so travels with the snapshot).
#   tools/grun.sh [--timeout S] -- '<command>'
set -e
cd "$(dirname "$0")/.."
python -m xpysom_dask_amd.build > /tmp/somhip_build.log 2>&1 || { tail -30 /tmp/somhip_build.log; exit 1; }
make -s -C oracle > /dev/null
exec /usr/local/graft/bin/gpurun "$@"
