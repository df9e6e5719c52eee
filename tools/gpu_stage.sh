#!/bin/bash
# Runs a command on the GPU box from a FROZEN copy of the tree (stage/<name>/, git-ignored): gpurun snapshots /root/repo only when
# it has got a box -- minutes after the call -- so a job started from the live tree sees whatever was being edited or rebuilt
# at that moment.  The copy is made now, the job runs inside it, results go to the real gpurun_out/ as usual.
#   tools/gpu_stage.sh <name> [--timeout S] -- <command run from the root of the copy>
# Inside the command: $PWD = the copy, $F3D_OUT = <repo>/gpurun_out (merged back by gpurun).
set -e
name=$1; shift
timeout=1200
if [ "$1" = "--timeout" ]; then timeout=$2; shift 2; fi
[ "$1" = "--" ] && shift
root=$(cd "$(dirname "$0")/.." && pwd)
rm -rf "$root/stage"            # one stage at a time: older copies would travel too
mkdir -p "$root/stage/$name"
tar -C "$root" --exclude=./.git --exclude=./gpurun_out --exclude=./stage --exclude=.pytest_cache --exclude=__pycache__ \
    --exclude=./cuda-flow3d_amd/build --exclude=./tests/cpu_device/_build -cf - . | tar -C "$root/stage/$name" -xf -
exec /usr/local/graft/bin/gpurun --timeout $timeout -- "cd stage/$name && export F3D_OUT=\$GRAFT_REPO_ROOT/gpurun_out && $*"
