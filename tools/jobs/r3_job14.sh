#!/bin/bash
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r3/job14
mkdir -p $O
python3 -X faulthandler -m pytest tests/test_gpu_kernels.py -q -m gpu -x -k "tile_height or frame_derivatives or fused" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
