#!/bin/bash
# round 4, GPU job 35: the out-of-core tests and configs on the final tree (page-locked HostVolumes of the tests in mappings of their own)
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job35
mkdir -p $O
timeout -k 10 600 python3 -X faulthandler -m pytest tests/test_gpu_piecemeal.py tests/test_gpu_configs.py tests/test_abi.py -q -m gpu -x > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
