#!/bin/bash
# Collects the evidence committed under profiles/: the default bench line, the rocprofv3 kernel-trace stats of the same
# command, and the HBM-traffic counters (separate --pmc passes, MI355X_MICROARCH.md "HBM") of the two solver kernels.
set -e
R=$(pwd)   # the tree the command was started in (a staged copy under tools/gpu_stage.sh)
O=${F3D_OUT:-$R/gpurun_out}/round
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -1 $O/bench_default.json
cd /tmp && export TMPDIR=/tmp
KB=${F3D_KBENCH_KERNELS:-all}   # every solver kernel: the fused launches on frames and on frame derivatives (what the resident operator runs)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 1 --warmup 0 --no-extra > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/tools/kbench.py --size 512 --reps 3 --kernel $KB > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/tools/kbench.py --size 512 --reps 3 --kernel $KB > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -- python3 $R/tools/kbench.py --size 512 --reps 3 --kernel $KB > $O/pmc_l2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $O/pmc_sq -- python3 $R/tools/kbench.py --size 512 --reps 3 --kernel $KB > $O/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc_sq2 -- python3 $R/tools/kbench.py --size 512 --reps 3 --kernel $KB > $O/pmc_sq2.log 2>&1
ls $O
