#!/bin/bash
# round 4, GPU job 1 (tree after the advisor fixes and the lab split): the whole GPU suite, smoke, the counter record on the
# shipped sources (tools/pmc_traffic.sh -> profiles/r04_pmc_traffic.json), the bench as the driver runs it, and -- once, as the
# round-3 advisor asked -- the command that crashed in rounds 1-2: counters WITHOUT a kernel trace over a whole bench step.
set -e
R=$(pwd)
O=${F3D_OUT:-$R/gpurun_out}/r4/job1
mkdir -p $O
timeout -k 10 1000 python3 -X faulthandler -m pytest tests -q -m gpu -x > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -30 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
F3D_OUT=$O timeout -k 10 900 bash tools/pmc_traffic.sh > $O/pmc_traffic.log 2>&1 || { tail -30 $O/pmc_traffic.log; exit 1; }
tail -3 $O/pmc_traffic.log
cp $O/traffic/pmc_traffic.json profiles/r04_pmc_traffic.json   # so that the bench of this call finds the record of its own sources
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20_warmup5.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
tail -1 $O/bench_steps20_warmup5.json | cut -c1-400
# the advisor's one run: rocprofv3 --pmc FETCH_SIZE (no --kernel-trace) over one bench step, python3 directly after "--"
cd /tmp && export TMPDIR=/tmp
export F3D_CRASH_MAPS=$O/pmc_only_crash_maps.txt
set +e
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_only -- python3 -X faulthandler $R/bench.py --steps 1 --warmup 0 --no-extra > $O/pmc_only.log 2>&1
echo "pmc-only run: exit code $?" | tee $O/pmc_only.exit
tail -2 $O/pmc_only.log | cut -c1-300
rm -rf $O/pmc_only
exit 0
