#!/bin/bash
# Board power and clocks while one solver kernel runs back to back (tools/kbench.py in the background, rocm-smi sampled beside it).
#   tools/power_probe.sh <kernel> [env assignments...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
K=${1:-sweep2}; shift
env "$@" python3 $R/tools/kbench.py --size 512 --reps 1500 --kernel $K > /tmp/kb_$K.log 2>&1 &
pid=$!
sleep 12
for i in 1 2 3; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|memory)" | tr -s ' ' | paste -s -d';'
  sleep 1
done
wait $pid
cat /tmp/kb_$K.log
