#!/bin/bash
# Clock / power while the C3 predictive sweep keeps the GPU busy (is the fp64-MFMA rate power- or clock-capped?)
cd "${GRAFT_REPO_ROOT:-.}"
python bench.py --no-cpu-baseline --steps 300 --warmup 3 > /tmp/bench_power.json 2>/dev/null &
PID=$!
sleep 5
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|Sensor junction" | sed 's/GPU\[0\]\s*: //' | tr '\n' ';'; echo
  sleep 1
done
wait $PID
python -c "import json; d=json.loads(open('/tmp/bench_power.json').read().strip().splitlines()[-1]); print('bench value', d['value'], 'pipeline TF', d['pipeline_tflops'])"
/opt/rocm/bin/rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -3
