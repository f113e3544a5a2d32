cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fit_small.py -x -q -m gpu > $O/tests_j.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error|assert" $O/tests_j.log | tail -5
rm -rf /tmp/fs; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/fs -- python tools/fit_small_bench.py > /dev/null 2> $O/fs_trace.log
db=$(find /tmp/fs -name '*.db' | head -1)
python - $db <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, count(*), avg(end-start), min(end-start) from kernels where name like '%fit_small%' group by name").fetchall()
for r in rows: print(r[0][:60], r[1], "avg %.1f us min %.1f us" % (r[2] / 1e3, r[3] / 1e3))
rows = cur.execute("select grid_x, count(*), avg(end-start) from kernels where name like '%fit_small%' group by grid_x").fetchall()
print(rows)
PY
timeout 300 python bench_configs.py C1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['C1']['host_api_fit_step'], d['C1']['fit_s'])"
GPX_FIT_SMALL=0 timeout 300 python bench_configs.py C1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('general', d['C1']['host_api_fit_step'], d['C1']['fit_s'])"
