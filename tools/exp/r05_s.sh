cd $GRAFT_REPO_ROOT
O=gpurun_out/r05; mkdir -p $O
S=8192 CTX=1 timeout 600 python tools/small_n_sweep.py 2>&1 | tail -5
S=4096 CTX=1 SIZES=128,1,100 timeout 600 python tools/small_n_sweep.py 2>&1 | tail -1
timeout 600 python bench_configs.py C4 C2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C4', d['C4']['posteriors_per_s'], d['C4']['batch'], 'C2 sweep', d['C2']['sweep_posteriors_per_s'])"
timeout 1500 python -m pytest tests/test_gpu_edges.py tests/test_gpu_exactgp.py tests/test_gpu_vgp.py tests/test_gpu_kernels.py -x -q -m gpu > $O/tests_s.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|Error" $O/tests_s.log | tail -3
bash tools/exp/r05_o.sh 2>&1 | head -8
