cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/vigp_pib}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multidevice.py tests/test_gpu_models.py "tests/test_gpu_parity_fullsize.py::test_c5_exact_vigp_on_the_512x512_image" -x -q -m gpu > $O/t.log 2>&1; echo "tests rc=$?"; tail -3 $O/t.log
timeout 600 python tools/c5_bench.py 2> $O/c5.err | tee $O/c5_sparse.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['viGP_exact_api'], d['viSparseGP_api'])"
