cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/sgp_ab}
mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "sparse or sgp or Sparse" > $O/t.log 2>&1; echo "sparse tests rc=$?"; tail -3 $O/t.log
for v in inverse sweep; do
GPX_SGP_SOLVE=$v timeout 600 python tools/c5_bench.py 2> $O/c5_$v.err | tee $O/c5_sparse_$v.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v bound %.2f ms  bound+grad %.2f ms (%.3f)  posterior %.1f ms (%.3f)  api step %.2f ms pred %.3f s' % (d['bound']['ms'], d['bound_and_gradient']['ms'], d['bound_and_gradient']['frac_of_fp64_peak'], d['posterior_all_pixels']['ms'], d['posterior_all_pixels']['frac_of_fp64_peak'], d['viSparseGP_api']['ms_per_svi_step'], d['viSparseGP_api']['predict_in_batches_all_pixels_s']))"
done
