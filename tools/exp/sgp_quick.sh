cd $GRAFT_REPO_ROOT
O=gpurun_out/sgp_ab; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "sparse or sgp or Sparse or forward_pass" > $O/t.log 2>&1; echo "sparse tests rc=$?"; tail -3 $O/t.log
timeout 600 python tools/c5_bench.py 2>/dev/null | tee $O/c5_new.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('bound %.2f ms (%.3f)  bound+grad %.2f ms (%.3f; counted %.3e model %.3e)  posterior %.1f ms (%.3f)  api step %.2f ms pred %.3f s' % (d['bound']['ms'], d['bound']['frac_of_fp64_peak'], d['bound_and_gradient']['ms'], d['bound_and_gradient']['frac_of_fp64_peak'], d['bound_and_gradient']['mfma_flops_counted'], d['bound_and_gradient']['mfma_flops_model'], d['posterior_all_pixels']['ms'], d['posterior_all_pixels']['frac_of_fp64_peak'], d['viSparseGP_api']['ms_per_svi_step'], d['viSparseGP_api']['predict_in_batches_all_pixels_s']))"
