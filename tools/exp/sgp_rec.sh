cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof
timeout 600 python tools/c5_bench.py > gpurun_out/prof/c5_sparse.json 2> gpurun_out/prof/c5_sparse.err
python -c "
import json; d=json.load(open('gpurun_out/prof/c5_sparse.json')); print('bound %.2f ms (%.3f)  bound+grad %.2f ms (%.3f; counted %.3e model %.3e)  posterior %.1f ms (%.3f)  api step %.2f ms pred %.3f s' % (d['bound']['ms'], d['bound']['frac_of_fp64_peak'], d['bound_and_gradient']['ms'], d['bound_and_gradient']['frac_of_fp64_peak'], d['bound_and_gradient']['mfma_flops_counted'], d['bound_and_gradient']['mfma_flops_model'], d['posterior_all_pixels']['ms'], d['posterior_all_pixels']['frac_of_fp64_peak'], d['viSparseGP_api']['ms_per_svi_step'], d['viSparseGP_api']['predict_in_batches_all_pixels_s']))"
