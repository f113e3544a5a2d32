cd $GRAFT_REPO_ROOT
GPX_SGP_SOLVE=ride timeout 900 python -m pytest tests/test_gpu_sparse.py tests/test_gpu_parity_fullsize.py -x -q -m gpu -k "sparse or c5" 2>&1 | tail -3
GPX_SGP_SOLVE=ride timeout 600 python tools/exp/sgp_soak.py 2>&1 | tail -2
GPX_LAT_GEMM=r5 timeout 900 python -m pytest tests/test_gpu_parity_fullsize.py -x -q -m gpu -k "c3_matern or c2_rbf or c4_shape" 2>&1 | tail -2
