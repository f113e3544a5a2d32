# bit-reproducibility soak of the sparse step under the sparse-solve schedules and the latency-GEMM kernels
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in ride inverse; do for lg in r5 r1; do
echo "== $v $lg (rep $rep)"; GPX_SGP_SOLVE=$v GPX_LAT_GEMM=$lg timeout 600 python tools/exp/sgp_soak.py 2>&1 | tail -4
done; done; done
