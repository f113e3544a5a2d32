cd $GRAFT_REPO_ROOT
for ot in 0 2; do for sz in 512,1,100 384,1,100 256,1,100; do
echo "batch OT $ot"; GPX_BATCH_OT=$ot S=8192 CTX=1 SIZES=$sz timeout 300 python tools/small_n_sweep.py 2>&1 | tail -1
done; done
