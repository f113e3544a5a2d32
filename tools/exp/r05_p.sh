cd $GRAFT_REPO_ROOT
for sz in 256,1,100 512,1,100 1024,2,256 2048,2,1024; do
for k in 512 128; do
echo "big-shape K threshold $k"; GPX_BATCH_BIG_K=$k S=4096 CTX=1 SIZES=$sz timeout 300 python tools/small_n_sweep.py 2>&1 | tail -1
done; done
