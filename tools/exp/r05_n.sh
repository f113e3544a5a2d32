cd $GRAFT_REPO_ROOT
for sz in 128,1,100 256,1,100 512,1,100 1024,2,256; do
for cap in 256 512 1024 2048; do
echo "cap $cap"; GPX_SWEEP_BATCH=$cap GPX_SWEEP_BATCH_CAP=$cap S=8192 CTX=1 SIZES=$sz timeout 300 python tools/small_n_sweep.py 2>&1 | tail -1
done; done
