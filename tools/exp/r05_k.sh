cd $GRAFT_REPO_ROOT
GPX_FS_TRACE=1 timeout 300 python tools/fit_small_bench.py 2>&1 | grep FS_TRACE | sort | uniq -c | sort -k3 | awk '{print}' | head -60
