cd $GRAFT_REPO_ROOT
for k in 1 0 1 0; do GRAM_KIND=$k timeout 300 python tools/gram_bench.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('kind $k gram median %.4f best %.4f ms written %.0f GB/s best %.0f  fill %.0f' % (d['median_ms'], d['best_ms'], d['written_GBps'], d['best_written_GBps'], d['write_only_fill_GBps']))"; done
