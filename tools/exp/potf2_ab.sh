cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f
mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_edges.py -x -q -m gpu -k "wave_specialised" > $O/t.log 2>&1; echo "test rc=$?"; tail -8 $O/t.log
for m in tile chain tile chain; do GPX_POTF2=$m timeout 120 python tools/potf2_time.py; done 2>&1 | tee $O/potf2_time.txt
for rep in 1 2; do for m in tile chain; do
  GPX_POTF2=$m timeout 300 python bench.py --no-cpu-baseline --steps 9 --warmup 3 > $O/bench_$m_$rep.json 2> $O/bench_$m_$rep.err
  python - $O/bench_$m_$rep.json $m <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
s = d["stages"]
print(f"potf2={sys.argv[2]} potrf {s['potrf_ms']:.2f} predict {s['predict_ms']:.2f} fit {s['fit_step_ms']:.2f} post/s {d['value']:.2f} frac {d['roofline']['frac']:.3f} lml {d['lml_check']} classes {d['kernel_classes_ms_per_predict']}")
PY
done; done 2>&1 | tee $O/ab.txt
for N in 512 1024 2048 4096; do for m in tile chain; do
  GPX_POTF2=$m timeout 300 python bench.py --N $N --M 256 --no-cpu-baseline --steps 6 --warmup 2 --inflight 1 2>/dev/null | python -c "
import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); s=d['stages']; print('N=$N potf2=$m potrf %.3f predict %.3f fit %.3f ms' % (s['potrf_ms'], s['predict_ms'], s['fit_step_ms']))"
done; done 2>&1 | tee $O/smalln.txt
