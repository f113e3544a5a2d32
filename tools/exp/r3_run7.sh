cd $GRAFT_REPO_ROOT
O=gpurun_out/r03g
mkdir -p $O
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --steps 9 --warmup 3 > $O/b_$name.json 2> $O/b_$name.err
  python - $O/b_$name.json "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
s = d["stages"]
print(f"{sys.argv[2]}: potrf {s['potrf_ms']:.2f} predict {s['predict_ms']:.2f} fit {s['fit_step_ms']:.2f} post/s {d['value']:.2f} frac {d['roofline']['frac']:.3f}")
PY
}
for rep in 1 2; do
run default_$rep A=1
run tail48_$rep GPX_TAIL_TILES=48
run tail56_$rep GPX_TAIL_TILES=56
run tail64_$rep GPX_TAIL_TILES=64
run tail88_$rep GPX_TAIL_TILES=88
run fau24_$rep GPX_FAR_AFTER_U1=24
run fau56_$rep GPX_FAR_AFTER_U1=56
run lg3_$rep GPX_LAZY_GROUP=3
done 2>&1 | tee $O/sweep.txt
