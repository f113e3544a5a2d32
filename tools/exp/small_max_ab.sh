#!/bin/bash
# EXPERIMENT: the tile count below which a launch of a chain that has the chip to itself takes the 64 x 64 shape
out=gpurun_out/r05/smallmax; mkdir -p $out
for v in 400 800 1400 100000; do
  GPX_SMALL_MAX=$v timeout 300 python bench_configs.py C2 C5 > $out/cfg_$v.json 2> $out/cfg_$v.err
  GPX_SMALL_MAX=$v timeout 200 python tools/exp/env_ab.py GPX_NOP 3000 5120 8192 > $out/ab_$v.json 2> $out/ab_$v.err
  python - <<PY
import json
r=json.load(open("$out/cfg_$v.json"))
c2=r["C2"]; c5=r["C5"]
print("small_max $v  C2", {k: round(x,3) for k,x in c2["stages"].items()}, "sweep", round(c2["sweep_posteriors_per_s"]), " C5 bound", round(c5["sparse_bound"]["ms"],2), "b+g", round(c5["sparse_bound_and_gradient"]["ms"],2), "all px", round(c5["sparse_posterior_all_pixels"]["ms"],1), "viGP step", round(c5["viGP_exact_api"]["ms_per_svi_step"],1))
PY
  cut -c1-140 $out/ab_$v.err
done
