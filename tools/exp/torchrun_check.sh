cd $GRAFT_REPO_ROOT
O=gpurun_out/torchrun; mkdir -p $O
# the driver's own launch line for N = 2, on this 1-GPU box: LOCAL_RANK 1 wraps to GPU 0, RCCL refuses the duplicate
# device, all ranks agree on the file transport
t0=$(date +%s)
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 6 --warmup 2 > $O/out.txt 2> $O/err.txt; echo "rc=$? seconds=$(( $(date +%s) - t0 ))"
python - $O/out.txt <<'PY'
import json, sys
lines = open(sys.argv[1]).read().splitlines()
js = [l for l in lines if l.startswith("{")]
print("stdout lines", len(lines), "json lines", len(js), "last line is json:", lines[-1].startswith("{") if lines else None)
d = json.loads(js[-1])
print({k: d.get(k) for k in ("value", "n_gpus", "steps", "multi_gpu_path", "rccl_ranks", "ms_per_step")})
print("c4", d.get("c4_sweep"))
print("node", d.get("node_sweep"))
PY
tail -5 $O/err.txt | cut -c1-300
