cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O
timeout 900 python -m pytest tests/test_bench_launch.py -x -q -m gpu 2>&1 | tail -4
GPX_LIB=gpax_amd/lib/libgpx_trace.so timeout 300 python tools/potf2_trace.py > $O/potf2_phase_trace.json 2> $O/potf2_phase_trace.err; echo "trace rc=$?"
python bench.py --force-rank-path --steps 12 --warmup 3 > $O/bench_rank1_rccl.json 2> $O/bench_rank1_rccl.err; echo "rank1 rc=$?"
python bench.py --gpus 2 --share-gpu --steps 8 --warmup 2 --c4-S 200 > $O/bench_2ranks_shared_gpu.json 2> $O/bench_2ranks_shared_gpu.err; echo "2ranks rc=$?"
GPX_RANK_FORCE_COLLECTIVES=1 python bench.py --force-rank-path --steps 4 --warmup 1 --c4-S 24 --no-node-record > $O/bench_rank1_rccl_forced_collectives.json 2> $O/forced.err; echo "forced rc=$?"
python bench.py > $O/bench_r04.json 2> $O/bench_r04.err; echo "bench rc=$?"
cut -c1-300 $O/bench_r04.json
