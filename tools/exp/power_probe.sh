# board power and shader clock while the C3 sweep runs (is the plateau a power limit?)
mkdir -p gpurun_out/r2
rocm-smi --showpower --showclocks --showmaxpower 2>&1 | grep -v "^$" | head -30 > gpurun_out/r2/power_idle.txt
( python bench.py --steps 240 --warmup 5 --no-cpu-baseline > gpurun_out/r2/power_bench.json 2> gpurun_out/r2/power_bench.err ) &
BP=$!
sleep 6
: > gpurun_out/r2/power_load.txt
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|mclk\|fclk" >> gpurun_out/r2/power_load.txt
  echo "--" >> gpurun_out/r2/power_load.txt
  sleep 0.7
done
wait $BP
cat gpurun_out/r2/power_idle.txt | head -20
cat gpurun_out/r2/power_load.txt | head -40
cut -c1-200 gpurun_out/r2/power_bench.json
