# sgp_soak REPS times under $1 = GPX_SGP_SOLVE value; prints the failures
cd $GRAFT_REPO_ROOT
v=${1:-ride}; reps=${2:-12}; fails=0
for i in $(seq 1 $reps); do
  out=$(GPX_SGP_SOLVE=$v timeout 300 python tools/exp/sgp_soak.py 2>&1 | tail -4)
  if ! echo "$out" | grep -q "sgp_soak ok"; then fails=$((fails+1)); echo "rep $i:"; echo "$out"; fi
done
echo "GPX_SGP_SOLVE=$v: $fails failures in $reps repetitions"
