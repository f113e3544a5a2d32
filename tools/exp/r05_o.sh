cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/sw; S=8192 CTX=1 SIZES=512,1,100 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/sw -- python tools/small_n_sweep.py > /dev/null 2>&1
db=$(find /tmp/sw -name '*.db' | head -1)
python tools/rocpd_summary.py $db | head -40
