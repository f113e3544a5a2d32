# kernel timeline of one C3 factorisation with the current defaults, and with the U1 split, for comparison
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
for v in 0 2; do
rm -rf /tmp/tl$v
GPX_U1_SPLIT=$v timeout 200 rocprofv3 --kernel-trace -d /tmp/tl$v -- python tools/timeline.py 16384 > /dev/null 2>/tmp/tl$v.err
db=$(find /tmp/tl$v -name '*.db' | head -1)
python tools/timeline_dump.py $db gpurun_out/r2/timeline_u1s$v.csv
python tools/timeline_analyze.py gpurun_out/r2/timeline_u1s$v.csv > gpurun_out/r2/timeline_u1s$v.txt
head -12 gpurun_out/r2/timeline_u1s$v.txt
done
