set -u
mkdir -p gpurun_out/h34
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/h34/kt -- python3 tools/bench_train.py --model recnext_m3 --batch 128 --steps 6 --which hip > gpurun_out/h34/kt.log 2>&1
f=$(find gpurun_out/h34/kt -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/h34/train_kernel_stats.csv; rm -rf gpurun_out/h34/kt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/h34/train_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
rcx=[r for r in rows if 'rcx::' in r['Name']]
print('total GPU ms', tot/1e6, 'rcx ms', sum(int(r['TotalDurationNs']) for r in rcx)/1e6)
for r in rcx[:22]: print(r['Name'][:78].ljust(78), r['Calls'], round(int(r['TotalDurationNs'])/1e6,2),'ms', round(float(r['AverageNs'])/1e3,1),'us')
PY
