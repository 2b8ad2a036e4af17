set -u
mkdir -p gpurun_out/h66
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/h66/kt -- python3 tools/bench_fwd_train.py 128 > gpurun_out/h66/log.txt 2>&1
f=$(find gpurun_out/h66/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# consecutive groups of 30 launches of the same kernel name
groups = []
for r in rows:
    name = r["Kernel_Name"]
    if "rcx::" not in name or "pack" in name: continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if groups and groups[-1][0] == name and len(groups[-1][1]) < 30: groups[-1][1].append(d)
    else: groups.append((name, [d]))
for name, ds in groups:
    ds = sorted(ds)
    print(f"{len(ds):3d} x  median {ds[len(ds)//2]:7.1f} us  min {ds[0]:7.1f}   {name[:90]}")
PY
rm -rf gpurun_out/h66/kt
