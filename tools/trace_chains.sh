cd /tmp && export TMPDIR=/tmp && export PYTHONPATH=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03/trace_chains -o tc -- python3 $GRAFT_REPO_ROOT/tools/time_chains.py > $GRAFT_REPO_ROOT/gpurun_out/r03/trace_chains.log 2>&1
cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r03/trace_chains/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'at::' in n or 'rocclr' in n: continue
    print(f"{n[:90]:90s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.2f} us")
PY
