set -e
for n in 65536; do
echo "== $n off"; GAS_XCD_ORDER_AUTO_MIN=4000000000 tools/ab_libs.sh "--sources-per-gpu $n" -
echo "== $n on"; tools/ab_libs.sh "--sources-per-gpu $n" -
echo "== $n native xcd directions, order off"; GAS_XCD_ORDER_AUTO_MIN=4000000000 tools/ab_libs.sh "--sources-per-gpu $n --xcd-directions" -
echo "== $n native xcd directions, order on"; tools/ab_libs.sh "--sources-per-gpu $n --xcd-directions" -
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in off on; do
  if [ $m = off ]; then export GAS_XCD_ORDER_AUTO_MIN=4000000000; else unset GAS_XCD_ORDER_AUTO_MIN; fi
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/xcd_$m -o f -- python3 $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extras --sources-per-gpu 65536 > $R/gpurun_out/xcd_$m.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/xcd_$m/**/*counter_collection.csv",recursive=True)[0]
agg={}
for r in csv.DictReader(open(f)):
    if r["Counter_Name"]=="FETCH_SIZE":
        a=agg.setdefault(r["Kernel_Name"][:40],[0,0]); a[0]+=float(r["Counter_Value"]); a[1]+=1
for k,(s,c) in agg.items():
    if "k_" in k: print("$m",k,c,"avg KiB",s/c)
PY
done
