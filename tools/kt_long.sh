cd /tmp && export TMPDIR=/tmp
for t in local semi; do
rm -rf /tmp/pl; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -o p -- python $GRAFT_REPO_ROOT/tools/full_dp_long_probe.py $t > /dev/null 2>&1
python - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("/tmp/pl/*kernel_stats.csv")[0])))
for r in rows:
    if "full_gotoh" in r["Name"]: print("$t %9.2f ms avg %8.2f ms x %3s  %s"%(float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6, r["Calls"], r["Name"][:80]))
PY
done
