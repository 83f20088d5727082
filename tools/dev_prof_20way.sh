#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_20way
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_20way -o p -- python3 $R/tests/dev/probe_20way.py > $R/gpurun_out/probe_20way.txt 2>&1
cd $R
f=$(find /tmp/prof_20way -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/probe_20way_kernel_stats.csv
grep -v "^W\|^E" gpurun_out/probe_20way.txt | tail -6
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/probe_20way_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:18]:
    print(f"{r['Name'][:100]:100s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
PY
