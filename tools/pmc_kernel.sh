#!/bin/bash
# per-launch means of rocprofv3 PMC counters for kernels matching $1, one pass per counter group:
#   bash tools/pmc_kernel.sh <kernel substring> "<counters of pass 1>" "<counters of pass 2>" ... -- <command>
pat=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
export TMPDIR=/tmp
for c in "${groups[@]}"; do
  d=/tmp/pmc_$$_$RANDOM
  (cd /tmp && rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d -- "$@" > $d.log 2>&1)
  f=$(ls -t $d/*/*counter_collection.csv 2>/dev/null | head -1)
  [ -z "$f" ] && { echo "no counter file for: $c"; tail -5 $d.log; continue; }
  python3 - "$f" "$pat" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print("%-28s launches %3d  mean per launch %.5g" % (k, len(v), sum(v) / len(v)))
PY
done
