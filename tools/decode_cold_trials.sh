#!/bin/bash
# N cold trials (one fresh process each, tools/decode_cold.py) of a case; JSON lines into $OUT, dumps of deviating trials beside it.
# usage: tools/decode_cold_trials.sh N case [extra decode_cold.py flags]     (run from the repo root on the GPU box)
N=${1:-16}; CASE=${2:-b1}; shift 2
OUT=${SATT_COLD_OUT:-gpurun_out/cold}
mkdir -p $OUT
for i in $(seq 1 $N); do
  timeout 120 python tools/decode_cold.py $CASE --dump $OUT --tag t$i "$@" 2>/dev/null | grep '^{' >> $OUT/trials_$CASE.jsonl
done
python - <<EOF
import json, collections
rows = [json.loads(l) for l in open("$OUT/trials_$CASE.jsonl")]
mel = collections.Counter(r["cold"]["mel"] for r in rows)
bad = [r for r in rows if not r["cold_equals_warm"]]
print("$CASE: %d trials, %d with cold != warm, distinct cold mel hashes: %s, mean %.1f s per trial" % (len(rows), len(bad), dict(mel), sum(r["secs"] for r in rows) / max(len(rows), 1)))
for r in bad:
    print("  deviating:", r["tag"], r.get("first_differing_step"), r.get("mel_cold_vs_warm_max"), r["differing"])
EOF
