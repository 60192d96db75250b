#!/bin/bash
# Run ON the GPU box: rocprofv3 kernel stats + the PMC passes of bench.py for the three single-GPU sizes of BASELINE.json.
#   bash tools/profile_round.sh <out_dir> <tag>        (tag: e.g. r02)
# Writes <out_dir>/<tag>_bench_n{30k,200k,1m}_kernel_stats.csv, ..._under_rocprof.json, ..._pmc_summary.csv and
# <out_dir>/<tag>_force_traffic.json (list, one entry per size).  Tracing and counters are never combined in one run.
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$(cd "$1" 2>/dev/null && pwd || (mkdir -p "$1" && cd "$1" && pwd))"; TAG="$2"
cd /tmp && export TMPDIR=/tmp
for cfg in "30k:30000:100" "200k:200000:50" "1m:1000000:10"; do
    name="${cfg%%:*}"; rest="${cfg#*:}"; n="${rest%%:*}"; steps="${rest#*:}"
    rm -rf "$OUT/trace_$name"
    rocprofv3 --kernel-trace --stats -d "$OUT/trace_$name" -o run --output-format csv -- python3 "$R/bench.py" --bodies "$n" --steps "$steps" --warmup 5 --no-cpu-baseline --no-other-configs \
        > "$OUT/${TAG}_bench_n${name}_under_rocprof.json" 2> "$OUT/trace_$name.err" || { echo "trace $name failed"; tail -5 "$OUT/trace_$name.err"; exit 1; }
    cp "$(find "$OUT/trace_$name" -name '*kernel_stats.csv' | head -1)" "$OUT/${TAG}_bench_n${name}_kernel_stats.csv"
    echo "trace $name done"
    bash "$R/tools/pmc_passes.sh" "$OUT/pmc_$name" --bodies "$n" --steps 10 --warmup 2 --no-cpu-baseline --no-other-configs || exit 1
    python3 "$R/tools/pmc_summary.py" "$OUT/pmc_$name" "$OUT/${TAG}_bench_n${name}_pmc_summary.csv" "$OUT/traffic_$name.json" "$n"
    echo "pmc $name done"
done
python3 - "$OUT" "$TAG" <<'PY'
import json, sys, os
out, tag = sys.argv[1], sys.argv[2]
entries = [json.load(open(os.path.join(out, f"traffic_{k}.json"))) for k in ("30k", "200k", "1m")]
json.dump(entries, open(os.path.join(out, f"{tag}_force_traffic.json"), "w"), indent=1)
PY
echo "profile round done"
