#!/bin/bash
# PMC passes for profiles/ (run ON the GPU box, from anywhere):  bash tools/pmc_passes.sh <out_dir> [bench args...]
# One counter group per rocprofv3 run, never combined with tracing; the program itself follows `--`.
set -o pipefail
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$1"; shift
ARGS=("$@"); [ ${#ARGS[@]} -eq 0 ] && ARGS=(--steps 10 --warmup 2 --no-cpu-baseline --no-other-configs)
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for grp in "fetch:FETCH_SIZE" "write:WRITE_SIZE" "sq:SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES" "grbm:GRBM_GUI_ACTIVE" "lds:SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS"; do
    name="${grp%%:*}"; counters="${grp#*:}"
    rocprofv3 --pmc $counters -d "$OUT/pmc_$name" -o run --output-format csv -- python3 "$R/bench.py" "${ARGS[@]}" \
        > "$OUT/bench_$name.json" 2> "$OUT/$name.err" || { echo "pass $name failed"; tail -5 "$OUT/$name.err"; exit 1; }
    echo "pass $name done"
done
