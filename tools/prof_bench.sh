#!/bin/bash
# rocprofv3 kernel stats of a short bench.py run -> gpurun_out/prof_bench/
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout ${2:-400} rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_bench -o bench -- python $ROOT/bench.py --steps ${1:-2} --warmup 1 --no-cpu-baseline > $ROOT/gpurun_out/prof_bench.log 2>&1 < /dev/null
tail -2 $ROOT/gpurun_out/prof_bench.log | cut -c1-400
F=$(find $ROOT/gpurun_out/prof_bench -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then python - "$F" <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))[1:]
tot = sum(float(r[2]) for r in rows)
print("total kernel time %.1f ms over %d kernels" % (tot / 1e6, len(rows)))
for r in rows[:28]:
    print("  %6.2f%%  calls=%-6s avg=%9.1fus  %s" % (float(r[4]), r[1], float(r[3]) / 1e3, r[0][:110]))
PY
else echo "no stats"; fi
