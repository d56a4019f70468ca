#!/bin/bash
# rocprofv3 kernel stats of the op-level probes of round 6's new kernels -> gpurun_out/r06/ (copied to profiles/r06/ by the builder)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $ROOT/gpurun_out/r06
cd /tmp && export TMPDIR=/tmp
for probe in conv1x1_fwd_probe conv_seg_probe; do
  rm -rf /tmp/prof_$probe
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$probe -o p -- python $ROOT/tools/$probe.py > /tmp/prof_$probe.log 2>&1
  S=$(find /tmp/prof_$probe -name "*kernel_stats.csv" | head -1)
  python - "$S" > $ROOT/gpurun_out/r06/r06_${probe}_kernel_stats.csv <<'PY'
import csv, re, sys
rows = list(csv.reader(open(sys.argv[1])))
print("kernel,calls,avg_us,min_us,max_us")
for r in rows[1:]:
    if any(k in r[0] for k in ("conv1x1_fwd", "conv1x1_bwd", "rpn_heads", "conv_seg", "conv_c0")):
        m = re.search(r'(\w+_kernel)(<[^>]*>)?', r[0])
        print('"%s",%s,%.1f,%.1f,%.1f' % (m.group(1) + (m.group(2) or ''), r[1], float(r[3]) / 1e3, float(r[5]) / 1e3, float(r[6]) / 1e3))
PY
  cat $ROOT/gpurun_out/r06/r06_${probe}_kernel_stats.csv
done
