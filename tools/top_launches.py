"""The longest kernel launches of the LAST step of a rocprofv3 kernel trace (steps are delimited by the optimizer's launch), with launch order,
grid size and duration: which LAYERS the convolution time of a step sits in.

    python tools/top_launches.py <kernel_trace.csv> [n=60] [step_marker=adam_flat]"""
import csv
import re
import sys


def main():
    path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
    marker = sys.argv[3] if len(sys.argv) > 3 else "adam_flat"
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", "")))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(marks) < 2:
        print("no two '%s' launches in the trace" % marker)
        return
    a, b = marks[-2] + 1, marks[-1] + 1
    step = rows[a:b]
    span = (step[-1][1] - step[0][0]) / 1e6
    print("# last step: %d launches, %.2f ms span, %.2f ms summed kernel time" % (len(step), span, sum(e - s for s, e, *_ in step) / 1e6))
    print("# order  ms      grid      kernel")
    order = sorted(range(len(step)), key=lambda i: step[i][0] - step[i][1])[:n]
    for i in sorted(order):
        s, e, name, grid, wg = step[i]
        name = re.sub(r"^void ", "", name)
        if name.startswith("_ZN2ck"):
            kind = "bwd_weight" if "bwd_weight" in name else ("fwd" if "fwd" in name else "ck")
            name = "ck::" + kind + " ..." + name[-34:]
        print("%6d %7.3f %10s  %s" % (i, (e - s) / 1e6, grid, re.sub(r"<.*", "", name)[:100]))


if __name__ == "__main__":
    main()
