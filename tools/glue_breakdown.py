"""Per-call view of the NON-convolution kernels of one training step, from a rocprofv3 --kernel-trace CSV: which torch elementwise / reduce /
sort launches (full functor name, grid) the ~4 ms/step of glue kernel time is made of.  The aggregated stats CSV truncates torch's
template names into two rows; this keeps them apart.
Usage: glue_breakdown.py <kernel_trace.csv> [n_steps_from_the_end=3] [marker=adam_flat]   (a step ends with the marker kernel)"""
import collections
import csv
import re
import sys


def short(name):
    m = re.search(r"at::native::(?:\(anonymous namespace\)::)?(\w+)<(.*)", name)
    if not m:
        return name[:110]
    inner = m.group(2)
    f = re.search(r"(\w+(?:Functor|functor|Op|Ops|_kernel_cuda|Kernel)\w*(?:<[^<>]*>)?)", inner)
    lam = re.search(r"at::native::(\w+)\(at::TensorIterator(?:Base)?&\)::\{lambda", inner)
    tag = f.group(1) if f else (lam.group(1) if lam else inner[:70])
    return "{}<{}>".format(m.group(1), tag)[:110]


def main():
    path = sys.argv[1]
    n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marker = sys.argv[3] if len(sys.argv) > 3 else "adam_flat"
    marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
    if len(marks) < n_steps + 1:
        raise SystemExit("fewer than {} steps in the trace".format(n_steps + 1))
    lo, hi = marks[-n_steps - 1] + 1, marks[-1] + 1
    win = rows[lo:hi]
    span = (int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])) / 1e6 / n_steps
    agg = collections.defaultdict(lambda: [0, 0.0])
    conv = collections.defaultdict(lambda: [0, 0.0])
    conv_ms = 0.0
    for r in win:
        n = r["Kernel_Name"]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        if n.startswith("ck::") or n.startswith("_ZN2ck") or "Cijk_" in n or "conv" in n.lower() or "igemm" in n.lower() or "naive" in n.lower():
            conv_ms += d / 1e3
            g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            ck = (re.sub(r"\(anonymous namespace\)::|void ", "", n)[:60], g, int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]))
            conv[ck][0] += 1
            conv[ck][1] += d
            continue
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        k = (short(n), grid)
        agg[k][0] += 1
        agg[k][1] += d
    tot = sum(v[1] for v in agg.values()) / 1e3 / n_steps
    print("# {} steps, {:.2f} ms/step span; convolution kernels {:.2f} ms/step; everything else {:.2f} ms/step in {} launches/step".format(
        n_steps, span, conv_ms / n_steps, tot, sum(v[0] for v in agg.values()) // n_steps))
    print("# us/step  calls/step  avg_us  threads  kernel")
    for (name, grid), (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
        print("{:8.1f} {:6.1f} {:8.1f} {:10d}  {}".format(us / n_steps, c / n_steps, us / c, grid, name))


    print("# convolution kernels by (name, grid threads, VGPRs): us/step  calls/step  avg_us")
    for (name, grid, vg), (c, us) in sorted(conv.items(), key=lambda kv: -kv[1][1])[:45]:
        print("{:8.1f} {:6.1f} {:8.1f} {:10d} {:4d}  {}".format(us / n_steps, c / n_steps, us / c, grid, vg, name))


    # the LAST step, launch by launch (time order): every convolution-class launch >= 60 us with its position in the step -- layers are
    # identified by order (forward: stem, C2..C5, FPN, RPN, heads; backward in reverse), the CK names say nothing
    last = rows[marks[-2] + 1:marks[-1] + 1]
    t0 = int(last[0]["Start_Timestamp"])
    print("# last step in time order: convolution-class launches >= 60 us:  t_ms  us  threads  vgpr  lds  name")
    for r in last:
        n = r["Kernel_Name"]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        is_conv = n.startswith("ck::") or n.startswith("_ZN2ck") or "Cijk_" in n or "conv" in n.lower()
        if is_conv and d >= 60:
            kind = "wgrad" if ("bwd_weight" in n or "wgrad" in n) else ("bwd_data" if "bwd_data" in n else "fwd-type")
            g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            print("{:8.2f} {:8.1f} {:9d} {:4d} {:6d}  {:9s} {}".format((int(r["Start_Timestamp"]) - t0) / 1e6, d, g, int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"]),
                                                              int(r["LDS_Block_Size"]), kind, re.sub(r"\(anonymous namespace\)::|void ", "", n)[:48]))


    print("# last step in time order: every OTHER launch >= 40 us:  t_ms  us  threads  name")
    for r in last:
        n = r["Kernel_Name"]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        is_conv = n.startswith("ck::") or n.startswith("_ZN2ck") or "Cijk_" in n or "conv" in n.lower()
        if not is_conv and d >= 40:
            g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
            print("{:8.2f} {:8.1f} {:9d}  {}".format((int(r["Start_Timestamp"]) - t0) / 1e6, d, g, short(n)[:100]))


if __name__ == "__main__":
    main()
