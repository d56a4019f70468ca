"""Steady-state kernel breakdown of the timed training steps from a rocprofv3 --kernel-trace csv of bench.py.

usage: python tools/steady_state.py <*_kernel_trace.csv> <steps> <ms_per_step> [out.csv]
The window ends at the last optimizer (adam_flat_kernel / multi_tensor_apply) kernel of the trace -- run bench.py with --no-h2d-leg
--no-rccl-selftest so that the last Adam launch belongs to the last timed step -- and spans steps * ms_per_step before it."""
import csv
import re
import sys
from collections import defaultdict


def category(n):
    if "crop_" in n or "nms_" in n or "bias_act" in n or "bias_grad" in n or "anchor" in n or "decode" in n or "wbc" in n \
            or "maxpool_k3" in n or "filter_flip" in n or "match_pass" in n or "upsample2x_" in n or "zero_fill_kernel" in n or "nms2to3d" in n or "adam_flat" in n or "adam_segments" in n or "s2d221_" in n \
            or "roi_levels" in n or "rpn_sample" in n or "detection_targets" in n or "refine_pre" in n or "refine_post" in n or "refine_fallback" in n or "rpn_patch" in n:
        return "mdt_hip (this repo)"
    if "conv1x1_wgrad" in n or "conv1x1_fwd" in n or "conv_c0" in n or "conv_seg" in n or "conv1x1_bwd" in n or "rpn_heads" in n or "conv3x3x3_small" in n or "conv_stem_wgrad" in n or "conv_stem_fwd" in n or "conv1x1_dgrad_add" in n or "conv_s221_wgrad" in n \
            or "conv_s221_fwd" in n or "conv_s221_dgrad" in n:
        return "mdt_hip convolution kernels (this repo, fp32 MFMA)"
    if n.startswith("_ZN2ck") or "ck::" in n or "miopen" in n.lower() or "Cijk" in n or "gemm" in n.lower() or "batched_transpose" in n \
            or "naive_conv" in n or "SubTensor" in n or "Im2" in n or "Col2" in n or "igemm" in n:
        return "MIOpen / CK / GEMM convolutions"
    if "max_pool" in n or "upsample" in n:
        return "torch pooling / upsampling"
    if "fillBuffer" in n or "copyBuffer" in n:
        return "fills / copies (runtime)"
    if "at::native" in n or "at_cuda" in n or "cub" in n or "rocprim" in n:
        return "torch elementwise / reduce / sort"
    return "other"


def short(n):
    n = re.sub(r"^void ", "", n)
    if n.startswith("_ZN2ck"):
        kind = "bwd_weight" if "bwd_weight" in n else ("fwd" if "fwd" in n else "ck")
        return "ck::" + kind + " (mangled) ..." + n[-24:]
    return re.sub(r"<.*", "", n)[:90]


def recategorize(path, steps):
    """Rebuild the per-category lines of an existing summary from its own per-kernel rows (after `category` learnt new kernel names)."""
    text = open(path).read().splitlines()
    head = [l for l in text if l.startswith("# window") or l.startswith("# idle") or l.startswith("# gap")]
    body = text[text.index("Name,Calls,TotalDurationNs,AverageNs,Percentage"):]
    cat = defaultdict(float)
    for r in csv.DictReader(body):
        cat[category(r["Name"])] += float(r["TotalDurationNs"])
    tot = sum(cat.values())
    lines = head[:1] + ["# %-52s %8.2f ms/step  %5.1f %% of kernel time" % (c, v / 1e6 / steps, 100.0 * v / tot)
                        for c, v in sorted(cat.items(), key=lambda x: -x[1])] + head[1:] + body
    open(path, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))


def main():
    if sys.argv[1] == "--recategorize":      # python tools/steady_state.py --recategorize <summary.csv> <steps>
        return recategorize(sys.argv[2], int(sys.argv[3]))
    path, steps, ms = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    rows = list(csv.DictReader(open(path)))
    ks = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
    adam = [e for s, e, n in ks if "multi_tensor_apply" in n or "adam_flat_kernel" in n]
    if not adam:
        raise SystemExit("no optimizer (multi_tensor_apply / adam_flat_kernel) launch in the trace")
    t1 = max(adam)
    t0 = t1 - int(steps * ms * 1e6)
    win = [(s, e, n) for s, e, n in ks if s >= t0 and e <= t1]
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in win:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += (cur_e - cur_s) if cur_e else 0
    span = t1 - t0
    # idle gaps between consecutive kernels (single in-order stream)
    gaps = []
    for (s0, e0, n0), (s1, e1, n1) in zip(win[:-1], win[1:]):
        if s1 > e0:
            gaps.append((s1 - e0, short(n0), short(n1), (e0 - t0) / 1e6))
    bins = [(0, 5), (5, 20), (20, 100), (100, 1000), (1000, 1e9)]
    gap_lines = ["# idle gaps: " + "; ".join("%g-%g us: n=%d, %.2f ms/step" % (
        lo, hi, sum(1 for g in gaps if lo * 1e3 <= g[0] < hi * 1e3), sum(g[0] for g in gaps if lo * 1e3 <= g[0] < hi * 1e3) / 1e6 / steps)
        for lo, hi in bins)]
    for g in sorted(gaps, reverse=True)[:14]:
        gap_lines.append("# gap %7.1f us at +%7.2f ms  after %-50s before %s" % (g[0] / 1e3, g[3], g[1][:50], g[2][:50]))
    cat, per = defaultdict(float), defaultdict(lambda: [0, 0.0])
    for s, e, n in win:
        cat[category(n)] += e - s
        k = short(n)
        per[k][0] += 1
        per[k][1] += e - s
    tot = sum(cat.values())
    lines = ["# window: last %d timed steps = %.1f ms; kernels %d; GPU busy %.1f ms (%.1f %%); summed kernel time %.1f ms"
             % (steps, span / 1e6, len(win), busy / 1e6, 100.0 * busy / span, tot / 1e6)]
    for c, v in sorted(cat.items(), key=lambda x: -x[1]):
        lines.append("# %-52s %8.2f ms/step  %5.1f %% of kernel time" % (c, v / 1e6 / steps, 100.0 * v / tot))
    lines += gap_lines
    lines.append("Name,Calls,TotalDurationNs,AverageNs,Percentage")
    for k, (c, v) in sorted(per.items(), key=lambda x: -x[1][1]):
        lines.append('"%s",%d,%d,%d,%.3f' % (k, c, v, v / c, 100.0 * v / tot))
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 4:
        open(sys.argv[4], "w").write(text)
    print("\n".join(lines[:60]))


if __name__ == "__main__":
    main()
