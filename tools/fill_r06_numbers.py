"""Fills the R6_* placeholders of README.md / DESIGN.md from the closing bench line (tools/r06_final.sh).  usage: fill_r06_numbers.py <bench line json> [--print]
[--tests N SECONDS]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(sys.argv[1]))
e = d.get("exec_equivalent") or {}
nr = d.get("no_readout_step") or {}
sec = d.get("secondary") or {}
c2 = (sec.get("config2_retina_unet_128_b8") or {}).get("value")
c5 = (sec.get("config5_inference_512x512x256_bf16") or {}).get("patches_per_s")
c1 = (sec.get("config1_toy2d_retina_net_64x64_b20") or {}).get("images_per_s")
vals = {
    "R6_VALUE": d["value"], "R6_MS": d["ms_per_step"], "R6_EXEC_SYNC": (e.get("synchronous_readout_form") or {}).get("value"),
    "R6_EXEC_GRAPH": (e.get("graphed_form") or e.get("eager_form") or {}).get("value"), "R6_EXEC": e.get("value"),
    "R6_NOREAD_MS": nr.get("ms_per_step"), "R6_NOREAD": nr.get("value"), "R6_HEADS": (d.get("heads_full_step") or {}).get("value"),
    "R6_DENSE": (d.get("dense_rpn_graph_step") or {}).get("value"), "R6_ROOF": "%s (%s µs)" % ((d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("avg_us")),
    "R6_CPU": (d.get("cpu_baseline") or {}).get("value"), "R6_SECONDARY": "%s / %s / %s" % (c2, c5, c1), "R6_RETINA": c2, "R6_CFG5": c5,
    "R6_GRAPH": (d.get("graphed_step") or {}).get("value"),
}
if "--print" in sys.argv:
    print({k: v for k, v in vals.items()})
    sys.exit(0)
if "--tests" in sys.argv:
    i = sys.argv.index("--tests")
    vals["R6_NTESTS"], vals["R6_TSEC"] = sys.argv[i + 1], sys.argv[i + 2]
import re
for name in ("README.md", "DESIGN.md"):
    p = os.path.join(ROOT, name)
    s = open(p).read()
    for k, v in vals.items():          # idempotent: the value sits between invisible markers  <!--R6_X-->value<!--/-->
        s = re.sub(r"<!--%s-->.*?<!--/-->" % k, "<!--%s-->%s<!--/-->" % (k, v), s)
    open(p, "w").write(s)
