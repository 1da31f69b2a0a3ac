#!/usr/bin/env python3
"""Copies one tools/profile_configs.sh run (gpurun_out/<tag>/cfgN/) into profiles/<round>_cfgN_<ver>.md: bench line, kernel trace, PMC, stats.
usage: python tools/store_configs.py r04_cfg v1 "note" """
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, ver = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
rnd = tag.split("_", 1)[0]
for c in (2, 3, 4, 5):
    src = os.path.join(ROOT, "gpurun_out", tag, f"cfg{c}")
    if not os.path.exists(os.path.join(src, "bench.json")):
        continue
    bench = json.load(open(os.path.join(src, "bench.json")))
    rd = lambda n: open(os.path.join(src, n)).read() if os.path.exists(os.path.join(src, n)) else "(missing)\n"
    out = f"# BASELINE config {c} on one MI355X (tools/profile_configs.sh, {rnd} {ver}: {note})\n\n## bench line (python bench.py --config {c})\n\n```json\n"
    out += json.dumps(bench, indent=1) + "\n```\n\n## kernel trace (rocprofv3 --kernel-trace --stats)\n\n" + rd("kernel_stats.md")
    out += "\n## PMC passes, dominant kernel (per launch)\n\n```\n" + rd("pmc_k_packet.txt") + "```\n\n## M2S_STATS\n\n```\n" + rd("stats.txt") + "```\n"
    dst = os.path.join(ROOT, "profiles", f"{rnd}_cfg{c}_{ver}.md")
    open(dst, "w").write(out)
    r = bench["roofline"]
    print(f"config {c}: {bench['ms_per_step']:.3f} ms per step, roofline frac {r['frac']:.5f}, traffic {r['traffic'] / 1e6 if r['traffic'] else 0:.0f} MB = {r['traffic'] / r['algorithmic_bytes_per_launch'] if r['traffic'] else 0:.2f} x algorithmic -> {dst}")
