#!/usr/bin/env python3
"""All per-section instruction-mix reports of a round (development aid):   python3 tools/bbprof_reports.py r06 [--src <trace_kernel.hip the profiled library was built from>]

For every workload with a basic-block profile under profiles/<round>/bbprof/ runs tools/bbprof_report.py with the SQ_INSTS_* counters of the
un-instrumented kernel from profiles/<round>/<workload>/summary.json (tools/summarize_profiles.py) and writes profiles/<round>/isa_mix_by_section_<workload>.md."""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.chdir(ROOT)
tag = sys.argv[1]
extra = sys.argv[2:]
for prof in sorted(glob.glob(f"profiles/{tag}/bbprof/bbprof_*.json")):
    wl = os.path.basename(prof)[len("bbprof_"):-len(".json")]
    pmc = []
    summary = f"profiles/{tag}/{wl}/summary.json"
    if os.path.exists(summary):
        s = json.load(open(summary))
        counters = s.get("pmc_mean_per_launch", {}).get("rayhs_trace_exact", {})
        pmc = [f"{k}={v}" for k, v in counters.items() if k.startswith("SQ_INSTS")]
    out = subprocess.run([sys.executable, "tools/bbprof_report.py", f"profiles/{tag}/bbprof/bbmap.json.gz", prof] + extra + pmc, capture_output=True, text=True)
    if out.returncode != 0:
        sys.exit(out.stderr)
    open(f"profiles/{tag}/isa_mix_by_section_{wl}.md", "w").write(out.stdout)
    print(f"profiles/{tag}/isa_mix_by_section_{wl}.md  ({len(pmc)} hardware counters beside the totals)")
