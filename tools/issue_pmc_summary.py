#!/usr/bin/env python3
"""Joins the rocprofv3 counter passes of tools/issue_pmc.sh per dispatch of tools/bench_issue.cpp.
Per (kernel, waves per SIMD): vector / matrix wave-instructions, GRBM_GUI_ACTIVE / 8 XCDs = shader cycles of the launch,
cycles per instruction and SIMD (1024 SIMDs), effective clock = cycles / kernel duration (kernel trace of the same pass).
usage: python tools/issue_pmc_summary.py <dir with p*/ ...csv>"""
import csv
import glob
import sys
from collections import OrderedDict, defaultdict

root = sys.argv[1]
disp = OrderedDict()
for f in sorted(glob.glob(root + "/p*/**/*counter_collection.csv", recursive=True)):
    pas = f.split("/p")[-1][:1] if "/p" in f else "?"
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0]
        if not name.startswith("k_") or name == "k_layout":
            continue
        key = (f, row["Dispatch_Id"])
        d = disp.setdefault(key, {"name": name, "grid": int(row["Grid_Size"]), "wg": int(row["Workgroup_Size"]), "c": {}})
        d["c"][row["Counter_Name"]] = d["c"].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
dur = {}
for f in sorted(glob.glob(root + "/p*/**/*kernel_trace.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        dur[(f.replace("kernel_trace", "counter_collection"), row["Dispatch_Id"])] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
# keep the LAST dispatch of each (file, kernel, grid): the tool launches three repetitions
last = OrderedDict()
for key, d in disp.items():
    last[(key[0], d["name"], d["grid"])] = (key, d)
agg = defaultdict(dict)
for (f, name, grid), (key, d) in last.items():
    waves = grid // 64
    wps = waves // 1024
    a = agg[(name, wps)]
    a.update(d["c"])
    if key in dur:
        a.setdefault("ns", []).append(dur[key])
print(f"{'kernel':30s} w/SIMD   valu_inst  mfma_inst   cycles(GRBM/8)  cyc/inst/SIMD  clock GHz  valu_active%  mfma_busy%  wait_inst%  wait_any%")
for (name, wps), a in agg.items():
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8.0
    nv, nm = a.get("SQ_INSTS_VALU", 0), a.get("SQ_INSTS_MFMA", 0)
    n = nv + nm if nm == 0 or nv < nm else nv   # SQ_INSTS_VALU includes the matrix instructions on this part: shown separately
    per = cyc / (max(nv, nm) / 1024.0) if max(nv, nm) else 0
    ns = sum(a.get("ns", [0])) / max(len(a.get("ns", [0])), 1)
    wc = a.get("SQ_WAVE_CYCLES", 0)
    # SQ_* cycle counters count quad-cycles summed over waves (guide): normalise by wave-cycles of the second pass
    act = a.get("SQ_ACTIVE_INST_VALU", 0)
    tot = a.get("SQ_ACTIVE_INST_ANY", 0) + a.get("SQ_WAIT_INST_ANY", 0) + a.get("SQ_WAIT_ANY", 0)
    pct = lambda x: 100.0 * x / tot if tot else 0.0
    mb = 100.0 * a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024) if cyc else 0.0
    print(f"{name:30s} {wps:6d} {nv:11.0f} {nm:10.0f} {cyc:16.0f} {per:14.2f} {cyc / ns if ns else 0:10.2f} {pct(act):12.1f} {mb:11.1f} {pct(a.get('SQ_WAIT_INST_ANY', 0)):11.1f} {pct(a.get('SQ_WAIT_ANY', 0)):10.1f}")
