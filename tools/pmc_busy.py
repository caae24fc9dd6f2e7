"""Matrix-core / wave occupancy counters per kernel from a rocprofv3 `--pmc ... --output-format csv` pass over the bench step.
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x kernel cycles); kernel cycles from GRBM_GUI_ACTIVE (summed over the 8
XCDs: / 8) when collected, else from SQ_BUSY_CYCLES.   python tools/pmc_busy.py counter_collection.csv [kernel_trace.csv]"""
import collections
import csv
import sys

def short(name):
    return name.replace("rnnt::(anonymous namespace)::", "").replace("void ", "").split("(")[0]


acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    n = short(r["Kernel_Name"])
    a = acc[n][r["Counter_Name"]]
    a[0] += float(r["Counter_Value"])
    a[1] += 1
dur = {}
if len(sys.argv) > 2:
    d = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(sys.argv[2])):
        n = short(r["Kernel_Name"])
        d[n][0] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3
        d[n][1] += 1
    dur = {k: v[0] / v[1] for k, v in d.items()}
print(f"{'kernel':58s} {'launches':>8s} {'avg_us':>9s} {'MFMA_BUSY/launch':>17s} {'mfma_util':>9s} {'WAIT_ANY%':>9s} {'WAIT_INST%':>10s} {'ACTIVE%':>8s}")
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", [0, 1])[0]):
    g = lambda name: (c[name][0] / c[name][1]) if name in c and c[name][1] else float("nan")
    n = max(v[1] for v in c.values())
    wc = g("SQ_WAVE_CYCLES")
    us = dur.get(k, float("nan"))
    # SIMD-cycles available in one launch: 1024 SIMDs x duration x the clock the chip held (GRBM_GUI_ACTIVE / 8 per launch when present)
    cyc = g("GRBM_GUI_ACTIVE") / 8.0 if "GRBM_GUI_ACTIVE" in c else us * 2.0e3
    util = g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc) if cyc == cyc and cyc > 0 else float("nan")
    print(f"{k[:58]:58s} {n:8d} {us:9.1f} {g('SQ_VALU_MFMA_BUSY_CYCLES'):17.3e} {util:9.3f} {100 * g('SQ_WAIT_ANY') / wc:9.1f} "
          f"{100 * g('SQ_WAIT_INST_ANY') / wc:10.1f} {100 * g('SQ_ACTIVE_INST_ANY') / wc:8.1f}")
