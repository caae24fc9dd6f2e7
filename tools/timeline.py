"""Prints the kernel timeline of ONE training step from a rocprofv3 --kernel-trace csv (…_kernel_trace.csv): start offset, duration,
queue, short name — to see what actually overlaps with the persistent recurrences.  usage: python tools/timeline.py <csv> [step_index]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.replace("rnnt::", "")
    return name.split("(")[0][:60]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), short(r["Kernel_Name"])) for r in rows]
    ev.sort()
    # steps are delimited by the AdamW kernel
    ends = [i for i, e in enumerate(ev) if e[3].startswith("adamw_flat_kernel")]
    lo, hi = ends[which - 1] + 1, ends[which] + 1
    t0 = ev[lo][0]
    print(f"step kernels {lo}..{hi}: {(ev[hi - 1][1] - t0) / 1e6:.3f} ms")
    for s, e, q, n in ev[lo:hi]:
        if e - s < 20000 and not n.startswith(("lstm", "gemm_hp")):
            continue
        print(f"{(s - t0) / 1e6:9.3f} ms  +{(e - s) / 1e3:9.1f} us  q{q:>3}  {n}")


if __name__ == "__main__":
    main()
