"""HBM-side bytes per launch of each hot-path kernel kind from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; they do not
fit one pass: MI355X_MICROARCH.md, TCC counter budget), corrected as that guide prescribes: FETCH_SIZE x2 on gfx950, WRITE_SIZE
as reported; both counters are in KB (1024 B).  Inputs are rocprofv3 `--output-format csv` counter_collection files.

    python tools/pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv CONFIG OUT.json "<command the passes profiled>"
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rnntransducer_amd.csrc.build import source_digest  # noqa: E402

KINDS = (("gemm_hp_kernel", ("gemm_hp_kernel", "gemm_hp3_kernel")),
         ("gemm_bf16s_kernel", ("gemm_bf16s_kernel", "gemm_bf16s256_kernel")), ("gemm_f32_kernel", ("gemm_f32_kernel",)),
         ("lstm_fwd_kernel", ("lstm_fwd5_kernel", "lstm_fwd3_kernel", "lstm_fwd2_kernel", "lstm_fwd_kernel")),
         ("lstm_bwd_kernel", ("lstm_bwd5f_kernel", "lstm_bwd5_kernel", "lstm_bwd4_kernel", "lstm_bwd2_kernel", "lstm_bwd_kernel")),
         ("hp_split_kernels", ("hp_split_kernel", "hp_split_t_kernel", "hp_split_both_kernel", "hp_colmax_kernel")),
         ("lse_kernel", ("lse_sep_kernel", "lse_dense_kernel")), ("alphabeta_kernel", ("alphabeta_kernel",)),
         ("lattice_grad_kernel", ("grad_sep_kernel", "grad_dense_kernel", "reduce_dc_kernel")))
HELPERS = ("reduce_dc", "hp_colmax")  # helper launches: their bytes count towards the kind, their launches do not


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return {k: (v[0], v[1]) for k, v in acc.items()}


def match(name, pats):
    return any(("::" + p + "<") in name or ("::" + p + "(") in name or name.startswith(p + "<") or name.startswith(p + "(") or
               (" " + p + "<") in name or (" " + p + "(") in name for p in pats)


def main():
    fetch_csv, write_csv, config, out_path, command = sys.argv[1:6]
    fetch, write = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    out, detail = {}, {}
    for kind, pats in KINDS:
        tot, launches, rd, wr = 0.0, 0, 0.0, 0.0
        for name, (val, n) in fetch.items():
            if match(name, pats):
                rd += 2.0 * val * 1024.0
                if not any(h in name for h in HELPERS):
                    launches += n
        for name, (val, n) in write.items():
            if match(name, pats):
                wr += val * 1024.0
        if launches:
            out[kind] = (rd + wr) / launches
            detail[kind] = {"launches": launches, "read_bytes_per_launch": rd / launches, "write_bytes_per_launch": wr / launches}
    json.dump({"source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) -- {command}; FETCH_SIZE doubled per "
                         "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE as reported; KB = 1024 B",
               "from": out_path, "config": config, "csrc_sha16": source_digest(), "hbm_bytes_per_launch": out, "detail": detail}, open(out_path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
