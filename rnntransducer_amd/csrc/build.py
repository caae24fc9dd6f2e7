"""Builds librnnt_hip.so (hand-written gfx950 HIP kernels behind the C ABI of include/rnnt_hip.h) in-tree.

    python -m rnntransducer_amd.csrc.build [--force]

hipcc cross-compiles for gfx950 without a GPU; the .so is git-ignored but travels to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
INCLUDE = os.path.join(ROOT, "include")
SOURCES = ["api.hip", "gemm.hip", "gemm_hp.hip", "loss.hip", "lstm.hip", "lstm5.hip", "decode.hip", "frontend.hip"]
LIB = os.path.join(HERE, "librnnt_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + INCLUDE, "-I" + HERE, "-Wno-unused-result"]
# Per-file extras (none by default).  Tried for lstm5.hip: -mllvm -amdgpu-mfma-vgpr-form (no MFMA accumulator in an AGPR: one wave issues a
# v_mfma_f32_16x16x32_f16 with an AGPR accumulator every 25 cycles instead of every 20, tools/mfma_f16_rate_probe.hip) — the
# recurrences then need 254 VGPRs and no AGPRs, and a c2 step takes 31.31 instead of 31.16 ms (paired runs): not used.
EXTRA_FLAGS = {}
if os.environ.get("RNNT_BUILD_VGPR_FORM"):
    EXTRA_FLAGS = {"lstm5.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def source_digest() -> str:
    """sha256 (first 16 hex digits) over the kernel sources: stamps measurements that belong to ONE build (profiles/pmc_traffic_latest.json;
    bench.py drops the stamped numbers when the sources have changed since)."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES + ["common.hpp", "lstm_shared.hpp"]):
        with open(os.path.join(HERE, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    return h.hexdigest()[:16]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, s) for s in SOURCES] + [os.path.join(HERE, "common.hpp"), os.path.join(HERE, "lstm_shared.hpp"),
                                                         os.path.join(INCLUDE, "rnnt_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(HERE, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(HERE, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    return LIB


def build_variant(name: str, defines, only=None) -> str:
    """A/B builds for measurements: librnnt_hip_<name>.so with extra -D flags (objects under build/<name>/); load it with RNNT_HIP_LIB.
    `only`: the sources the flags concern (the other objects are taken from the main build)."""
    build()
    out_dir = os.path.join(HERE, "build", name)
    os.makedirs(out_dir, exist_ok=True)
    lib = os.path.join(HERE, f"librnnt_hip_{name}.so")
    procs, objs = [], []
    for src in SOURCES:
        if only and src not in only:
            objs.append(os.path.join(HERE, src.replace(".hip", ".o")))
            continue
        obj = os.path.join(out_dir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [HIPCC] + FLAGS + list(defines) + EXTRA_FLAGS.get(src, []) + ["-c", os.path.join(HERE, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout)
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:   # python -m rnntransducer_amd.csrc.build --variant NAME -DFOO=1 ...
        i = sys.argv.index("--variant")
        only = [a[len("--only="):].split(",") for a in sys.argv if a.startswith("--only=")]
        print(build_variant(sys.argv[i + 1], [a for a in sys.argv[i + 2:] if a.startswith("-D")], only[0] if only else None))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
