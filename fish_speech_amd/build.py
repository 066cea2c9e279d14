"""Build libfishmi.so (HIP, gfx950 only) in-tree with hipcc.  `python -m fish_speech_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libfishmi.so")
SOURCES = ["common.cpp", "dualar_kernels.hip", "dualar_gemm.hip", "dualar_attn.hip", "dualar_sample.hip", "dualar.hip",
           "dac_kernels.hip", "dac.hip"]
HEADERS = ["common.h", "dualar_kernels.h", "dualar_dev.h", "dac_kernels.h", os.path.join("..", "..", "include", "fishmi.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result"]


def decode_sources_sha() -> str:
    """sha1 over the sources of the Dual-AR decode kernels: what a PMC traffic pass (profiles/pmc_traffic.json) was
    collected for.  bench.py compares it with the tree it runs from, so a stale figure cannot pass as current."""
    import hashlib

    h = hashlib.sha1()
    for name in ("common.h", "dualar_kernels.h", "dualar_dev.h", "dualar_kernels.hip", "dualar_gemm.hip", "dualar_attn.hip",
                 "dualar_sample.hip", "dualar.hip"):
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source for gfx950 and link the shared library.  Returns its path."""
    if not force and not _stale():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = []
    t0 = time.time()
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        if not force and os.path.exists(o) and os.path.getmtime(o) > max(
                os.path.getmtime(s), *(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS
                                       if os.path.exists(os.path.join(CSRC, h)))):
            continue
        cmd = [_hipcc(), *FLAGS, "-c", s, "-o", o]
        if s.endswith(".cpp"):
            cmd = [_hipcc(), *FLAGS, "-x", "hip", "-c", s, "-o", o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    link = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    r = subprocess.run(link, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        sys.stderr.write(r.stdout.decode(errors="replace"))
        raise RuntimeError("link failed")
    if verbose:
        print(f"[fish_speech_amd] built {LIB} in {time.time() - t0:.1f}s", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
