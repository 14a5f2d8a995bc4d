"""Build librfx.so (the HIP C-ABI library) in-tree with hipcc for gfx950.

``python -m remixfusion_amd.build`` or ``remixfusion_amd.build.build_library()``.
The library is compiled with -ffp-contract=off so that only explicit fmaf() calls fuse
(bit-parity with the oracle), and with IEEE-correct fp32 division / sqrt.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from typing import List

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librfx.so")
SOURCES = ["rfx_tsdf.hip", "rfx_field.hip", "rfx_render.hip", "rfx_tracker.hip", "rfx_mesh.hip", "rfx_pose.hip", "rfx_ba.hip", "rfx_optim.hip"]
HEADERS = ["rfx_common.h", "rfx_field_device.h", "rfx_field_mlp.h", os.path.join("..", "..", "include", "rfx.h")]

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
    "-Wall", "-Wno-unused-function", "-DHASH_GROUP=4", "-DFWD_WAVES=2", "-DRENDER_WAVES=2", "-DBWD_WAVES=2",
]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: librfx.so cannot be built (no CPU fallback exists)")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def sources_digest() -> str:
    """sha256 over the files that define the kernels (csrc/*.hip, csrc/*.h, include/rfx.h, the compiler flags): the identity
    of the binary for measurements taken in another run -- profiles/*_pmc_*.json carry it, and bench.py refuses a `traffic`
    figure whose digest is not this tree's (the GPU box has no .git to ask)."""
    import hashlib
    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS):
        path = os.path.join(CSRC, name)
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:16]


def build_library(force: bool = False, verbose: bool = False, extra: List[str] | None = None, out: str | None = None) -> str:
    """``extra``/``out``: A/B builds (tools/build_variant.py) -- later -D flags override the defaults above; such a
    library is selected at run time with RFX_LIB_PATH (see _lib.py)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if out is not None:
        cmd = [_hipcc()] + FLAGS + (extra or []) + ["-o", out] + srcs
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            sys.stderr.write(res.stdout + res.stderr)
            raise RuntimeError("hipcc failed building " + out)
        return out
    if not force and not _stale():
        return LIB
    cmd = [_hipcc()] + FLAGS + (extra or []) + ["-o", LIB + ".tmp"] + srcs
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed building librfx.so")
    if verbose and res.stderr:
        sys.stderr.write(res.stderr)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
