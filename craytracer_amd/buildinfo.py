"""Identity of the kernel sources a measurement belongs to.

Hardware-counter figures are collected in runs of their own (rocprofv3 --pmc) and quoted later beside
fresh timings; the hash of the kernel sources ties the two together: a record is only quoted as
current when it was taken on the sources that are in the tree now.
"""
from __future__ import annotations

import glob
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def kernel_source_files():
    """The files libcaster_hip.so is built from, in a fixed order."""
    csrc = os.path.join(_HERE, "csrc")
    files = []
    for pat in ("*.hpp", "*.hip", "*.h", "*.cpp", "Makefile"):
        files += glob.glob(os.path.join(csrc, pat))
    files.append(os.path.join(os.path.dirname(_HERE), "include", "caster_hip.h"))
    return sorted(files)


def kernel_source_hash() -> str:
    """sha256 over the names and bytes of the kernel sources (first 16 hex digits)."""
    h = hashlib.sha256()
    for path in kernel_source_files():
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]
