"""Fingerprint of the kernel sources (pymra_amd/csrc/* and include/mra_hip.h): the PMC traffic summaries under profiles/ are
stamped with it when they are collected, and bench.py quotes their byte counts only while the stamp matches the sources the
running library was built from (the sources travel with the library: gpurun snapshots the tree)."""
import glob
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha256(root=ROOT):
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(root, "pymra_amd", "csrc", "*"))) + [os.path.join(root, "include", "mra_hip.h")]
    for f in files:
        if os.path.isfile(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    return h.hexdigest()


if __name__ == "__main__":
    print(csrc_sha256())
