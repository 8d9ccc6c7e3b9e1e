"""Hash of the kernel sources a profile was taken with: profiles/pmc_traffic.json and the SQ counter
summary carry it, and bench.py only quotes them when the hash still matches the tree it runs."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ("slam_amd/csrc/orbx_extract.hip", "slam_amd/csrc/orbx_match.hip", "slam_amd/csrc/orbx_internal.h",
         "slam_amd/csrc/orbx_common.hip")


def kernel_source_hash():
    h = hashlib.sha256()
    for f in FILES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())
