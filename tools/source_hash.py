"""Hash of the kernel sources a profile was taken with: profiles/pmc_traffic.json and the SQ counter
summary carry it, and bench.py only quotes them when the hash still matches the tree it runs."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ("slam_amd/csrc/orbx_extract.hip", "slam_amd/csrc/orbx_match.hip", "slam_amd/csrc/orbx_internal.h",
         "slam_amd/csrc/orbx_common.hip")


def kernel_source_hash():
    """Lines are hashed without their indentation, trailing blanks and empty lines, so that a change
    of layout alone does not make the committed counter summaries look stale."""
    h = hashlib.sha256()
    for f in FILES:
        for line in open(os.path.join(ROOT, f), "rb").read().split(b"\n"):
            line = line.strip()
            if line:
                h.update(line + b"\n")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_source_hash())
