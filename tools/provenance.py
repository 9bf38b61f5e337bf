"""What a committed profile was taken FROM: a hash of the kernel sources and the C-ABI header.  tools/pmc_summary.py and
tools/trace_kernel_avg.py store it in the files they write under profiles/; bench.py quotes a profile's figures beside a run only when the
profile's configuration key AND this hash match the tree it runs from (task 8)."""
import hashlib
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha() -> str:
    h = hashlib.sha256()
    base = os.path.join(REPO, "compactfusion_amd", "csrc")
    files = sorted(os.path.join(base, f) for f in os.listdir(base) if f.endswith((".hip", ".h")))
    files.append(os.path.join(REPO, "include", "cfx.h"))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_sha())
