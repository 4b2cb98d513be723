#!/usr/bin/env python3
"""Capture the reference outputs recorded by the survey session into tests/golden/.

The survey ran the UNMODIFIED reference
cpu/cpu_baseline.cpp in this container and left its inputs and outputs under
/tmp/oracle.  This script copies those DATA files (inputs + expected outputs,
no reference source) into small committed fixtures.  It does not build or run
the reference.  If /tmp/oracle is gone the committed fixtures stay as they are.

  ref_ties_fwd / ref_ties_rev : the 9-vector tie probes of SURVEY.md 0.1-5
  ref_synth10k                : 10 000 x 128 synthetic SIFT-like base, 100 queries
"""
import os
import shutil
import sys

import numpy as np

SRC = sys.argv[1] if len(sys.argv) > 1 else "/tmp/oracle"
DST = os.path.dirname(os.path.abspath(__file__))


def rf(p):
    a = np.fromfile(p, dtype=np.int32)
    d = int(a[0])
    return a.reshape(-1, d + 1)[:, 1:].view(np.float32)


def main():
    if not os.path.isdir(SRC):
        print("no survey scratch at", SRC, "- keeping committed fixtures")
        return
    for tag, sub, name in (("fwd", "tie/siftsmall", "siftsmall"), ("rev", "tie/sift", "sift")):
        shutil.copy(f"{SRC}/{sub}/{name}_base.fvecs", f"{DST}/ref_ties_{tag}_base.fvecs")
        shutil.copy(f"{SRC}/{sub}/{name}_query.fvecs", f"{DST}/ref_ties_{tag}_query.fvecs")
        shutil.copy(f"{SRC}/tie/{name}_results.txt", f"{DST}/ref_ties_{tag}_results.txt")
    base = rf(f"{SRC}/siftsmall/siftsmall_base.fvecs")
    query = rf(f"{SRC}/siftsmall/siftsmall_query.fvecs")
    assert np.all(base == np.rint(base)) and base.min() >= 0 and base.max() <= 255
    assert np.all(query == np.rint(query)) and query.min() >= 0 and query.max() <= 255
    np.savez_compressed(f"{DST}/ref_synth10k_inputs.npz", base=base.astype(np.uint8), query=query.astype(np.uint8))
    shutil.copy(f"{SRC}/siftsmall_results.txt", f"{DST}/ref_synth10k_results.txt")
    print("captured")


if __name__ == "__main__":
    main()
