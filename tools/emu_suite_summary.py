#!/usr/bin/env python
"""tools/emu_suite_summary.py -- TEST INFRASTRUCTURE: one line per `-m gpu` test from the logs of runs on the CPU model of the HIP execution model
(tests/emu; HOP_TEST_EMU=1 python -m pytest tests -m gpu -v ...), later logs overriding earlier ones.
    python tools/emu_suite_summary.py gpurun_out/emu/*.log > profiles/r04_emu_gpu_suite.txt"""
import re
import sys

res, dur = {}, {}
for path in sys.argv[1:]:
    for ln in open(path, errors="replace"):
        m = re.search(r"(PASSED|FAILED|ERROR)\s+(tests/\S+)", ln)
        if m:
            res[m.group(2)] = m.group(1)
        m = re.match(r"\s*([0-9.]+)s call\s+(tests/\S+)", ln)
        if m:
            dur[m.group(2)] = float(m.group(1))
n_pass = sum(v == "PASSED" for v in res.values())
print("# `-m gpu` tests executed on the CPU model of the HIP execution model (tests/emu) -- NOT on hardware: the kernel sources' semantics, no timing.")
print("# %d tests with a result: %d passed, %d failed" % (len(res), n_pass, len(res) - n_pass))
for k in sorted(res):
    print("%-7s %8s  %s" % (res[k], ("%.0f s" % dur[k]) if k in dur else "", k))
