#!/usr/bin/env python3
"""Scan gfx950 device assembly for callable (non-kernel) functions whose long-branch expansions write s[30:31] — the return
address — without the prologue having saved it (see VA_KEEP_RETURN_ADDRESS in csrc/va_rt.hpp).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o /tmp/eng.s cedarsim.jl_amd/csrc/ch_engine.hip
    python scripts/check_return_address.py /tmp/eng.s        (exit code 1 when a function has the pattern)
"""
import re
import sys

text = open(sys.argv[1]).read()
kernels = set(re.findall(r"\.amdhsa_kernel (\S+)", text))
bad = []
for m in re.finditer(r"^(\w+):\s*; @\1\n(.*?)^\s*\.size\s+\1,", text, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if name in kernels or "s_setpc_b64 s[30:31]" not in body:
        continue
    writes = len(re.findall(r"s_getpc_b64 s\[30:31\]", body))
    saved = re.search(r"v_writelane_b32 v\d+, s30, \d+", body) is not None or re.search(r"s_mov_b64 s\[\d+:\d+\], s\[30:31\]", body) is not None
    if writes and not saved:
        bad.append((name, writes))
for name, n in bad:
    print("return address clobbered by %d long-branch expansions and never saved: %s" % (n, name))
print("%d callable functions checked against the pattern, %d bad" % (len(re.findall(r"^\w+:\s*; @", text, re.M)) - len(kernels), len(bad)))
sys.exit(1 if bad else 0)
