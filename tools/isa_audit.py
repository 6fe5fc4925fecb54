#!/usr/bin/env python3
"""ISA audit of the LDS-DMA kernels: `s_waitcnt vmcnt(0)` INSIDE a loop (the compiler's own wait in front of a register that was loaded
before the loop, or in front of an LDS access it cannot tell from an in-flight DMA: the prefetch then degenerates to load-then-wait).
usage: tools/isa_audit.py file.s [...]   (hipcc -S --cuda-device-only output)"""
import re, sys
for path in sys.argv[1:]:
    fn, depth_of = None, {}
    cur_depth = 0
    hits = {}
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            fn, cur_depth = m.group(1), 0
            continue
        if line.startswith(".Lfunc_end"):
            fn = None
            continue
        if fn is None:
            continue
        m = re.match(r"^\.LBB\d+_\d+:\s*;\s*(.*)$", line)
        if m:
            c = m.group(1)
            d = re.search(r"Depth=(\d+)", c)
            cur_depth = int(d.group(1)) if d else (cur_depth if "in Loop" in c else 0)
            continue
        if re.match(r"^\.LBB\d+_\d+:", line):
            cur_depth = 0
            continue
        if "s_waitcnt vmcnt(0)" in line and cur_depth > 0:
            hits.setdefault(fn, []).append((ln, cur_depth))
    for f, h in hits.items():
        print(f"{path}: {f[:90]}: {len(h)} x vmcnt(0) inside loops at lines {[l for l, _ in h][:8]}")
