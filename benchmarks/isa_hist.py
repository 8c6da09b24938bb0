#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -S listing: python benchmarks/isa_hist.py file.s <mangled-name-substring>..."""
import collections
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
for key in sys.argv[2:]:
    starts = [i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l]
    if not starts:
        print(key, "not found")
        continue
    start = starts[0]
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    ops = collections.Counter(m.group(1) for l in lines[start:end] if (m := re.match(r"\s+([a-z_0-9]+)", l)))
    meta = [l.strip() for l in lines if key in l and (".num_vgpr" in l or ".numbered_sgpr" in l or "scratch" in l and ".set" in l)]
    print(f"== {key}: {end - start} lines")
    print("  ", ", ".join(f"{o} {c}" for o, c in ops.most_common(22)))
    for m in meta[:3]:
        print("  ", m.split(".")[-1])
