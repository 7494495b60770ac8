"""Side-by-side per-op table of the variants in a tools/variant_rows.sh log: python tools/ab_table.py <log>"""
import re
import sys

txt = open(sys.argv[1]).read()
blocks = re.split(r"=== variant \[(.*?)\]", txt)[1:]
names, rows = [], {}
for i in range(0, len(blocks), 2):
    v = blocks[i]
    names.append(v)
    for line in blocks[i + 1].splitlines():
        m = re.match(r"(\S.*?)\s+(\(\d+, \d+\)|None)\s+([\d.]+) us", line)
        if m:
            rows.setdefault(m.group(1).strip()[:34] + " " + m.group(2), {})[v] = float(m.group(3))
        m = re.match(r"(fp16 B.*forward|mode \d: eager.*forward) ([\d.]+) us", line)
        if m:
            rows.setdefault(m.group(1)[:30], {})[v] = float(m.group(2))
print(" " * 48 + " ".join(f"{n[-10:]:>10s}" for n in names))
for k, d in rows.items():
    print(f"{k[:48]:48s}" + " ".join(f"{d.get(n, 0):10.1f}" for n in names))
