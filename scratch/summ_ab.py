# usage: summ_ab.py file   -- means per variant of a scratch/ab_ms.sh log with "--<args>" section lines
import re, collections, sys
sec = None
acc = collections.OrderedDict()
for l in open(sys.argv[1]):
    l = l.strip()
    if l.startswith("--"):
        sec = l
        continue
    m = re.match(r"(\w+): ([\d.]+) ([\d.]+)", l)
    if m:
        acc.setdefault((sec, m.group(1)), []).append(float(m.group(2)))
for k, v in acc.items():
    print(k[0], k[1], " ".join("%.2f" % x for x in v), "mean %.2f" % (sum(v) / len(v)))
