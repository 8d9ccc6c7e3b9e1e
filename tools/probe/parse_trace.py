import collections, csv, re, sys
per = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"\bk_[a-z0-9_]+", r["Kernel_Name"])
    if m:
        per[m.group(0)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print("%-28s calls %5d  avg %8.1f us  total %9.1f us" % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e3))
