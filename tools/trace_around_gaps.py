import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))
t_lo = int(ev[0]["Start_Timestamp"]) + (int(ev[-1]["End_Timestamp"]) - int(ev[0]["Start_Timestamp"])) * 0.7
ev = [e for e in ev if int(e["Start_Timestamp"]) >= t_lo]
cur_end, shown = int(ev[0]["Start_Timestamp"]), 0
print(list(ev[0].keys()))
for i, e in enumerate(ev):
    s, en = int(e["Start_Timestamp"]), int(e["End_Timestamp"])
    if s - cur_end > 60000 and shown < 3:
        shown += 1
        print("---- gap %.1f us" % ((s - cur_end) / 1e3))
        for j in range(max(0, i - 6), min(len(ev), i + 4)):
            x = ev[j]
            print("%s q%s  start %+9.1f dur %7.1f  %s" % ("->" if j == i else "  ", x.get("Queue_Id", "?"), (int(x["Start_Timestamp"]) - s) / 1e3,
                                                          (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3, x["Kernel_Name"][:70]))
    cur_end = max(cur_end, en)
