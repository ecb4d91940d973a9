"""table of the 1,200-epoch self-play runs filed under profiles/ (train.log of hanabi_sad_amd.selfplay): seed, eval score and share of perfect
games at the last epoch, mean / min / max of the last 20 evaluations, train rate, wall time.  usage: python tools/convergence_summary.py [logs...]"""
import glob, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
logs = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_selfplay_convergence_1200*train.log")))
rows = []
for path in logs:
    txt = open(path).read()
    ev = [(int(e), float(s), float(p)) for e, s, p in re.findall(r"epoch (\d+), eval score: ([\d.]+), perfect: ([\d.]+)", txt)]
    if not ev:
        continue
    seed = re.search(r"'seed': (\d+)", txt)
    rate = [float(x) for x in re.findall(r"Speed: train: ([\d.]+)", txt)]
    tt = re.findall(r"Total Time: (\d+)H (\d+)M (\d+)S", txt)
    last = [s for _, s, _ in ev[-20:]]
    rows.append((os.path.basename(path), seed.group(1) if seed else "?", ev[-1][0], ev[-1][1], ev[-1][2], sum(last) / len(last), min(last), max(last),
                 sum(rate[-50:]) / max(len(rate[-50:]), 1), (int(tt[-1][0]) * 60 + int(tt[-1][1]) + int(tt[-1][2]) / 60.0) if tt else float("nan")))
print("| run | seed | last epoch | eval score | perfect % | mean of last 20 evals (min–max) | train k seq/s | minutes |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    print("| `%s` | %s | %d | %.2f | %.1f | %.2f (%.2f–%.2f) | %.1f | %.1f |" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8] / 1e3, r[9]))
if len(rows) > 1:
    import statistics as st
    fin = [r[5] for r in rows]
    print("\nmean over runs of the last-20 mean: %.2f, spread (stdev) %.2f, range %.2f–%.2f" % (st.mean(fin), st.pstdev(fin), min(fin), max(fin)))
