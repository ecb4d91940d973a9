"""Generates tests/golden/topk_saver_trace.json by driving the REFERENCE's TopkSaver (pyhanabi/common_utils/saver.py) with fixed score
sequences: per call the returned flag and, afterwards, which model{i}.pthw holds which score.  Data only.  Run:

    python tests/golden/make_topk_trace.py"""
import json
import os
import sys
import tempfile

sys.path.insert(0, "/root/reference/pyhanabi")
import torch  # noqa: E402
from common_utils.saver import TopkSaver  # noqa: E402  (the reference)

CASES = {"topk3": (3, [5.0, 1.0, 3.0, 0.5, 4.0, 4.0, 6.0, 2.0, 7.0, 6.5]),
         "topk5": (5, [0.2, 0.1, 0.4, 0.3, 0.35, 0.5, 0.05, 0.45, 0.6, 0.41, 0.42]),
         "topk1": (1, [1.0, 0.5, 2.0, 2.0, 3.0])}
out = {}
for name, (k, scores) in CASES.items():
    with tempfile.TemporaryDirectory() as d:
        s = TopkSaver(d, k)
        steps = []
        for p in scores:
            flag = s.save(None, {"w": torch.tensor([p])}, p)
            files = {f: float(torch.load(os.path.join(d, f))["w"]) for f in sorted(os.listdir(d)) if f.startswith("model")}
            steps.append({"score": p, "saved": bool(flag), "files": files})
        out[name] = {"topk": k, "steps": steps}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "topk_saver_trace.json"), "w"), indent=1)
print("wrote", {k: len(v["steps"]) for k, v in out.items()})
