"""one-GPU training rate with and without the early draw, alternating on ONE box (bench.training_bench)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
for rep in range(3):
    for early in (1, 0):
        r = bench.training_bench(torch.device("cuda:0"), extra_args=("--early_draw", str(early)))
        print("early_draw %d: %.0f sequences/s, %.3f ms per iteration, host issue %.3f (%.3f on an empty queue)" % (
            early, r["value"], r["ms_per_iteration"], r["host_issue_ms_per_iteration"], r["host_issue_ms_per_iteration_on_an_empty_queue"]), flush=True)
