"""Run-level utilities of a self-play job: log tee, top-k checkpoint saver, wall-clock section timer, running statistics and
the Tachometer that DEFINES the throughput metric.

They reproduce what the reference's drivers print and save, so that logs and checkpoint directories of the two stacks can be
compared line by line (pyhanabi/common_utils/{logger,saver,stopwatch,multi_counter,helper}.py, pyhanabi/utils.py:218-251), but
hold no device work: everything hot stays in libhsad."""
import os
import random
import sys
import time

import numpy as np
import torch


def set_all_seeds(seed):
    """python / numpy / torch (CPU, GPU) generators from one seed with the reference's offsets (helper.py:134-138)"""
    random.seed(seed)
    np.random.seed(seed + 1)
    torch.manual_seed(seed + 2)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed + 3)


def sec2str(seconds):
    s = int(seconds)
    return "%dH %02dM %02dS" % (s // 3600, (s % 3600) // 60, s % 60)


def num2str(n):
    for div, unit in ((1e6, "M"), (1e3, "K")):
        if n >= div:
            return "%.3f%s" % (n / div, unit)
    return str(n)


class Logger:
    """tee of stdout into <save_dir>/train.log (selfplay.py:96-97).  The first pprint(vars(args)) in that file doubles as the
    saved configuration (utils.get_train_config parses it back), so the driver prints it first."""

    def __init__(self, path, mode="w"):
        if mode not in ("w", "a"):
            raise ValueError("Logger mode must be 'w' or 'a'")
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        self.terminal = sys.stdout
        self.log = open(path, mode if os.path.exists(path) else "w")

    def write(self, message):
        self.terminal.write(message)
        self.log.write(message)
        self.log.flush()

    def flush(self):
        self.terminal.flush()


class TopkSaver:
    """the reference's checkpoint bookkeeping (common_utils/saver.py:17-61; called selfplay.py:267-273 with model=None, the online
    net's state_dict and the eval score), slot for slot -- including its quirks, so that a save_dir is comparable file by file:
    the score list starts as [-inf]; while it is shorter than `topk` every accepted candidate is written to the CURRENT worst slot
    (slot 0, whose -inf entry is only replaced once the list is full) and appended; from then on a candidate replaces the worst slot
    iff it beats it strictly, and the first minimal entry becomes the next worst slot.  Forced names (model_epochN) and latest are
    written first, whatever the score.  Returns whether the candidate was written to a model{i} slot."""

    def __init__(self, save_dir, topk):
        self.save_dir, self.topk = save_dir, int(topk)
        self.perfs = [-float("inf")]
        self.worst, self.worst_slot = self.perfs[0], 0
        os.makedirs(save_dir, exist_ok=True)

    def _write(self, model, state_dict, stem):
        if model is not None:
            model.save(os.path.join(self.save_dir, stem + ".pthm"))
        if state_dict is not None:
            torch.save(state_dict, os.path.join(self.save_dir, stem + ".pthw"))

    def save(self, model, state_dict, perf, save_latest=False, force_save_name=None):
        if force_save_name is not None:
            self._write(model, state_dict, force_save_name)
        if save_latest:
            self._write(model, state_dict, "latest")
        if perf <= self.worst:
            return False
        self._write(model, state_dict, "model%d" % self.worst_slot)
        if len(self.perfs) < self.topk:          # still filling: the list grows, the worst slot stays where it is
            self.perfs.append(perf)
            return True
        self.perfs[self.worst_slot] = perf
        self.worst = min(self.perfs)
        self.worst_slot = self.perfs.index(self.worst)      # first minimal entry
        return True


class Stopwatch:
    """named wall-clock sections between successive time() calls; summary() prints mean ms and share per section
    (common_utils/stopwatch.py:17-54; the learner loop's five buckets, selfplay.py:215-241)"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.last = time.perf_counter()
        self.times, self.keys = {}, []

    def time(self, key):
        now = time.perf_counter()
        if key not in self.times:
            self.times[key] = []
            self.keys.append(key)
        self.times[key].append((now - self.last) * 1e3)
        self.last = time.perf_counter()

    def summary(self):
        if not self.keys:
            return
        n = len(self.times[self.keys[0]])
        total = sum(sum(v) for v in self.times.values())
        width = max(len(k) for k in self.keys)
        print("@@@Time")
        for k in self.keys:
            v = self.times[k]
            print("\t%s: %d MS, %.2f%%" % (k.ljust(width), np.mean(v), 100.0 * sum(v) / max(total, 1e-12)))
        print("@@@total time per iter: %.2f ms" % (total / max(n, 1)))
        self.reset()


class ValueStats:
    """count / mean / min / max (with the index they occurred at) of a fed scalar series (multi_counter.py:7-57)"""

    def __init__(self, name=None):
        self.name = name
        self.reset()

    def reset(self):
        self.counter, self.summation = 0, 0.0
        self.max_value, self.min_value, self.max_idx, self.min_idx = -1e38, 1e38, None, None

    def feed(self, v):
        v = float(v)
        self.summation += v
        if v > self.max_value:
            self.max_value, self.max_idx = v, self.counter
        if v < self.min_value:
            self.min_value, self.min_idx = v, self.counter
        self.counter += 1

    def mean(self):
        if self.counter == 0:
            raise ZeroDivisionError("ValueStats %s is empty" % self.name)
        return self.summation / self.counter

    def summary(self, info=None):
        head = "%s%s" % (info or "", self.name or "")
        if self.counter == 0:
            return "%s[0]" % head
        return "%s[%4d]: avg: %8.4f, min: %8.4f[%4d], max: %8.4f[%4d]" % (
            head, self.counter, self.mean(), self.min_value, self.min_idx, self.max_value, self.max_idx)


class MultiCounter:
    """stat["loss"].feed(x) / stat.inc("event") / stat.summary(epoch)  (multi_counter.py:60-114; selfplay.py:204,243-244,252)"""

    def __init__(self, root=None, verbose=False):
        self.verbose = verbose
        self.stats, self.counts = {}, {}
        self.total_count, self.max_key_len = 0, 0
        self.last_time = None

    def __getitem__(self, key):
        self.max_key_len = max(self.max_key_len, len(key))
        if key in self.counts:
            return self.counts[key]
        if key not in self.stats:
            self.stats[key] = ValueStats()
        return self.stats[key]

    def inc(self, key):
        if self.verbose:
            print("[MultiCounter]: %s" % key)
        self.counts[key] = self.counts.get(key, 0) + 1
        self.total_count += 1
        if self.last_time is None:
            self.last_time = time.time()

    def reset(self):
        for v in self.stats.values():
            v.reset()
        self.counts, self.total_count = {}, 0
        self.last_time = time.time()

    def time_elapsed(self):
        return time.time() - self.last_time

    def summary(self, global_counter):
        print("[%d] Time spent = %.2f s" % (global_counter, self.time_elapsed() if self.last_time else 0.0))
        for key, count in self.counts.items():
            print("%s: %d/%d" % (key, count, self.total_count))
        for k in sorted(self.stats):
            info = (str(global_counter) + ":" + k).ljust(self.max_key_len + 4)
            print(self.stats[k].summary(info=info))


class Tachometer:
    """The metric of BASELINE.json (pyhanabi/utils.py:218-251): per lap
         train = factor * num_train / dt            (sequences sampled by the learner per second)
         act   = factor * d(sum of actor.num_act) / dt   (one unit = one game advanced one move; R2D2Actor::numAct += num_envs)
         buffer_add = factor * d(replay.num_add()) / dt
    with factor = num_player for VDN (one transition holds every player's row), 1 for IQL (selfplay.py:246-250)."""

    def __init__(self):
        self.num_act = self.num_buffer = self.num_train = 0
        self.t, self.total_time = None, 0.0

    def start(self):
        self.t = time.time()

    def lap(self, actors, replay_buffer, num_train, factor):
        dt = time.time() - self.t
        self.total_time += dt
        num_act = sum(a.num_act() if callable(getattr(a, "num_act", None)) else a.num_act for a in actors)
        num_buffer = replay_buffer.num_add()
        rates = (factor * num_train / dt, factor * (num_act - self.num_act) / dt, factor * (num_buffer - self.num_buffer) / dt)
        print("Speed: train: %.1f, act: %.1f, buffer_add: %.1f, buffer_size: %d" % (rates + (replay_buffer.size(),)))
        self.num_act, self.num_buffer = num_act, num_buffer
        self.num_train += num_train
        print("Total Time: %s, %ds" % (sec2str(self.total_time), self.total_time))
        print("Total Sample: train: %s, act: %s" % (num2str(self.num_train), num2str(self.num_act)))
        return rates
