"""The device replay at BASELINE capacity (131,072 sequences of 80 steps of the 2-player SAD transition = 37 GB of HBM)
through size-independent properties: every sampled sequence comes back exactly as it was stored (content is a function of
the sequence's tag), eviction keeps the newest `capacity` sequences, sizes / num_add follow the reference's bookkeeping,
importance weights follow (N w / sum)^-beta / max, and priority updates move the sampling mass."""
import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]
DEV = "cuda:0"
T, CAP, CHUNK, B = 80, 131072, 8192, 128
FIELDS = [("priv_s", 838, torch.float32), ("legal_move", 21, torch.float32), ("eps", 1, torch.float32),
          ("own_hand", 15, torch.float32), ("a", 1, torch.int64), ("greedy_a", 1, torch.int64)]


def chunk(tag0, n):
    tag = torch.arange(tag0, tag0 + n, device=DEV, dtype=torch.float32)
    t = torch.arange(T, device=DEV, dtype=torch.float32)
    f = {}
    for name, w, dt in FIELDS:
        col = torch.arange(w, device=DEV, dtype=torch.float32)
        v = (tag.view(n, 1, 1) * 3 + t.view(1, T, 1) * 5 + col.view(1, 1, w) * 7) % 251
        f[name] = v.to(dt)
    reward = (tag.view(n, 1) + t.view(1, T)) % 17
    terminal = ((tag.view(n, 1) + t.view(1, T)) % 2).to(torch.uint8)
    bootstrap = 1 - terminal.float()
    seq_len = (tag % T) + 1
    prio = (tag % 13) * 0.25 + 0.25
    return f, reward, terminal, bootstrap, seq_len, prio


def test_replay_at_baseline_capacity():
    from hanabi_sad_amd.replay import DeviceReplay
    rep = DeviceReplay(CAP, 3, 0.9, 0.6, 3, T, FIELDS, DEV)
    assert rep.bytes() > 36e9
    total = CAP + 2 * CHUNK                      # overfill: the two oldest chunks must be gone afterwards
    for t0 in range(0, total, CHUNK):
        rep.add(*chunk(t0, CHUNK))
        if rep.size() > CAP:                     # the reference pops the overflow on sample(); alternate like its learner
            rep.sample(B)
            rep.update_priority(torch.full((B,), 1.0, device=DEV))
    rep.check_errors()
    assert rep.num_add() == total and rep.size() <= CAP
    oldest_alive = total - rep.size()
    (f, reward, terminal, bootstrap, seq_len), w = rep.sample(B)
    # content check: every field of a sequence is a function of r = (3 tag) % 251, readable from eps[t = 0]
    r = f["eps"][0, :, 0]
    t = torch.arange(T, device=DEV, dtype=torch.float32).view(T, 1, 1)
    for name, wd, dt in FIELDS:
        col = torch.arange(wd, device=DEV, dtype=torch.float32).view(1, 1, wd)
        want = (r.view(1, B, 1) + t * 5 + col * 7) % 251
        assert torch.equal(f[name].float(), want.to(dt).float()), name
    assert bool(((terminal.float() + bootstrap) == 1).all())
    assert bool((seq_len >= 1).all()) and bool((seq_len <= T).all())
    assert float(w.max()) == 1.0 and float(w.min()) > 0
    # eviction: element 0 of the ring is the oldest survivor = tag `oldest_alive`
    f0, r0, t0_, b0, sl0 = rep.get(0)
    assert float(f0["eps"][0, 0]) == float((oldest_alive * 3) % 251)
    assert float(sl0.reshape(-1)[0]) == float(oldest_alive % T + 1)
    # priority updates move the mass: give the sampled batch a huge priority, most of the next batch must come from it
    ids = rep.last_ids(B).cpu().numpy()
    rep.update_priority(torch.full((B,), 1e6, device=DEV))
    rep.sample(B)
    ids2 = rep.last_ids(B).cpu().numpy()
    rep.update_priority(torch.full((B,), 1.0, device=DEV))
    assert np.isin(ids2, ids).mean() > 0.9
    rep.check_errors()
    rep.close()
