"""CPU checks of the reference-built checker (oracle/_ref): known answers and a dry run of the actor data
flow used by the GPU parity test — which also proves that flow can never hit the reference's blocking append."""
import numpy as np
import pytest

from oracle import ref_rela as R

pytestmark = [pytest.mark.timeout(120), pytest.mark.skipif(not R.available(), reason="oracle/_ref not built")]


def test_aggregate_priority_known_answer():
    # SURVEY.md §8c probe: aggregate_priority([[1,2],[3,4],[5,6]], [2,3], 0.9) = [2.9, 5.8]
    assert np.allclose(R.aggregate_priority([[1, 2], [3, 4], [5, 6]], [2, 3], 0.9), [2.9, 5.8])


def test_multistep_known_answers():
    # rela/transition_buffer.h:51-99 with n=2, gamma=0.5: return r0 + 0.5 r1, bootstrap unless a terminal is inside
    m = R.MultiStepBuffer(2, 2, 0.5, 1)
    rs = [[1, 1], [2, 2], [4, 4]]
    ts = [[0, 0], [0, 1], [0, 0]]
    for k in range(3):
        m.push_obs_action(np.full((2, 1), k, np.float32), np.array([k, k]))
        m.push_reward_terminal(np.array(rs[k], np.float32), np.array(ts[k], np.uint8))
    tr = m.pop()
    assert list(tr["reward"]) == [2.0, 2.0] and list(tr["bootstrap"]) == [1.0, 0.0]
    assert list(tr["obs"][:, 0]) == [0, 0] and list(tr["next_obs"][:, 0]) == [2, 2] and list(tr["terminal"]) == [0, 0]


def test_replay_duplicates_and_eviction():
    T, d = 4, 3
    rp = R.Replay(8, 1, 0.9, 0.6, T, d)
    def add(first, n):
        obs = np.zeros((n, T, d), np.float32)
        obs[:, 0, 0] = np.arange(first, first + n)
        z = np.zeros((n, T))
        rp.add(obs, z, z, z, z, np.full(n, T), np.linspace(0.2, 1.0, n))
    add(0, 4); add(4, 4)
    s = rp.sample(4)
    rp.update_priority([1, 1, 1, 1])
    add(8, 2)
    assert rp.size() == 10
    rp.sample(4)                      # size 10 > capacity 8: evicts the two oldest
    rp.update_priority([1, 2, 3, 4])
    assert rp.size() == 8 and rp.num_add() == 10 and rp.get(0)[0][0, 0] == 2.0


@pytest.mark.parametrize("case", [0, 1])
def test_actor_flow_dry_run_on_reference_only(case):
    from tests.test_replay_parity_gpu import FLOW_CASES, drive_actor_flow
    drive_actor_flow(*FLOW_CASES[case], with_device=False)
