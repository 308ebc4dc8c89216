"""Which chain stream a call's range goes to, from the sequence of calls alone (plan_turn in icsp_sched.cpp, run on the CPU through the
test hook icsp_debug_plan_turns): the properties the resident regimes rest on."""
import ctypes as C

import numpy as np

from icspcodec_amd import capi


def plan(period, calls):
    lib = capi.load()
    k = len(calls)
    f = (C.c_int * k)(*[c[0] for c in calls])
    n = (C.c_int * k)(*[c[1] for c in calls])
    w, t, s = (C.c_int * k)(), (C.c_int * k)(), (C.c_int * k)()
    assert lib.icsp_debug_plan_turns(period, k, f, n, w, t, s) == 0
    return list(w), list(t), list(s)


def test_two_ranges_alternating_take_two_streams_in_turn():
    for period in (0, 10):
        calls = [(0, 300), (300, 300)] * 6
        whole, three, turn = plan(period, calls)
        assert whole == [0] + [1] * 11                      # the first call has no predecessor to be disjoint from
        assert not any(three)
        # each range keeps its stream: it never has to wait for its own previous pass on another one
        assert len({t for (c, t, w) in zip(calls, turn, whole) if w and c[0] == 0}) == 1
        assert len({t for (c, t, w) in zip(calls, turn, whole) if w and c[0] == 300}) == 1
        assert set(turn[1:]) == {0, 1}


def test_three_all_intra_ranges_in_rotation_take_three_streams():
    calls = [(0, 300), (300, 300), (600, 300)] * 5
    whole, three, turn = plan(0, calls)
    assert whole[1:] == [1] * 14 and three[:2] == [0, 0] and all(three[2:])
    assert set(turn[2:]) == {0, 1, 2}
    # from the second round on every range comes back on the stream of its previous pass
    for r in range(3):
        assert len({turn[i] for i in range(3 + r, 15, 3)}) == 1
    # four ranges over three streams: all three stay in use, ranges move
    calls = [(150 * (k % 4), 150) for k in range(16)]
    whole, three, turn = plan(0, calls)
    assert all(three[2:]) and set(turn[2:]) == {0, 1, 2}


def test_ippp_ranges_never_take_a_third_chain_stream():
    calls = [(0, 300), (300, 300), (600, 300)] * 4
    whole, three, turn = plan(10, calls)
    assert whole[1:] == [1] * 11 and not any(three) and set(turn) <= {0, 1}


def test_the_same_range_again_is_not_placed_whole_and_overlaps_break_a_rotation():
    whole, three, turn = plan(0, [(0, 300)] * 4)
    assert not any(whole) and not any(three)
    # A, B, then a range overlapping A: whole against B, but no three-way rotation; back to two streams
    whole, three, turn = plan(0, [(0, 300), (300, 300), (100, 100), (300, 300), (0, 300), (600, 300)])
    assert whole == [0, 1, 1, 1, 1, 1]
    assert three == [0, 0, 0, 0, 0, 1]                      # (0,300), (300,300) ... only the last three calls are pairwise disjoint
    assert all(t in (0, 1) for t in turn[:5])


def test_random_sequences_keep_the_turn_inside_the_streams_that_exist():
    rng = np.random.default_rng(5)
    for period in (0, 5):
        L = max(period, 1)
        calls = [(int(rng.integers(0, 40)) * L * 4, int(rng.integers(1, 5)) * L * 4) for _ in range(400)]
        whole, three, turn = plan(period, calls)
        for w, t3, t in zip(whole, three, turn):
            assert t in ((0, 1, 2) if t3 else (0, 1))
            assert w or (t == 0 and not t3)
        assert period == 0 or not any(three)
