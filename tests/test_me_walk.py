"""k_me<false>'s compile-time model of state 0's search walk (icsp_me.hip.inc: me_cand0, me_mv_of_step0, me_sad_state0) restated on the
CPU and checked for what the kernel's correctness rests on: the closed form is the reference's walk (motionEstimation, ENC:2111-2125),
the byte shift of step 16*cs + j does not depend on cs, its window address is one of four per-lane bases plus a non-negative
constant, every candidate row stays inside the staged 48x48 window, and the reduce-scatter's lane / register bookkeeping leaves the
SAD of step l in lane l."""
import numpy as np

K_WIN_DW = 18


def reference_walk():
    """(x0, y0) of the 64 steps from the start state (flag, xflag, yflag) = (0, +1, -1): ENC:2095, 2111-2125"""
    flag, xflag, yflag = 0, 1, -1
    x0 = y0 = xcnt = ycnt = 0
    out = []
    for _ in range(64):
        if not flag:
            x0 = x0 + xcnt if xflag <= 0 else x0 - xcnt
            flag = 1; xcnt += 1; xflag = -xflag
        else:
            y0 = y0 + ycnt if yflag < 0 else y0 - ycnt
            flag = 0; ycnt += 1; yflag = -yflag
        out.append((x0, y0))
    return out


def me_cand0(j):
    m0 = j // 2 + 1
    m_even = (m0 & 1) == 0
    k = m0 if (j & 1) else m0 - 1
    return (m0 // 2 if m_even else -((m0 - 1) // 2), (k - 1) // 2 if (k & 1) else -(k // 2), 1 if m_even else -1, 1 if (k & 1) else -1)


def me_combo(j):
    _, _, sx, sy = me_cand0(j)
    return (2 if sx > 0 else 0) + (1 if sy > 0 else 0)


def me_off(j):
    dx0, dy0, _, _ = me_cand0(j)
    return dy0 * K_WIN_DW + ((16 + dx0) >> 2) - 4


def me_combo_min(c):
    return min(me_off(j) for j in range(16) if me_combo(j) == c)


def me_mv_of_step0(s):
    m = (s >> 1) + 1
    k = m if (s & 1) else m - 1
    return ((m - 1) >> 1) if (m & 1) else -(m >> 1), -((k - 1) >> 1) if (k & 1) else (k >> 1)


def test_closed_form_is_the_reference_walk():
    walk = reference_walk()
    assert len(set(walk)) == 63 and walk[0] == walk[1] == (0, 0)       # the second zero-SAD candidate of ENC:2136-2141 can be step 1
    for l in range(64):
        cs, j = l >> 4, l & 15
        dx0, dy0, sx, sy = me_cand0(j)
        assert (dx0 + 4 * cs * sx, dy0 + 4 * cs * sy) == walk[l]
        assert me_mv_of_step0(l) == (-walk[l][0], -walk[l][1])
        assert (16 + walk[l][0]) & 3 == (16 + dx0) & 3                    # the byte shift is the group's, a constant of the instruction


def test_addresses_are_base_plus_nonnegative_constant_and_stay_inside_the_window():
    walk = reference_walk()
    for c in range(4):
        assert all(me_off(j) - me_combo_min(c) >= 0 for j in range(16) if me_combo(j) == c)
    for r in range(16):
        for cs in range(4):
            centre = (16 + r) * K_WIN_DW + 4
            for j in range(16):
                c = me_combo(j)
                sx, sy = (1 if c & 2 else -1), (1 if c & 1 else -1)
                base = centre + cs * (sy * 4 * K_WIN_DW + sx) + me_combo_min(c)
                assert base >= 0
                first = base + me_off(j) - me_combo_min(c)
                dx, dy = walk[16 * cs + j]
                assert first == (16 + dy + r) * K_WIN_DW + ((16 + dx) >> 2)
                ndw = 4 if (16 + dx) & 3 == 0 else 5
                row, col0 = divmod(first, K_WIN_DW)
                assert 0 <= row < 48 and col0 + ndw <= 13                 # 48 rows; dwords 0..12 of the macroblock's part (12: the next 4 bytes, shift > 0 only)
                assert (16 + dx) + 15 <= 47 and 0 <= 16 + dx


def _dpp_rows(v, perm):
    """v: [64] values; perm(i) -> source lane within the DPP row of 16"""
    out = np.empty_like(v)
    for l in range(64):
        out[l] = v[(l & ~15) | perm(l & 15)]
    return out


def test_reduce_scatter_leaves_step_l_in_lane_l():
    rng = np.random.default_rng(5)
    sd = rng.integers(0, 4081, size=(64, 16)).astype(np.uint32)          # [lane][candidate j]: a row's SAD
    P = [(sd[:, 2 * k] | (sd[:, 2 * k + 1] << 16)).astype(np.uint32) for k in range(8)]
    lane = np.arange(64)
    r = lane & 15
    bank = r >> 2
    Q = []
    for k in range(4):                                                    # row_ror:8 pairs lane i with i ^ 8 whatever the direction
        q = P[k] + _dpp_rows(P[k], lambda i: i ^ 8)
        hi = P[k + 4] + _dpp_rows(P[k + 4], lambda i: i ^ 8)
        Q.append(np.where(bank >= 2, hi, q))                             # bank_mask 0xc
    R = []
    for k in range(2):                                                    # row_half_mirror: i <-> 7 - i inside each half of the row
        q = Q[k] + _dpp_rows(Q[k], lambda i: (i & 8) | (7 - (i & 7)))
        hi = Q[k + 2] + _dpp_rows(Q[k + 2], lambda i: (i & 8) | (7 - (i & 7)))
        R.append(np.where((bank & 1) == 1, hi, q))                       # bank_mask 0xa
    a = R[0] + _dpp_rows(R[0], lambda i: i ^ 2)
    b = R[1] + _dpp_rows(R[1], lambda i: i ^ 2)
    S = np.where((r & 2) != 0, b, a)
    T = S + _dpp_rows(S, lambda i: i ^ 1)
    got = (T >> ((r & 1) * 16)) & 0xffff
    want = np.array([sd[(l & ~15):(l & ~15) + 16, l & 15].sum() for l in range(64)], np.uint32)
    assert np.array_equal(got, want)
