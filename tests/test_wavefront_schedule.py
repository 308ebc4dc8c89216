"""The block wavefront of the intra luma kernels as a model on the CPU: which (step, wave, lane group) takes which 8x8 block, for the
plain wavefront t = c8 + 2 r8 and for block rows chained in groups of GC (t = c8 + r8 + r8 / GC), with the slots of a step rotating
over the waves.  The formulas are those of k_intra_luma8<NW, RING, GC> (icspcodec_amd/csrc/icsp_blk8.hip.inc) and of
intra_slots / intra_waves_chained (icsp_sched.cpp) restated; the test checks the properties the kernel's correctness rests on,
for the geometries of the GPU parity suite:

  * every block is taken exactly once, by an active lane group, inside the frame;
  * left, up and up-left neighbours (pixels, mode predictor, ENC:665-728, 1334-1350) were taken in an EARLIER step;
  * the upper-right neighbour, whose reconstructed DC enters the DC predictor's median (ENC:3760-3818), was taken in an earlier
    step -- or, in a chained group, in the SAME step by the lane group one below in the SAME wave (DcLink), after at most GC - 1
    rounds of the chain;
  * the reconstruction ring: the rows live in one step map to distinct ring slots, and a row keeps its slot while it is live;
  * the neighbour records (MR rolling rows): a record is not overwritten before its last reader has run;
  * a step fits one round of the workgroup that launch_intra8_g would pick.
No GPU, no library: host logic only."""
import pytest


def steps_of(cols8, rows8, gc):
    return cols8 + (rows8 - 1) + (rows8 - 1) // gc if gc else cols8 + 2 * (rows8 - 1)


def step_shape(t, cols8, rows8, gc):
    """(r_lo, nact, r_first) of step t: slot q <-> row r_lo + q, of which the first r_first - r_lo are idle."""
    if gc:
        g1 = gc + 1
        tp = t - (cols8 - 1)
        r_first = 0 if tp <= 0 else gc * (tp // g1) + min(tp % g1, gc)
        r_last = min(rows8 - 1, gc * (t // g1) + min(t % g1, gc - 1))
        r_lo = r_first & ~(gc - 1)
        return r_lo, r_last - r_lo + 1, r_first
    r_lo = t - (cols8 - 1)
    r_lo = 0 if r_lo <= 0 else (r_lo + 1) >> 1
    return r_lo, min(rows8 - 1, t >> 1) - r_lo + 1, r_lo


def col_of(t, r8, gc):
    return t - (r8 + r8 // gc) if gc else t - 2 * r8


def waves_needed(cols8, rows8, gc):
    widest = max(step_shape(t, cols8, rows8, gc)[1] for t in range(steps_of(cols8, rows8, gc)))
    return (widest + 7) // 8


def built_variant(nw):
    for v in (1, 2, 3, 4, 5, 6, 8):
        if nw <= v:
            return v
    return None


def schedule(cols8, rows8, gc, nw):
    """{(r8, c8): (t, wave, jb)} plus per-step lists, as the kernel's loops produce them."""
    taken, per_step = {}, []
    for t in range(steps_of(cols8, rows8, gc)):
        r_lo, nact, r_first = step_shape(t, cols8, rows8, gc)
        blocks = []
        for wave in range(nw):
            wrole = (wave + t) % nw
            q0 = wrole * 8
            rounds = 0
            while q0 < nact:
                rounds += 1
                for jb in range(8):
                    rq = r_lo + q0 + jb
                    active = (q0 + jb) < nact and rq >= r_first
                    r8 = max(min(rq, r_lo + nact - 1), r_first) if gc else r_lo + min(q0 + jb, nact - 1)
                    c8 = col_of(t, r8, gc)
                    assert 0 <= r8 < rows8 and 0 <= c8 < cols8, "shadow lanes address a block of the frame"
                    if active:
                        assert (r8, c8) not in taken, f"block {(r8, c8)} taken twice"
                        taken[(r8, c8)] = (t, wave, jb)
                        blocks.append((r8, c8, wave, jb))
                q0 += nw * 8
            assert rounds <= 1, "one round per step (a row is live in one ring slot)"
        per_step.append(blocks)
    return taken, per_step


GEOMS = [(44, 36), (44, 72), (88, 72), (8, 6), (4, 2), (2, 2), (2, 12), (6, 8), (20, 4), (60, 2), (4, 36), (30, 34)]


@pytest.mark.parametrize("cols8,rows8", GEOMS)
@pytest.mark.parametrize("gc", [0, 2])
def test_every_block_once_and_neighbours_first(cols8, rows8, gc):
    nw = built_variant(waves_needed(cols8, rows8, gc))
    if nw is None:
        pytest.skip("wider than the eight-wave workgroup of the ring forms")
    taken, per_step = schedule(cols8, rows8, gc, nw)
    assert len(taken) == cols8 * rows8
    for (r8, c8), (t, wave, jb) in taken.items():
        for dr, dc in ((0, -1), (-1, 0), (-1, -1)):
            if r8 + dr >= 0 and c8 + dc >= 0:
                assert taken[(r8 + dr, c8 + dc)][0] < t
        lul = ((r8 & 1) and (c8 & 1)) or c8 == cols8 - 1                      # the median takes up-left instead (ENC:3760-3818)
        if r8 > 0 and c8 > 0 and not lul:
            tu, wu, ju = taken[(r8 - 1, c8 + 1)]
            link = gc > 1 and (r8 & (gc - 1)) != 0
            if link:
                assert (tu, wu, ju) == (t, wave, jb - 1), "chained: the upper-right neighbour is one lane group below, same wave, same step"
            else:
                assert tu < t


@pytest.mark.parametrize("cols8,rows8", GEOMS)
@pytest.mark.parametrize("gc", [2])
def test_chain_depth(cols8, rows8, gc):
    """The DC of a chained block is final after as many rounds as blocks below it in its group: at most GC - 1."""
    nw = built_variant(waves_needed(cols8, rows8, gc))
    if nw is None:
        pytest.skip("wider than eight waves")
    taken, _ = schedule(cols8, rows8, gc, nw)
    for (r8, c8) in taken:
        depth, r, c = 0, r8, c8
        while gc > 1 and (r & (gc - 1)) != 0 and c > 0 and not (((r & 1) and (c & 1)) or c == cols8 - 1):
            r, c, depth = r - 1, c + 1, depth + 1
        assert depth <= gc - 1


@pytest.mark.parametrize("cols8,rows8", GEOMS)
@pytest.mark.parametrize("gc", [0, 2])
def test_ring_slots_and_record_rows(cols8, rows8, gc):
    nw = built_variant(waves_needed(cols8, rows8, gc))
    if nw is None:
        pytest.skip("wider than eight waves")
    taken, per_step = schedule(cols8, rows8, gc, nw)
    slots = nw * 8 if gc else next(p for p in (8, 16, 32, 64) if nw * 8 <= p)   # ring_slots_exact / ring_slots
    first = {r: min(taken[(r, c)][0] for c in range(cols8)) for r in range(rows8)}
    last = {r: max(taken[(r, c)][0] for c in range(cols8)) for r in range(rows8)}
    for r in range(rows8):
        for r2 in range(r + 1, rows8):
            if r % slots == r2 % slots:
                assert last[r] < first[r2], "two rows that share a ring slot are never live together"
    # neighbour records: row r's record of column c is written in step taken[(r, c)] and read by (r, c + 1) as left, by
    # (r + 1, c - 1 .. c + 1) as up-right / up / up-left.  Row r + MR writes the same storage row.
    mr = 4 if gc else 2
    for r in range(rows8 - mr):
        for c in range(cols8):
            readers = [taken[(r, c + 1)][0]] if c + 1 < cols8 else []
            readers += [taken[(r + 1, cc)][0] for cc in (c - 1, c, c + 1) if 0 <= cc < cols8]
            assert max(readers) < taken[(r + mr, c)][0], "a record row is reused only after its last reader"


def test_steps_and_waves_of_cif():
    """The figures DESIGN.md section 5 quotes for CIF."""
    assert steps_of(44, 36, 0) == 114 and steps_of(44, 36, 2) == 96
    assert waves_needed(44, 36, 0) == 3 and waves_needed(44, 36, 2) == 4
    for gc, tasks in ((0, 246), (2, 240)):
        nw = built_variant(waves_needed(44, 36, gc))
        _, per_step = schedule(44, 36, gc, nw)
        n = sum(len({(w) for (_, _, w, _) in blocks}) for blocks in per_step)    # wave-tasks: waves with an active block, per step
        if tasks is not None:
            assert n == tasks

