"""The ticket order of the one-launch Cholesky factorisation of the global bundle adjustment (csrc/ba_factor.hip): the launch
cannot deadlock because every work item only waits for items with SMALLER tickets, or for stages of the chain workgroup that in
turn wait only for such items -- the lowest unfinished ticket is then always held by a running workgroup whose inputs arrive,
whatever the number of resident workgroups.  Checked here on the library's own decoding of a ticket (cdv_ba_factor_ticket, a
host function: no GPU), for every matrix size the path serves (33 .. 1024 free poses: 4 .. 96 block columns)."""
import ctypes

import pytest

from cdv_slam_amd import _lib


def _items(nb):
    lib = _lib.load()
    out = (ctypes.c_int32 * 3)()
    total = lib.cdv_ba_factor_ticket(0, nb, out)
    items = []
    for t in range(total):
        assert lib.cdv_ba_factor_ticket(t, nb, out) == total
        items.append((out[0], out[1], out[2]))
    return items


@pytest.mark.parametrize("nb", [4, 5, 6, 7, 12, 29, 57, 96])
def test_every_item_waits_only_for_smaller_tickets(nb):
    items = _items(nb)
    assert items[0][0] == 0                                   # ticket 0: the chain workgroup
    pos_block, pos_p = {}, {}
    for t, (kind, c, r) in enumerate(items[1:], start=1):
        if kind == 1:
            assert (r, c) not in pos_block and 0 <= c < nb and (c + 2 <= r < nb or r == nb)
            pos_block[(r, c)] = t
        else:
            assert kind == 2 and r == c and 3 <= c < nb and c not in pos_p
            pos_p[c] = t
    # complete: every block two or more below the diagonal, every right-hand-side block, the P item of every block row >= 3
    assert set(pos_block) == {(r, c) for c in range(nb) for r in list(range(c + 2, nb)) + [nb]}
    assert set(pos_p) == set(range(3, nb))
    # chain stage s (factor (s, s), solve (s + 1, s)): the largest ticket it waits for, directly or through the stages before it
    need = []
    for s in range(nb):
        t = need[s - 1] if s else 0
        if s >= 3:
            t = max(t, pos_p[s])                              # its diagonal block, pre-accumulated
        if s + 1 < nb:
            if s + 1 >= 3:
                t = max(t, pos_p[s + 1])                      # its neighbour block, pre-accumulated
            if s >= 1:
                t = max(t, pos_block[(s + 1, s - 1)])         # ... and the one product that is applied by the chain's helpers
        need.append(t)
    for (r, c), t in pos_block.items():
        deps = [pos_block[(r, k)] for k in range(c)]                          # L(r, k), k < c
        deps += [pos_block[(c, k)] for k in range(c - 1)]                     # L(c, k), k <= c - 2
        deps += [need[c]] + ([need[c - 1]] if c else [])                      # L(c, c), L(c, c - 1): the chain's stages
        assert all(d < t for d in deps), (nb, r, c)
    for c, t in pos_p.items():
        deps = [pos_block[(c, k)] for k in range(c - 2)] + [pos_block[(c - 1, k)] for k in range(c - 2)]
        assert all(d < t for d in deps), (nb, c)
