"""Interaction lists for sctl_amd.ListsPlan / sctl_amd_lists_*: the near-field (U-list) pattern of a uniform tree level.

A tree code sorts its particles by leaf box and lists, for every target box, the source boxes that are its neighbours
(PVFMM's U-list; the reference reaches it through PVFMMKernelFn, fmm-wrapper.txx:756-786).  These helpers build that
structure for a uniform grid — enough for tests and for bench.py's p2p_lists workload; an adaptive tree only changes how the
same four arrays are produced."""
import numpy as np


def grid_neighbour_lists(grid, trg_counts, src_counts):
    """Lists (trg_off, trg_cnt, src_off, src_cnt) of a grid x grid x grid arrangement of boxes whose particles are stored box by box
    (box b = (ix * grid + iy) * grid + iz holds trg_counts[b] targets and src_counts[b] sources): every box interacts with
    itself and its up to 26 neighbours, lists in box order, neighbours in lexicographic order."""
    trg_counts = np.asarray(trg_counts, dtype=np.int64)
    src_counts = np.asarray(src_counts, dtype=np.int64)
    nb = grid ** 3
    assert trg_counts.size == nb and src_counts.size == nb
    t_off = np.concatenate([[0], np.cumsum(trg_counts)[:-1]])
    s_off = np.concatenate([[0], np.cumsum(src_counts)[:-1]])
    idx = np.arange(nb).reshape(grid, grid, grid)
    lt, ls = [], []
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dz in (-1, 0, 1):
                sl_t = tuple(slice(max(0, -d), grid - max(0, d)) for d in (dx, dy, dz))
                sl_s = tuple(slice(max(0, d), grid - max(0, -d)) for d in (dx, dy, dz))
                lt.append(idx[sl_t].ravel())
                ls.append(idx[sl_s].ravel())
    lt, ls = np.concatenate(lt), np.concatenate(ls)
    order = np.lexsort((ls, lt))                 # by target box, then source box
    lt, ls = lt[order], ls[order]
    return t_off[lt], trg_counts[lt], s_off[ls], src_counts[ls]


def points_in_boxes(grid, counts, rng, dtype=np.float64):
    """counts[b] points uniformly inside box b of the unit cube's grid, stored box by box (AoS, 3 values per point)."""
    counts = np.asarray(counts, dtype=np.int64)
    box = np.repeat(np.arange(grid ** 3), counts)
    corner = np.stack([box // (grid * grid), (box // grid) % grid, box % grid], 1).astype(np.float64)
    return ((corner + rng.random((box.size, 3))) / grid).astype(dtype).ravel()
