"""CPU restatement (numpy) of the volume-touching steps of Solver.clustering (reference utils/modeler.py:762-858).

TEST INFRASTRUCTURE - never imported by the product path.

PARITY UNPINNED against the reference module: utils/modeler.py imports open3d, mrcfile, superpose3d and Bio at module
top (none installed), so Solver cannot be instantiated here and the reference holds no fixture for these steps.  The
functions below are the reference's own numpy statements, copied as arithmetic (same calls, same order, same dtypes), so
numpy itself is the witness for their floating-point behaviour.
"""
from __future__ import annotations

import numpy as np


def threshold_points(ca_prob, thr):
    """:767   pcd_numpy = np.array(np.where(self.CAProb > thr)).T"""
    return np.array(np.where(ca_prob > thr)).T


def gather(vol, pts):
    """:780, :786, :800   vol[pcd[:, 0], pcd[:, 1], pcd[:, 2]]"""
    return vol[pts[:, 0], pts[:, 1], pts[:, 2]]


def refine_candidates(ca_prob, aa_prob, ca_cands):
    """:834-858.  Returns (new_cands float64 [m,3], new_AAs float32 [m,20], kept indices into ca_cands): candidates whose
    neighbourhood leaves the volume raise inside the reference's try block and are skipped."""
    new_cands, new_AAs, kept = [], [], []
    for idx, cand in enumerate(ca_cands):
        try:
            coord = [0, 0, 0]
            AA_list = []
            cand = np.array(cand)
            with np.errstate(all="ignore"):
                weights = ca_prob[cand[0]-1:cand[0]+2, cand[1]-1:cand[1]+2, cand[2]-1:cand[2]+2] / \
                    np.sum(ca_prob[cand[0]-1:cand[0]+2, cand[1]-1:cand[1]+2, cand[2]-1:cand[2]+2])
                for di in [-1, 0, 1]:
                    for dj in [-1, 0, 1]:
                        for dk in [-1, 0, 1]:
                            this_coord = cand + [di, dj, dk]
                            coord += this_coord * weights[di+1, dj+1, dk+1]
                            AA_list.append(aa_prob[:, this_coord[0], this_coord[1], this_coord[2]] *
                                           weights[di+1, dj+1, dk+1])
            new_cands.append(coord)
            new_AAs.append(np.sum(AA_list, axis=0))
            kept.append(idx)
        except Exception:
            pass
    return np.array(new_cands), np.array(new_AAs), np.array(kept, dtype=np.int64)
