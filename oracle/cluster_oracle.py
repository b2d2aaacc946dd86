"""CPU restatement (numpy) of the volume-touching steps of Solver.clustering (reference utils/modeler.py:762-858).

TEST INFRASTRUCTURE - never imported by the product path.

PINNED (round 3): oracle/gen_golden_r3.py runs the reference's own Solver.clustering (utils/modeler.py:762-899, unmodified, on a
duck-typed `self`) with scikit-learn's DBSCAN standing in for open3d's - DBSCAN is the caller's step and outside this repo's
scope; any labelling serves - and asserts that the functions below reproduce its CA_cands, CA_cands_AAProb, CA_cands_AA,
cand_self_dis, the four neighbour lists and neigh_mat bit for bit (tests/golden/cluster_ref.json).  The functions are the
reference's own numpy statements, copied as arithmetic (same calls, same order, same dtypes).  numpy here is 2.2.6; the one place
where the reference's pinned numpy 1.19.1 computes differently (value-based promotion of the density sums) is stated explicitly
by neighbour_matrix(..., numpy_legacy=True).
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


def cluster_scores(bb_prob, pcd_numpy, labels):
    """:775-797.  labels: DBSCAN output (int per point, -1 = noise).  Returns (labels_scores_sum list, labels_scores_avg
    list, val_mat bool per point) exactly as the reference builds them."""
    labels_scores_sum = []
    for label in range(labels.max()+1):
        pcd = pcd_numpy[np.where(labels == label)]
        labels_scores_sum.append(np.sum(bb_prob[pcd[:,0],pcd[:,1],pcd[:,2]]))

    labels_scores_avg = []
    for label in range(labels.max()+1):
        if labels_scores_sum[label] > np.max(labels_scores_sum)/10:
            pcd = pcd_numpy[np.where(labels == label)]
            labels_scores_avg.append(np.mean(bb_prob[pcd[:,0],pcd[:,1],pcd[:,2]]))
        else:
            labels_scores_avg.append(0)

    val_mat = np.zeros_like(labels).astype(bool)
    max_labels_score = np.max(labels_scores_avg)
    for label in range(labels.max()+1):
        if labels_scores_avg[label] > max_labels_score/2:
            val_mat[np.where(labels == label)] = True
    return labels_scores_sum, labels_scores_avg, val_mat


def sorted_pred_list(ca_prob, pcd_numpy, val_mat):
    """:799-818: CAProb_clusted, pred_list rows [score, i, j, k] (float64) sorted by descending score with numpy's
    default argsort (order among equal scores is whatever that sort gives)."""
    clustered_coords = pcd_numpy[np.where(val_mat)]
    clusted = np.zeros_like(ca_prob)
    clusted[clustered_coords[:, 0], clustered_coords[:, 1], clustered_coords[:, 2]] \
        = ca_prob[clustered_coords[:, 0], clustered_coords[:, 1], clustered_coords[:, 2]]
    pred_list = []
    indexes = np.where(val_mat)
    for i in range(indexes[0].shape[0]):
        pred_list.append([clusted[pcd_numpy[indexes[0][i]][0], pcd_numpy[indexes[0][i]][1], pcd_numpy[indexes[0][i]][2]],
                          pcd_numpy[indexes[0][i]][0], pcd_numpy[indexes[0][i]][1], pcd_numpy[indexes[0][i]][2]])
    pred_list = np.array(pred_list)
    return pred_list[np.argsort(-pred_list[:, 0], axis=0)]


def nms(pred_list, thr, nms_radius):
    """:820-831: greedy non-maximum suppression over the sorted list.  Returns CA_cands (list of [i, j, k])."""
    CA_cands = []
    while (pred_list.shape[0] > 0 and pred_list[0][0] >= thr):
        CA_cands.append([int(pred_list[0, 1]), int(pred_list[0, 2]), int(pred_list[0, 3])])
        delete_list = np.where(
            (pred_list[:, 1] - pred_list[0, 1]) ** 2 + (pred_list[:, 2] - pred_list[0, 2]) ** 2 + (
                    pred_list[:, 3] - pred_list[0, 3]) ** 2 <= nms_radius)
        pred_list = np.delete(pred_list, delete_list, 0)
    return CA_cands


def calc_dis(coordList1, coordList2):
    """:174-181."""
    y = [coordList2 for _ in coordList1]
    y = np.array(y)
    x = [coordList1 for _ in coordList2]
    x = np.array(x)
    x = x.transpose(1, 0, 2)
    a = np.linalg.norm(np.array(x) - np.array(y), axis=2)
    return a


def neighbour_matrix(ca_cands, bb_prob, numpy_legacy=False):
    """:860-888.  ca_cands float64 [n,3] (the refined positions).  Returns (cand_self_dis, the four neighbour lists,
    neigh_mat).  NOTE (parity): evaluated with the numpy of this container (2.x, NEP 50): BB_dens accumulates in float32
    and, where the distance term is a Python number, the final sum is formed in float32; under the reference's pinned
    numpy 1.19 the same statements promote to float64 (value-based casting: Python int + np.float32 -> float64), which
    numpy_legacy=True states explicitly (np.float64 operands) so that it needs no old numpy."""
    cand_self_dis = calc_dis(ca_cands, ca_cands)
    n = ca_cands.shape[0]
    neighbors2to6, neighbors0to6, neighbors0to7, neighbors2to7 = [], [], [], []
    for i in range(n):
        neighbors2to6.append(np.where((cand_self_dis[i] <= 6) * (cand_self_dis[i] >= 2))[0])
    for i in range(n):
        neighbors0to6.append(np.where(cand_self_dis[i] <= 6)[0])
    for i in range(n):
        neighbors0to7.append(np.where(cand_self_dis[i] <= 7)[0])
    for i in range(n):
        neighbors2to7.append(np.where((cand_self_dis[i] <= 7) * (cand_self_dis[i] >= 2))[0])
    neigh_mat = np.zeros_like(cand_self_dis)
    for cand in range(n):
        for neigh in neighbors2to6[cand]:
            BB_dens = np.float64(0) if numpy_legacy else 0
            dis = max(0, abs(cand_self_dis[cand, neigh] - 3.8) - 0.5)
            dis_score = max(0, 1 - dis / 2)
            for j in range(1, 5):
                coord = np.round(j/5 * ca_cands[neigh] + (5-j)/5 * ca_cands[cand]).astype(int)
                BB_dens += np.float64(bb_prob[coord[0], coord[1], coord[2]]) if numpy_legacy else bb_prob[coord[0], coord[1], coord[2]]
            neigh_mat[cand, neigh] = (dis_score + BB_dens/4) / 2
    return cand_self_dis, (neighbors2to6, neighbors0to6, neighbors0to7, neighbors2to7), neigh_mat
