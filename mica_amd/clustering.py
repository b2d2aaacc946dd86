"""Device-side helpers for the first consumer of the volumes, Solver.clustering (reference utils/modeler.py:762-899).

Every step that touches whole volumes or is quadratic in the candidates runs on the GPU, so that the volumes - 12.3 GB at
512^3, 10.7 GB of it the amino-acid probabilities - stay in HBM and only point lists travel.  DBSCAN (open3d, :770) stays
with the caller, and so do the scalar decisions on per-cluster numbers and the sort of the candidate list (host numpy on
1e2-1e5 values, the reference's own statements).

    pts, ca, bb = candidate_points(eng, vols, thr)              # :767 pcd_numpy ; CAProb and BBProb at those points
    labels = dbscan(pts)                                         # caller (open3d)
    sums, avgs, val_mat = cluster_scores(eng, bb, labels)        # :775-797
    CA_cands = nms(eng, ca, pts, val_mat, vol_shape, thr, r)     # :799-831
    new_cands, new_AAs, kept = refine(eng, vols, CA_cands)       # :834-858
    dis, lists, neigh_mat = neighbours(eng, vols, new_cands)     # :860-888
"""
from __future__ import annotations

import numpy as np
import torch


def candidate_points(engine, volumes: dict, thr: float):
    """volumes: the dict VolumePredictor.predict_volume returns (device tensors).  -> (pcd_numpy int64 [n,3] in np.where
    order, CAProb float32 [n], BBProb float32 [n]) as host arrays."""
    ca = volumes["carbon_alpha_probability"]
    bb = volumes["backbone_probability"]
    n1, n2 = ca.shape[1], ca.shape[2]
    idx = engine.threshold_points(ca, thr)
    cav = engine.gather_values(ca, idx).cpu().numpy()
    bbv = engine.gather_values(bb, idx).cpu().numpy()
    lin = idx.cpu().numpy()
    pts = np.stack([lin // (n1 * n2), lin // n2 % n1, lin % n2], axis=1)
    return pts, cav, bbv


def refine(engine, volumes: dict, ca_cands):
    """modeler.py:834-858 for a list of integer candidate positions.  -> (new_cands float64 [m,3], new_AAs float32 [m,20],
    kept int64 [m]): candidates on the volume's faces are skipped as the reference does ('found at boundary')."""
    ca = volumes["carbon_alpha_probability"]
    aa = volumes["amino_acid_probability"]
    c = torch.as_tensor(np.asarray(ca_cands, dtype=np.int32).reshape(-1, 3)).to(ca.device)
    coord, aao, ok = engine.refine_candidates(ca, aa.contiguous(), c)
    ok = ok.cpu().numpy()
    return coord.cpu().numpy()[ok], aao.cpu().numpy()[ok], np.nonzero(ok)[0]


def gather_at(engine, vol: torch.Tensor, pts):
    """vol[pts[:,0], pts[:,1], pts[:,2]] (:856 AAPred at rounded candidates, :884 BBProb along candidate pairs)."""
    pts = np.asarray(pts, dtype=np.int64).reshape(-1, 3)
    n1, n2 = vol.shape[-2], vol.shape[-1]
    lin = torch.as_tensor((pts[:, 0] * n1 + pts[:, 1]) * n2 + pts[:, 2]).to(vol.device)
    return engine.gather_values(vol, lin).cpu().numpy()


def cluster_scores(engine, bb_at_points, labels):
    """modeler.py:775-797.  bb_at_points: BBProb at pcd_numpy (float32 [n], host or device), labels: DBSCAN labels (int [n],
    -1 = noise).  The per-cluster sums run on the GPU in numpy's float32 summation order (mica_segment_sums), so the
    reference's comparisons on sums and means fall the same way.  -> (labels_scores_sum, labels_scores_avg, val_mat)."""
    labels = np.asarray(labels)
    nlab = int(labels.max()) + 1 if labels.size else 0
    if nlab <= 0:
        return [], [], np.zeros(labels.shape, dtype=bool)
    order = np.argsort(labels, kind="stable")                     # points of one cluster stay in np.where order
    order = order[labels[order] >= 0]
    counts = np.bincount(labels[order], minlength=nlab)
    off = np.zeros(nlab + 1, dtype=np.int64)
    np.cumsum(counts, out=off[1:])
    vals = torch.as_tensor(bb_at_points).to(engine.device, torch.float32)
    sums = engine.segment_sums(vals[torch.as_tensor(order).to(engine.device)].contiguous(), torch.as_tensor(off).to(engine.device)).cpu().numpy()
    labels_scores_sum = [np.float32(v) for v in sums]
    labels_scores_avg = []
    for label in range(nlab):                                     # :783-789, np.mean = float32 sum / count
        if labels_scores_sum[label] > np.max(labels_scores_sum) / 10:
            labels_scores_avg.append(np.float32(labels_scores_sum[label] / np.float32(counts[label])))
        else:
            labels_scores_avg.append(0)
    val_mat = np.zeros_like(labels).astype(bool)                  # :791-796
    max_labels_score = np.max(labels_scores_avg)
    for label in range(nlab):
        if labels_scores_avg[label] > max_labels_score / 2:
            val_mat[np.where(labels == label)] = True
    return labels_scores_sum, labels_scores_avg, val_mat


def nms(engine, ca_at_points, pts, val_mat, vol_shape, thr: float, nms_radius: float):
    """modeler.py:799-831.  ca_at_points: CAProb at pcd_numpy (host float32 [n]), pts: pcd_numpy.  The candidate list is
    sorted on the host exactly as the reference sorts pred_list (np.argsort of the negated float64 scores); the greedy
    suppression runs on the GPU.  -> CA_cands as an int array [m,3] in the reference's order."""
    idx = np.where(val_mat)[0]
    score = np.asarray(ca_at_points)[idx].astype(np.float64)      # pred_list is a float64 array (:816)
    order = np.argsort(-score, axis=0)
    score, p = score[order], np.asarray(pts)[idx][order]
    n = int(np.searchsorted(-score, -float(thr), side="right"))  # the loop stops at the first score < thr (:824)
    if n == 0:
        return np.zeros((0, 3), dtype=np.int64)
    d = torch.as_tensor(np.ascontiguousarray(p[:n], dtype=np.int32)).to(engine.device)
    keep = engine.nms_points(d, vol_shape, nms_radius).cpu().numpy()
    return p[:n][keep].astype(np.int64)


def neighbours(engine, volumes: dict, ca_cands, numpy_legacy: bool = False):
    """modeler.py:860-888.  ca_cands float64 [n,3] (refined positions).  -> (cand_self_dis, (neighbors2to6, neighbors0to6,
    neighbors0to7, neighbors2to7), neigh_mat) as host arrays / lists of index arrays.  numpy_legacy=True reproduces the
    reference's pinned numpy 1.19.1 (float64 density sums) instead of numpy 2's NEP 50 promotion."""
    bb = volumes["backbone_probability"]
    c = torch.as_tensor(np.ascontiguousarray(ca_cands, dtype=np.float64).reshape(-1, 3)).to(bb.device)
    dis, mat = engine.neighbour_matrix(c, bb, numpy_legacy)
    dis, mat = dis.cpu().numpy(), mat.cpu().numpy()
    n = dis.shape[0]
    lists = ([np.where((dis[i] <= 6) * (dis[i] >= 2))[0] for i in range(n)], [np.where(dis[i] <= 6)[0] for i in range(n)],
             [np.where(dis[i] <= 7)[0] for i in range(n)], [np.where((dis[i] <= 7) * (dis[i] >= 2))[0] for i in range(n)])
    return dis, lists, mat
