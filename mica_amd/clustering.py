"""Device-side helpers for the first consumer of the volumes, Solver.clustering (reference utils/modeler.py:762-858).

Only the steps that touch whole volumes run here (threshold + compaction, gathers, 3x3x3 refinement), so that the
volumes - 12.3 GB at 512^3, 10.7 GB of it the amino-acid probabilities - can stay in HBM and only the candidate points
(1e4-1e5) travel to the host.  DBSCAN (open3d, :770), the cluster scores (:776-797) and the greedy non-maximum
suppression (:822-831) work on the point list and stay in the caller's numpy code.

    pts, ca, bb = candidate_points(eng, vols, thr)          # :767 pcd_numpy ; CAProb and BBProb at those points
    ... labels = dbscan(pts) ; scores from bb ; NMS over (ca, pts) -> CA_cands      (reference code, unchanged)
    new_cands, new_AAs, kept = refine(eng, vols, CA_cands)  # :834-858
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
