"""Tile-file reader mirroring CryoEMTestDataset (reference dataset/dataset.py:179-224) for callers
that still hand tiles over as .npz files (the on-disk "wire" format of SURVEY.md section 8b)."""
from __future__ import annotations

import numpy as np

AF3_TYPES = ['CA', 'N', 'C', 'O', 'ALA', 'CYS', 'ASP', 'GLU', 'PHE', 'GLY', 'HIS', 'ILE', 'LYS', 'LEU', 'MET',
             'ASN', 'PRO', 'GLN', 'ARG', 'SER', 'THR', 'VAL', 'TRP', 'TYR']      # dataset.py:184-188


class CryoEMTestDataset:
    def __init__(self, data_dir, transform=None):
        self.data_dir = list(data_dir)
        self.transform = transform
        self.types = list(AF3_TYPES)

    def __len__(self):
        return len(self.data_dir)

    def __getitem__(self, idx):
        """-> (map f32[1,W,W,W], af f32[24,W,W,W], metadata) as numpy arrays; any failure loading the
        24 AF3 tiles yields zeros, like the reference (dataset.py:209-219)."""
        map_path = self.data_dir[idx]
        data = np.load(map_path)
        grid = data['grid']
        metadata = {k: data[k] for k in ('i', 'j', 'k', 'di', 'dj', 'dk', 'orig_shape')}
        metadata['filename'] = map_path.split("/")[-1].split(".")[0]
        try:
            feats = []
            for t in self.types:
                p = map_path.replace('normalized_map_grids', f"AF3_encoding_grids/{t}_grids")
                p = p.replace('normalized_map', f"{t}")
                feats.append(np.load(p)['grid'])
            af = np.stack(feats, axis=0)
        except Exception:
            af = np.zeros((24,) + grid.shape)
        return grid[None].astype(np.float32), af.astype(np.float32), metadata
