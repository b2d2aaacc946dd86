"""Atoms of the docked AF3 model -> the 24-channel binary encoding volume, on the GPU (`mica_rasterise_atoms`).

Mirrors the atom loop of DataPreprocessor.create_AF3_encodings (reference utils/preprocessing.py:225-347): channel
order ['CA','N','C','O'] + 20 amino acids (:254-263); every atom of a residue whose hetero flag is blank marks its
residue's amino-acid channel, backbone atoms also their own (:283-298).

The reference parses the PDB with Bio.PDB.PDBParser (:52, :269); Bio is not installed here, so `read_pdb_atoms` is a
fixed-column reader of the records PDBIO writes (the file is produced by the reference's own dock_in_map.py:698):
ATOM records of every model (counted as Bio counts them: one per MODEL record, implicit ones outside), coordinates kept as float32 like Bio's; HETATM records carry a non-blank hetero flag and
are skipped like :277 does; of alternate locations the highest occupancy wins (what iterating a Bio residue yields);
an atom name repeated with the same altloc is ignored after its first occurrence (Bio's PERMISSIVE behaviour).
"""
from __future__ import annotations

import numpy as np
import torch

BACKBONE_ATOMS = ['CA', 'N', 'C', 'O']
AMINO_ACIDS = ['ALA', 'CYS', 'ASP', 'GLU', 'PHE', 'GLY', 'HIS', 'ILE', 'LYS', 'LEU',
               'MET', 'ASN', 'PRO', 'GLN', 'ARG', 'SER', 'THR', 'VAL', 'TRP', 'TYR']
CHANNEL_NAMES = BACKBONE_ATOMS + AMINO_ACIDS


def read_pdb_atoms(path: str):
    """-> (coords float32[n,3], atom_names list[str], res_names list[str]) for the atoms create_AF3_encodings visits."""
    chosen = {}          # (model, chain, resseq, icode, atom name) -> [occupancy, altloc, coord, resname]
    order = []
    model, model_open = -1, False
    with open(path, "r") as f:
        for line in f:
            rec = line[0:6]
            # models as Bio's parser counts them: every MODEL record opens a new one whatever serial it carries, ENDMDL closes it, and
            # an atom outside any model opens one implicitly - atoms of different models never replace each other
            if rec == "MODEL ":
                model += 1
                model_open = True
                continue
            if rec == "ENDMDL":
                model_open = False
                continue
            if rec != "ATOM  ":
                continue                                   # HETATM: hetero flag 'H_xxx' / 'W' -> not ' ' (:277)
            if not model_open:
                model += 1
                model_open = True
            name = line[12:16].strip()
            altloc = line[16]
            resname = line[17:20].strip()
            key = (model, line[21], int(line[22:26]), line[26], name)
            coord = np.array((float(line[30:38]), float(line[38:46]), float(line[46:54])), dtype=np.float32)
            try:
                occ = float(line[54:60])
            except ValueError:
                occ = 1.0
            if key not in chosen:
                chosen[key] = [occ, altloc, coord, resname]
                order.append(key)
            elif altloc != chosen[key][1] and altloc != " " and occ > chosen[key][0]:
                chosen[key] = [occ, altloc, coord, chosen[key][3]]
    coords = np.zeros((len(order), 3), dtype=np.float32)
    names, resnames = [], []
    for i, k in enumerate(order):
        coords[i] = chosen[k][2]
        names.append(k[4])
        resnames.append(chosen[k][3])
    return coords, names, resnames


def channel_indices(atom_names, res_names):
    """Per atom: backbone channel 0..3 or -1, amino-acid channel 4..23 or -1 (get_aa_channel_index, :180-185)."""
    bb = np.array([BACKBONE_ATOMS.index(a) if a in BACKBONE_ATOMS else -1 for a in atom_names], dtype=np.int32)
    aa = np.array([4 + AMINO_ACIDS.index(r) if r in AMINO_ACIDS else -1 for r in res_names], dtype=np.int32)
    return bb, aa


def rasterise(engine, coords, atom_names, res_names, origin, shape) -> torch.Tensor:
    """-> float32 [24, nz, ny, nx] on the engine's device."""
    bb, aa = channel_indices(atom_names, res_names)
    dev = engine.device
    xyz = torch.from_numpy(np.ascontiguousarray(coords, dtype=np.float32).reshape(-1, 3)).to(dev)
    return engine.rasterise_atoms(xyz, torch.from_numpy(bb).to(dev), torch.from_numpy(aa).to(dev), origin, shape)
