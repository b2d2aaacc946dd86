"""CPU restatement (numpy) of the atom loop of DataPreprocessor.create_AF3_encodings
(reference utils/preprocessing.py:172-186 and 254-298).

TEST INFRASTRUCTURE - never imported by the product path.

PINNED (round 3): oracle/gen_golden_r3.py runs the reference's own DataPreprocessor.create_AF3_encodings (utils/preprocessing.py,
imported unmodified under I/O-only adapters for `mrcfile` and `Bio.PDB`) on five synthetic atom lists and asserts that
rasterise_atoms() equals the 24 channels it writes bit for bit, including the two non-cubic cases where its scatter raises IndexError
(tests/golden/af3_ref.json).  The arithmetic below is the reference's own five numpy calls (subtract, divide by 1.0, np.round,
astype(int), np.clip) applied in the reference's order; the channel tables are copied as data from :254-260.
"""
from __future__ import annotations

import numpy as np

BACKBONE_ATOMS = ['CA', 'N', 'C', 'O']                                                   # :254
AMINO_ACIDS = ['ALA', 'CYS', 'ASP', 'GLU', 'PHE', 'GLY', 'HIS', 'ILE', 'LYS', 'LEU',     # :255-260
               'MET', 'ASN', 'PRO', 'GLN', 'ARG', 'SER', 'THR', 'VAL', 'TRP', 'TYR']
CHANNEL_NAMES = BACKBONE_ATOMS + AMINO_ACIDS                                             # :263


def transform_coordinates(coord, origin, shape):
    """:172-178.  coord float32[3] (Bio.PDB atom.get_coord()), origin = header origin (x, y, z) as float32 record fields,
    shape = the map array's shape (nz, ny, nx) - applied to (x, y, z) in that order, as the reference does."""
    coord_shifted = coord - np.array((origin[0], origin[1], origin[2]))
    indices = coord_shifted / 1.0
    indices = np.round(indices).astype(int)
    indices = np.clip(indices, 0, np.array(shape) - 1)
    return indices


def get_aa_channel_index(residue_name):
    """:180-185"""
    try:
        return len(BACKBONE_ATOMS) + AMINO_ACIDS.index(residue_name)
    except ValueError:
        return -1


def rasterise_atoms(coords, atom_names, res_names, origin, shape):
    """:268-298 for the atoms of standard (hetero flag ' ') residues, in file order.  Returns float32 [24, *shape] (the
    reference keeps float64 in RAM and casts to float32 when it writes each channel, :196).  Raises IndexError exactly
    where the reference's volume[ch, idx[2], idx[1], idx[0]] would (non-cubic maps)."""
    origin = np.asarray(origin, dtype=np.float32)
    feature_volume = np.zeros((len(CHANNEL_NAMES), *shape))
    for coord, name, res in zip(np.asarray(coords, dtype=np.float32), atom_names, res_names):
        aa_idx = get_aa_channel_index(res)
        with np.errstate(invalid="ignore"):
            idx = transform_coordinates(coord, origin, shape)
        if name in BACKBONE_ATOMS:
            feature_volume[BACKBONE_ATOMS.index(name), idx[2], idx[1], idx[0]] = 1.0
        if aa_idx >= 0:
            feature_volume[aa_idx, idx[2], idx[1], idx[0]] = 1.0
    return feature_volume.astype(np.float32)
