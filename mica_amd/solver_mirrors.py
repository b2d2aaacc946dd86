"""The three mirrors configured for the reference's Solver flow - the import shim of INTEGRATION.md section 2.

`Solver.getData` / `Solver.nnPred` (reference utils/modeler.py:673-760) call, in ONE process and in this order,

    DataPreprocessor(...).resample_and_normalize_map()  [.create_AF3_encodings(pdb)]
    GridCreator(...).create_normalized_map_grids(...)   [.create_AF3_encodings_grids(...)]
    CryoEMPredictor(...).run_prediction()
    shutil.rmtree(grids) / os.remove(normalised map) / shutil.rmtree(encodings)            (:755-757)

and nothing but these three classes looks at the normalised MRC, the 24 encoding MRCs or the 25 tile files per tile in between.
The plain mirrors (mica_amd.preprocessing / create_grids / predict) keep the reference's file contract to the letter - every file
is complete when the call that is asked for it returns ("sync") - because they cannot know who reads the directory next.  Here the
flow is known, so the classes below default to writing those files BEHIND the calls: each stage hands its volume to the next on the
GPU (mica_amd/handoff.py) and `run_prediction()` joins every writer before it returns, so that all files exist - complete, under the
reference's names, keys and dtypes - by the time `nnPred` deletes them, and a `KeyboardInterrupt` / process exit joins them too.  A
file shows up under its final name only when complete (hidden temporary name + rename).

    from mica_amd.solver_mirrors import DataPreprocessor, GridCreator, CryoEMPredictor     # three lines in utils/modeler.py:14-16

MICA_GPUS=N (or `CryoEMPredictor(..., gpus=N)`) shards the tiles of every map over N GPUs of the node (mica_amd/multi.py).
"""
from __future__ import annotations

from . import create_grids as _cg
from . import predict as _pr
from . import preprocessing as _pp


class DataPreprocessor(_pp.DataPreprocessor):
    def __init__(self, map_path, AF3_results, quiet=False, engine=None, device=0, write_files="background"):
        super().__init__(map_path, AF3_results, quiet, engine=engine, device=device, write_files=write_files)
        # getData begins here (utils/modeler.py:675): with MICA_GPUS > 1 the worker ranks start NOW, so that their python + torch import
        # and their engines' workspaces come up beside the normaliser and the tilers, not in front of the first tile
        from . import multi
        multi.prestart()


class GridCreator(_cg.GridCreator):
    def __init__(self, quiet=False, engine=None, device=0, write_files="background"):
        super().__init__(quiet, engine=engine, device=device, write_files=write_files)


CryoEMPredictor = _pr.CryoEMPredictor
