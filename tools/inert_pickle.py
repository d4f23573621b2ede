"""Kept for the golden generator's import path; the reader lives in the package."""
from magnet_amd.utils.inert_pickle import *          # noqa: F401,F403
from magnet_amd.utils.inert_pickle import read_sparse_list, load_inert, to_numpy   # noqa: F401
