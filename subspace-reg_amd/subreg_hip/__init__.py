"""subreg_hip — MI355X-native (gfx950) hot path of feyzaakyurek/subspace-reg.

Host-side mirror of the reference's module surface for the incremental-episode
path (models/resnet_language.py, eval/language_eval.py) over the C-ABI library
`libsubreg_hip.so` (include/subreg_hip.h).  Importing the package is cheap and
works without a GPU; every compute entry point loads the HIP library on first
use and raises if it is missing (there is NO CPU fallback).
"""
__all__ = ["synthetic"]
