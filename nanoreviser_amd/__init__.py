"""nanoreviser_amd - MI355X-native engine for NanoReviser's window reviser (model1+model2).

Scope: the hot path of SURVEY.md section 8 only.  `Reviser` (engine.py) is the drop-in for
the two Keras `Model.predict` callables of nanorevutils/output_handeler.py:206-307; it
runs hand-written gfx950 HIP kernels through the C-ABI of include/nanorev.h and has no
CPU fallback.
"""
from .weights import ModelWeights, load_model, load_species  # noqa: F401

__version__ = "0.1.0"
