"""sapr_amd — MI355X-native hot path of frankcholula/sapr assignment2.

Host-side mirror of the reference's module API (same names, arguments and error
behaviour) over hand-written HIP kernels in ``libsapr_hip.so``:

* ``sapr_amd.mfcc_extract``  ↔ assignment2/mfcc_extract.py
* ``sapr_amd.custom_hmm``    ↔ assignment2/custom_hmm.py
* ``sapr_amd.hmmlearn_hmm``  ↔ assignment2/hmmlearn_hmm.py (+ a GaussianHMM-shaped model object)
* ``sapr_amd.decoder``       ↔ assignment2/decoder.py

``sapr_amd/compat`` holds same-named top-level shims so the reference's ``train.py`` /
``eval.py`` / tests import the drop-in unmodified.
"""
__version__ = "0.1.0"
