"""CPU oracle for the sapr hot path (MFCC front-end + Gaussian-HMM trellis).

TEST INFRASTRUCTURE ONLY.  Nothing under ``sapr_amd/`` imports this package.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import, call, link or execute anything here, and there only as the
checker — never as the thing measured as the product or shipped.

Pinning status (details in DESIGN.md §3):

* ``custom_hmm_oracle``  – PINNED.  Validated in the build container against the
  imported reference ``assignment2/custom_hmm.py`` (bit-for-bit on the
  recurrences, <=1e-12 on BLAS-order-dependent pieces) and against the
  dataset-independent known answers in the reference's
  ``pytest_results/*.txt``.  Golden vectors live in ``tests/golden/``.
* ``hmmlearn_oracle``    – PARITY UNPINNED.  hmmlearn 0.3.3 is an un-vendored
  dependency (``assignment2/poetry.lock:430-431``) that is absent from the
  container; the file restates its published algorithm and is anchored on the
  reference's call sites (``hmmlearn_hmm.py:27-43,103-104``, ``decoder.py:43``).
* ``mfcc_oracle``        – PARITY UNPINNED.  librosa 0.10.2.post1 is un-vendored
  (``assignment2/poetry.lock:679-680``) and absent; restated from its published
  algorithm, anchored on ``mfcc_extract.py:12-23`` and cross-checked piecewise
  against scipy / torch.stft and, end to end, against the librosa-compatible
  routines of ``transformers.audio_utils`` (filterbank 4e-16, cepstra 4.1e-5).
"""
