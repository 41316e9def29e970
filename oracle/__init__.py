"""CPU oracle for the SM_HPSS_MTL hot path.  TEST INFRASTRUCTURE ONLY.

This package restates, in numpy (and a little C under ``oracle/c``), the arithmetic of the
reference's hot path -- the librosa/scipy/sklearn/Cython front end of ``lib/preprocessing.py`` and
the keras / keras-tcn B3_MTL network of ``lib/proposed_architectures.py``.  It exists so that the
HIP product path can be checked against something that is *not* itself.

Rules (enforced by ``tests/test_host_logic.py::test_product_never_imports_oracle``):

* only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
  anything from here; the product package ``sm_hpss_mtl_amd`` never does and has no CPU fallback;
* nothing here imports the product package.

Parity status ("pins"):

* median filters, patch indexing, standardisation: pinned in-container against the very routines the
  reference delegates to (``scipy.ndimage.median_filter(mode='reflect')``, the reference's own
  ``tools.pyx`` compiled into ``oracle/_ref`` by ``oracle/Makefile``, ``sklearn.StandardScaler``);
  see ``tests/test_oracle_pins.py`` and the fixtures in ``tests/golden``.
* STFT / softmask / mel / power_to_db / B3_MTL: **parity unpinned** -- librosa, tensorflow and
  keras-tcn are not installable here (no network) and the reference ships no tests or golden
  vectors, so these are restated from the published algorithms of librosa 0.8, Keras 2.x and
  keras-tcn 2.3 and checked only through closed-form identities.
"""
