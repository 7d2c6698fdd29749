"""CPU oracle: a numpy restatement of the SNAP BEV-fusion + pose-matching hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  ``snap_amd`` never imports it and has no CPU fallback.

PARITY UNPINNED.  The reference (google-research/snap) ships no tests, golden
vectors or fixtures, and it cannot be imported in the build container (jax,
flax, ml_collections, chex, dataclass_array, etils and absl are absent and
there is no network).  The oracle is therefore pinned only against
  * ``scipy.ndimage.map_coordinates(order=1, mode='nearest')`` -- the documented
    model of ``jax.scipy.ndimage.map_coordinates`` used by
    ``snap/utils/grids.py:109-137``;
  * ``scipy.signal.convolve(method='direct')`` -- the model of
    ``jax.scipy.signal.convolve`` used by
    ``snap/models/pose_exhaustive_voting.py:86-91``;
  * ``torch.nn.functional`` conv2d / max_pool2d / interpolate on CPU for the
    encoder arithmetic (``snap/models/resnet.py``, ``image_encoder.py``);
  * analytic known-answer tests (identity / planted pose).
See ``tests/test_oracle_*.py`` and ``tests/golden/``.

Every function cites the reference file:line it restates.  All functions are
dtype-generic: float32 inputs give float32 arithmetic, float64 inputs float64.
"""
