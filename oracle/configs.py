"""Oracle (test infrastructure): config helpers.

The oracle consumes any mapping supporting ``cfg['key']`` and ``cfg.get('key')``
with the key names of ``snap/configs/defaults.py``.
"""


def get_block_desc(depth):
  """snap/models/resnet.py:158-167."""
  if isinstance(depth, list):
    depth = tuple(depth)
  return {
      26: [2, 2, 2, 2],
      50: [3, 4, 6, 3],
      101: [3, 4, 23, 3],
      152: [3, 8, 36, 3],
      200: [3, 24, 36, 3],
  }.get(depth, depth)
