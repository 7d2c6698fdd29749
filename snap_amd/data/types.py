"""Scene configuration consumed by the hot path (``snap/data/types.py:52-68``)."""
import dataclasses
from typing import Tuple


@dataclasses.dataclass
class SceneConfig:
  grid_size: Tuple[int, int, int] = (24, 32, 12)
  grid_z_offset: int = 4
  num_views: int = 10
  streetview_hfov_deg: float = 72.0
  camera_frustum_depth: float = 16.0


@dataclasses.dataclass
class BuildConfig:
  scene_config: SceneConfig = dataclasses.field(default_factory=SceneConfig)
