"""Evaluation overrides (model part of snap/configs/eval_localization.py:26-43)."""
from snap_amd.utils.config_dict import ConfigDict


def get_config() -> ConfigDict:
  return ConfigDict(
      batch_size=4, rng_seed=0, dtype_str='float32',
      model=dict(num_pose_samples=20_000, num_pose_sampling_retries=8,
                 do_grid_refinement=True),
  ).lock()
