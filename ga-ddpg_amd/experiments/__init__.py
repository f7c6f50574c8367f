"""Configuration loading for the update-step path: `config.py` mirrors the reference's `experiments/config.py`
surface (`cfg`, `cfg_from_file`, `process_cfg`, attribute-style dictionaries) on top of the yaml files under
`ga-ddpg_amd/configs/` -- imported as `ga_ddpg_amd.experiments.config`."""
