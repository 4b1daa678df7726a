"""Drop-in Python classes for the reference's Hydra `_target_` plugin points (same class names, constructor kwargs, forward
signatures and state_dict key names as multi_view_generation.modules.*), backed by libbevgen_hip.  See INTEGRATION.md."""
