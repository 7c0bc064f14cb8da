"""MI355X-native batched fruit-fly environment (hot path of talmolab/flybody `vnl_ray`).

`flybody_amd.fly_envs.flight_imitation(...)` returns a `BatchedFlyEnv` whose `step()` is one HIP kernel launch.
"""

__all__ = ["fly_envs", "BatchedFlyEnv"]


def __getattr__(name):
    if name == "BatchedFlyEnv":
        from .batched_env import BatchedFlyEnv

        return BatchedFlyEnv
    if name == "fly_envs":
        import importlib

        return importlib.import_module(".fly_envs", __name__)
    raise AttributeError(name)
