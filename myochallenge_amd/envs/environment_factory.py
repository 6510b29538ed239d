"""``EnvironmentFactory.create(name, **kwargs)`` — same name and call shape as the reference
(/root/reference/src/envs/environment_factory.py:8-63), returning a batched GPU environment.

The reference returns ONE gym env per call and builds 16 of them in worker processes
(/root/reference/src/main_baoding.py:56-65).  Here the extra keyword ``num_envs`` (default 1)
sizes the batch; every other kwarg is the reference's env config, unchanged.
"""
from __future__ import annotations

from .baoding import BaodingVecEnv
from .config import REGISTRATION

_BATCH_KEYS = ("num_envs", "device", "seed", "dtype", "model", "integrator")


class EnvironmentFactory:
    """Static factory to instantiate environments by name."""

    @staticmethod
    def create(env_name, **kwargs):
        """Creates an environment given its name as a string, and forwards the kwargs to it.

        Raises:
            ValueError: if the name of the environment is unknown (as the reference does)
        """
        batch_kw = {k: kwargs.pop(k) for k in _BATCH_KEYS if k in kwargs}
        num_envs = batch_kw.pop("num_envs", 1)
        if env_name == "MixtureModelBaodingEnv":        # src/envs/baoding.py:650-714
            from .mixture import MixtureModelBaodingVecEnv
            mix = {k: kwargs.pop(k) for k in ("base_model_path", "base_env_path", "base_env_name", "base_env_config",
                                             "n_steps_base_model", "base_policy", "base_normalizer", "pool_size") if k in kwargs}
            return MixtureModelBaodingVecEnv(env_name, num_envs, kwargs, **mix, **batch_kw)
        if env_name in REGISTRATION:
            return BaodingVecEnv(env_name, num_envs, kwargs, **batch_kw)
        if env_name in ("CustomMyoReorientP1", "CustomMyoReorientP2"):          # src/envs/reorient.py
            from .reorient import ReorientVecEnv
            return ReorientVecEnv(env_name, num_envs, kwargs, **batch_kw)
        known_elsewhere = ("MyoFingerPoseFixed", "MyoFingerPoseRandom", "MyoFingerReachFixed",
                           "MyoFingerReachRandom", "MyoHandKeyTurnFixed", "MyoHandKeyTurnRandom",
                           "MyoBaodingBallsP1",
                           "MyoBaodingBallsP2", "CustomMyoElbowPoseFixed",
                           "CustomMyoElbowPoseRandom", "CustomMyoFingerPoseFixed",
                           "CustomMyoFingerPoseRandom", "CustomMyoHandPoseFixed",
                           "CustomMyoHandPoseRandom", "CustomMyoPenTwirlRandom")
        if env_name in known_elsewhere:
            raise NotImplementedError(
                f"{env_name}: named by the reference but outside this build's hot-path scope "
                "(SURVEY.md §8: the Baoding P1 / P2 / MixtureModel and die-reorient P1 / P2 envs are implemented so far)")
        raise ValueError("Environment name not recognized:", env_name)
