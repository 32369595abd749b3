"""MI355X-native flexibility-provision environment + safe-MADDPG hot path.

Drop-in surface (same names as the reference):
    FlexibilityProvisionEnv, TransReplayBuffer, PGTrainer, MADDPG, SAFEMADDPG
plus the batched device objects they are built on (VecFlexProvisionEnv, ...).
"""
from .network import create_network, build_tables, NetTables          # noqa: F401
from .series import SeriesTable, make_synthetic_series, from_frames   # noqa: F401


def __getattr__(name):
    # torch-dependent modules are imported on first use so that `import safe_marl_amd` stays cheap
    if name in ("VecFlexProvisionEnv", "FlexibilityProvisionEnv", "pf_solve_batch"):
        from . import flex_env
        return getattr(flex_env, name)
    if name in ("TransReplayBuffer", "DeviceReplayBuffer"):
        from . import replay_buffer
        return getattr(replay_buffer, name)
    if name in ("MADDPG", "SAFEMADDPG", "MATD3", "IDDPG", "Model"):
        from . import learner
        return getattr(learner, name)
    if name == "PGTrainer":
        from . import trainer
        return trainer.PGTrainer
    if name == "PGTester":
        from . import tester
        return tester.PGTester
    if name in ("opf_model", "BatchedOPF"):
        from . import opf
        return getattr(opf, name)
    raise AttributeError(name)
