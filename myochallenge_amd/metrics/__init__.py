from .evaluation import evaluate_policy  # noqa: F401
from .custom_callbacks import CheckpointCallback, EnvDumpCallback, EvalCallback, EvaluateLSTM, TensorboardCallback  # noqa: F401
