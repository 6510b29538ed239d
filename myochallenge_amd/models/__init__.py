from .classifier import DIMS_PER_OBS, N_OBS_PER_TRIAL, TaskClassifier, load_scaler  # noqa: F401
