"""MI355X-native GST-Tacotron inference hot path (HIP kernels behind a C-ABI)."""
from .hparams import load_hp, Dims  # noqa: F401
