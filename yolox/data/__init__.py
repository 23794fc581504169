from .synthetic import SyntheticClipLoader, get_timing_signal_1d
