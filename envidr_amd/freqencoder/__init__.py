from .freq import FreqEncoder, freq_encode  # noqa: F401
