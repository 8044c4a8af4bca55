from .mimi import Mimi, MimiConfig, MimiDecoder, MimiEncoder, mimi_202407  # noqa: F401
