from .mimi import MimiConfig, MimiDecoder, mimi_202407  # noqa: F401
