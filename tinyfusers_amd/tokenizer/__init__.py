from .clip import ClipTokenizer  # noqa: F401
