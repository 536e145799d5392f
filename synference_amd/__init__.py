"""MI355X-native amortised-posterior flow engine with Synference's API surface for that path."""
from .spec import FlowSpec  # noqa: F401

__all__ = ["FlowSpec"]
