from dataclasses import dataclass


@dataclass
class ModelConfig:
    """Base of every model configuration (reference tfimm/models/config.py:4-9)."""

    name: str = ""
    url: str = ""
