"""Reference import path `lavila.models.coca` -> the two classes of it the narrator uses (`CrossAttention`, `LayerNorm`,
coca.py:25-131), MI355X-native (lavila_amd.narrator)."""
from lavila_amd.narrator import CrossAttention, LayerNorm  # noqa: F401
