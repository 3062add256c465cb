"""On-device synthetic cone data: batches with the output contracts of the reference's datasets."""
from .synth import SyntheticCones, SyntheticConeCrops  # noqa: F401
