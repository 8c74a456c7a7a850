"""MI355X-native MuZero search / self-play engine with the reference's Python API."""
__version__ = "0.1.0"
