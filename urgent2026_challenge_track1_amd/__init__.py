"""MI355X-native URGENT-2026 track-1 speech-enhancement hot path (BSRNN train / eval)."""
__version__ = "0.1.0"
