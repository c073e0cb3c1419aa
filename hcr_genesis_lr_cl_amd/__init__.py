"""MI355X-native legged-locomotion environment engine (drop-in `Simulator` backend + fused env)."""
__version__ = "0.1.0"
