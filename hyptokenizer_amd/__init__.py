"""hyptokenizer_amd: MI355X-native hyperbolic merge engine (hot path of HypTokenizer)."""
