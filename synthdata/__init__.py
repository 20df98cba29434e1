"""Seeded synthetic inputs and weights (data only; shared by bench.py, tests/, tools/ and the oracle)."""
