"""Data-side types either side of the hot path (interface of turbdiff/data/ofles.py)."""
