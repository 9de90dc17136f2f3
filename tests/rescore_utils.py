"""Synthetic Feature tables for the post-search rescoring tests."""
from sage_amd.synthetic import synthetic_features  # noqa: F401
