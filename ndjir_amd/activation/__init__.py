"""Activation operators (reference: python/activation/)."""
