"""Solver adapters (same call signatures as ``pockit.optimizer``)."""
