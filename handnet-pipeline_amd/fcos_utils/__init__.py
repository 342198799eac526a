"""Drop-in for the reference package `fcos_utils` (inference entry point only)."""
