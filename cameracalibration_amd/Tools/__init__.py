"""Drivers around the hot path that the reference keeps under Tools/ (SURVEY.md 8f item 2)."""
