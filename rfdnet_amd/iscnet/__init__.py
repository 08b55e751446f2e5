"""Host-side mirror of the reference's models/iscnet modules on the hot path."""
