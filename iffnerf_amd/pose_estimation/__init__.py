"""Host-side mirror of the reference's ``pose_estimation`` package for the hot path (same module/function names)."""
