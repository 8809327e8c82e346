"""CPU oracle of the hot path — test infrastructure only (see pose_oracle.py header)."""
