"""rroi_align._ext -- ctypes binding of librroi_align_hip.so (see rroi_align/__init__.py)."""
