"""rroi_align.functions -- the autograd surface (see rroi_align.py)."""
