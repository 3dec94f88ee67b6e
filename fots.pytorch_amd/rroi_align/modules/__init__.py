"""rroi_align.modules -- the nn.Module surface (see rroi_align.py)."""
