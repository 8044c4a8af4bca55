from .snac import SNAC, make_snac_weights  # noqa: F401
