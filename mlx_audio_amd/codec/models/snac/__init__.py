from .snac import SNAC, make_snac_encoder_weights, make_snac_weights  # noqa: F401
