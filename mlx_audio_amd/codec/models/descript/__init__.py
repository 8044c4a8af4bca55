from .base import DACFile  # noqa: F401
from .dac import DAC, make_dac_encoder_weights, make_dac_weights  # noqa: F401
