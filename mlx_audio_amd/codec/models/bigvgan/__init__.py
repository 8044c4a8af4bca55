from .bigvgan import BigVGAN, BigVGANConfig, make_bigvgan_weights  # noqa: F401
