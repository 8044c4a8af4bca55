from .vocos import ISTFTHead, MelSpectrogramFeatures, EncodecFeatures, Vocos, VocosBackbone, make_vocos_weights  # noqa: F401
