from .encodec import Encodec, EncodecConfig, make_encodec_encoder_weights, make_encodec_weights, preprocess_audio  # noqa: F401
