"""Only the one constant of reference tools/tools.py the hot path needs (the speech encoders,
schedulers and alignment helpers there are out of scope)."""
from lds.arch import get_encoder_out_channels


def get_encdoer_out_channels(encoder):  # sic: the reference's spelling (tools/tools.py:257-264)
    return get_encoder_out_channels(encoder)
