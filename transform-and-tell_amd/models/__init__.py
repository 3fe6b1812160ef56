"""Host-side mirrors of tell/models (decoders + caption models) on the MI355X path."""
from .decoders import (Decoder, DecoderLayer, DynamicConvDecoder,  # noqa: F401
                       DynamicConvFacesObjectsDecoder, DynamicConvDecoderLayer)
from .transformer import (TransformerFacesObjectModel, TransformerFlattenedModel,  # noqa: F401
                          CaptionModel)
from .decoder_lstm import LSTMDecoder  # noqa: F401
from .baseline_glove import BaselineGloveModel, TransformerGloveModel  # noqa: F401
