"""MI355X-native compress/decompress hot path of the convolutional autoencoder.

Drop-in names of the reference's ``models`` package for this path
(``src/models/tasks/_autoencoders.py``): ``Analyzer``, ``Synthesizer``,
``setup_modules``, ``load_state_dict``, ``autoencoder_from_state_dict``,
``ConvolutionalAutoencoder`` ('cae'), ``ConvolutionalAutoencoderBottleneck`` ('cae_bn');
plus ``EntropyBottleneck`` / ``GDN`` standing in for the compressai classes the reference imports.
"""
from ._lib import CaeError, LIB_PATH, build  # noqa: F401
from .entropy import EntropyBottleneck, pmf_to_quantized_cdf  # noqa: F401
from .modules import (GDN, Analyzer, DownsamplingUnit, ResidualDownsamplingUnit, ResidualUpsamplingUnit,  # noqa: F401
                      Synthesizer, UpsamplingUnit, initialize_weights)
from .codec import (ConvolutionalAutoencoder, ConvolutionalAutoencoderBottleneck,  # noqa: F401
                    autoencoder_from_state_dict, load_state_dict, register_codecs, setup_modules)

__all__ = ['Analyzer', 'Synthesizer', 'GDN', 'EntropyBottleneck', 'DownsamplingUnit', 'UpsamplingUnit', 'ResidualDownsamplingUnit',
           'ResidualUpsamplingUnit',
           'initialize_weights', 'setup_modules', 'load_state_dict', 'autoencoder_from_state_dict',
           'ConvolutionalAutoencoder', 'ConvolutionalAutoencoderBottleneck', 'register_codecs',
           'pmf_to_quantized_cdf', 'build', 'CaeError', 'LIB_PATH']
from . import criteria, metrics, train  # noqa: F401,E402  (validation objective; GPU metrics of the reference harness)
