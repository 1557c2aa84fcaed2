"""dae_rnn_news_recommendation_amd -- MI355X-native denoising-autoencoder article-embedding trainer.

Drop-in for the training path of louislung/DAE_RNN_News_Recommendation
(``DenoisingAutoencoder.fit()/transform()`` + the ``main_autoencoder.py`` CLI), with the per-mini-batch
hot path implemented as hand-written gfx950 HIP kernels behind the C ABI of ``include/dae_hip.h``.
"""
__version__ = "0.1.0"
