"""rassengine_amd — MI355X-native embedding + exact cosine k-NN engine that drops in behind
RASSEngine's ``embed_*()`` functions and ``OpenSearchIndexer`` (reference app/main.py:225-274,
1395-1560).  The compute path is hand-written HIP for gfx950 behind a C ABI
(``include/rass_engine.h``); there is no CPU fallback.
"""
__version__ = "0.1.0"
