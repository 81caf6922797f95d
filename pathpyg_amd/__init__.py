"""pathpyg_amd — MI355X-native (gfx950) engine behind pathpyG's higher-order-graph API."""
__version__ = "0.1.0"
