"""graph-detr4d_amd: MI355X-native decoder hot path of Graph-DETR4D."""
__version__ = '0.1.0'
