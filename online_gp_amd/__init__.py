"""online_gp_amd -- MI355X-native WISKI streaming-update hot path
(drop-in for the corresponding path of wjmaddox/online_gp)."""
__version__ = "0.1.0"
