"""MI355X-native bundle adjustment behind the reference's bundle_adjustment.h interface."""
__version__ = "0.1.0"
