"""crispresso2_amd -- MI355X-native align + classify hot path for CRISPResso2."""
__version__ = "0.1.0"
