"""MI355X-native forward/backward for a CTR model zoo."""
__version__ = "0.1.0"
