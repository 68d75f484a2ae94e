"""Drop-in for the reference's `model.classifier` import path (eval/main_classifier.py:26):
re-exports the MI355X-native implementation."""
from coclr_amd.model.classifier import LinearClassifier  # noqa: F401
