"""Part of the MI355X-native detection path."""
